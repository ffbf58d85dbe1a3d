// scan_topk.hip — K1+K2 of SURVEY §8a: fused flat cosine scan + top-k for gfx950.
//
// Replaces the arithmetic behind OpenSearchIndexer.semantic_search's knn query
// (reference app/main.py:1527-1560; HNSW walk in the k-NN plugin) with an exact
// scan: S[q][row] = <Qn[q], Xn[row]> over a row-major fp32 corpus resident in HBM,
// top-k per query under the total order (score desc, id asc).
//
// Shape of the kernel (one 512-thread workgroup per CU, persistent over row tiles):
//   * HBM layout ("tile16", kernels.h): the corpus is stored in 16-row blocks; inside a block
//     the 16 rows x 16 columns (64 B each) of chunk j are ONE contiguous 1 KiB in MFMA lane
//     order, lane (m = lane&15, g = lane>>4) <-> X[16b+m][16j + 4g .. +3].  So every wave-level
//     load is a fully coalesced 1 KiB burst that lands directly in A-operand layout (measured:
//     6.6 TB/s for this pattern vs 5.6 TB/s for 16 rows x 64 B fragment-shaped loads from a
//     plain row-major slab; scripts/microbench/stream_patterns.hip)
//   * a tile is 32 corpus rows (two blocks); the 8 waves split the K (=dim) axis, wave w owns
//     chunks [w*CH, (w+1)*CH) of every block (dim_padded = 128*CH)
//   * corpus bytes go HBM -> VGPR directly (read-once stream, GEMV-shaped: no LDS
//     round trip), 16 B per lane, as the A operand of v_mfma_f32_16x16x4_f32
//   * the query fragments (B operand) stay in registers for the whole launch
//   * loads are buffer loads through a per-tile descriptor, so rows past n_rows and
//     the run-ahead prefetch of a tile past the end cost no HBM traffic (range check)
//   * two register tile buffers per wave; each 16 B register is re-loaded for the tile two
//     steps ahead right after the MFMAs that consumed it, so ~32 KiB per wave (256 KiB per
//     CU) stay in flight continuously; ONE VGPR carries the lane's load offset, the M-tile /
//     chunk part goes into the instruction's SGPR and immediate offset fields
//   * per tile, each wave dumps its 32x(16*NT) partial scores to LDS; two tiles share one barrier
//     (four LDS images).  The RANKING of a tile pair — wave w sums the 8 K-partials of "its" queries
//     in a fixed order, filters by threshold (ballot) and inserts into the half-wave sorted lists —
//     runs one iteration later, cut into 2*NT parts placed BETWEEN the MFMA chunks of the next pair,
//     so its LDS reads and VALU work issue while the matrix pipe executes (round 1 ranked in a phase
//     of its own behind the barrier: the pipe idled while all 8 waves ranked, 72 us of 690 at B = 32;
//     profiles/r02_scan_phase_experiments.txt).  Row tags and per-query filters live in LDS
//   * the insertion itself is register / scalar only: DPP wave_shr for the shift, readlane for the
//     broadcasts, bitwise predicates + selects instead of exec-mask branches (scan_core.h)
//   * at the end each workgroup writes a sorted top-k list per query; merge_topk.hip
//     reduces [n_workgroups][nq][k] -> [nq][k]
//   * EXT variant (template flag, same main loop): masked tag filters and the continuation bound of
//     a multi-pass top-k (k > 32), parameters in LDS
//
// Numerics: v_mfma_f32_16x16x4_f32 is an exact f32 fmaf chain in k order, so a score is
// a fixed sequence of fmaf's per K-slice followed by 7 f32 adds: independent of the grid,
// of the shard split and of the query batch (oracle/rass_oracle.c restates that order
// on the CPU and the GPU tests require bit-equality with it).
//
// Algorithmic bytes per launch (roofline.achieved in bench.py): n_rows * row_stride * 4.
// The corpus must be allocated in whole 16-row blocks (rows past n_rows are masked).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "scan_core.h"

namespace rass {

#ifdef RASS_SCAN_DIAG  // diagnostic builds only (scripts/probe_scan_diag.py): launch-wide event counters
// [0] ranking calls, [1] calls with >= 1 candidate, [2] candidates, [4] candidates the sample floor rejected
__device__ unsigned long long g_scan_diag[8];
#endif

#ifdef RASS_SCAN_CLOCKS  // scripts/microbench/scan_tail.hip only: per-workgroup start / end wall clocks
__device__ unsigned long long g_scan_clocks[2 * 1024];
__device__ unsigned long long g_scan_core[2];
#endif

// The ascending sequence of work-item indices one workgroup handles.  Plain: b, b+G, b+2G, ...
// With XCD skew s: the items are cut into super-rounds of s*G + G/2; in each, every workgroup
// takes s items round-robin (j*G + b) and the even workgroups one more (s*G + b/2).
struct ItemSeq {
    int base, j, lim, skew, G, b, period;
    __device__ __forceinline__ ItemSeq(int b_, int G_, int skew_) : base(0), j(0), skew(skew_), G(G_), b(b_) {
        lim = skew == 0 ? 1 : ((b & 1) ? skew : skew + 1);
        period = skew == 0 ? G : skew * G + (G >> 1);
    }
    __device__ __forceinline__ int next() {
        const int r = (skew == 0 || j < skew) ? base + j * G + b : base + skew * G + (b >> 1);
        if (++j == lim) {
            j = 0;
            base += period;
        }
        return r;
    }
};

template <int MODE>
__device__ __forceinline__ WorkItem get_work(const ScanArgs& p, int i, int n_items) {
    WorkItem w;
    if (!mode_is_flat(MODE)) {
        const bool ok = i < n_items;
        w.tile = ok ? p.work_tile[i] : 0;
        w.rows = ok ? p.work_rows[i] : 0;
        w.mask = ok ? p.work_mask[i] : 0u;
    } else {
        int rows = p.n_rows - i * kTileRows;
        rows = rows < 0 ? 0 : (rows > kTileRows ? kTileRows : rows);
        w.tile = i < n_items ? i : 0;
        w.rows = i < n_items ? rows : 0;
        w.mask = 0xffffffffu;
    }
    return w;
}

// The slab / tag array a work item streams: the launch's one slab, or (MULTI) the item's own.  `i` past the end
// reads item 0's pointers (rows = 0 there, so the descriptor has zero records and nothing is fetched).
template <int MODE>
__device__ __forceinline__ const float* slab_of(const ScanArgs& p, int i, int n_items) {
    return MODE == kMulti ? p.work_base[i < n_items ? i : 0] : p.corpus;
}
template <int MODE>
__device__ __forceinline__ const int32_t* tags_of(const ScanArgs& p, int i, int n_items) {
    return MODE == kMulti ? p.work_tags[i < n_items ? i : 0] : p.row_tag;
}

template <int CH, int NT, int MODE, bool EXT>
__global__ __launch_bounds__(kThreads, 2) void scan_topk_f32_kernel(ScanArgs p) {
    constexpr bool IVF = !mode_is_flat(MODE);  // a work list with per-item query masks
    constexpr int NQ = NT * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][kWaves][NQ][kPitch]

    const int lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    // number of work items: tiles of the slab (flat) or entries of the probe plan (IVF; written
    // by plan_probe_kernel earlier on this stream)
    int G = gridDim.x;
    int bid = blockIdx.x;   // this workgroup's position among the G that share its queries
    if (MODE == kFlatGroups || MODE == kIvfGroups || MODE == kFlatSampleGroups) {
        const int grp = bid / p.wgs_per_group;
        bid -= grp * p.wgs_per_group;
        G = p.wgs_per_group;
        p.q_padded += (int64_t)grp * p.q_group_stride;
        p.part_scores += (int64_t)grp * p.part_group_stride;
        if (MODE != kFlatSampleGroups) p.part_ids += (int64_t)grp * p.part_group_stride;
        if (p.q_filter != nullptr) p.q_filter += grp * 32;
        if (MODE == kFlatSampleGroups) p.nq = min(32, p.nq_total - 32 * grp);
        if (MODE == kIvfGroups) {
            p.work_tile += (int64_t)grp * p.work_group_stride;
            p.work_rows += (int64_t)grp * p.work_group_stride;
            p.work_mask += (int64_t)grp * p.work_group_stride;
            p.n_work += grp;
            p.nq = min(32, p.nq_total - 32 * grp);
        }
    }
    const int n_tiles = IVF ? __builtin_amdgcn_readfirstlane(*p.n_work) : (p.n_rows + kTileRows - 1) / kTileRows;
#ifdef RASS_SCAN_CLOCKS
    if (threadIdx.x == 0) g_scan_clocks[2 * blockIdx.x] = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) g_scan_core[0] = clock64();
#endif

    // Query fragments: lane (n = m, g) holds Qn[nt*16 + n][slice + 16j + 4g .. +3].
    f32x4 qf[NT][CH];
    {
        const float* qbase = p.q_padded + (int64_t)m * p.row_stride + wid * 16 * CH + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < CH; ++j)
                qf[nt][j] = *reinterpret_cast<const f32x4*>(qbase + (int64_t)nt * 16 * p.row_stride + 16 * j);
    }
    // Byte offset of this lane inside a tile: block 0, this wave's chunk 0, 16 B per lane.
    const int voff_lane = wid * CH * 1024 + lane * 16;

    // Top-k state: pass pq handles query pq*16 + (lane>>5)*8 + wid for row lane&31.
    TopList L[NT];
    float tau[NT];
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) {
        L[pq].s = -INFINITY;
        L[pq].i = 0x7fffffff;
        tau[pq] = -INFINITY;
    }
    // Per-query filter parameters and the dumped tiles' row tags live in LDS (1.5 KiB next to the 144 KiB of
    // partial images), not in registers: they are read in the ranking step only, and the main loop has no VGPRs
    // to spare at CH = 8, NT = 2 (the ranking runs between the MFMA chunks, so its operands are live there).
    __shared__ int sh_qfilt[32];
    __shared__ float sh_floor[32];  // the launch's per-query score floors (ScanArgs::sample_best), -inf without
    __shared__ int sh_tags[4][32];
    __shared__ int sh_qmask[32];
    __shared__ float sh_after_s[32];
    __shared__ int64_t sh_after_i[32];
    if (threadIdx.x < 32) {
        const int q = threadIdx.x;
        const bool live = q < p.nq;
        sh_qfilt[q] = (p.q_filter != nullptr && live) ? p.q_filter[q] : -1;
        sh_floor[q] = -INFINITY;
        if (EXT) {
            sh_qmask[q] = (p.q_filter_mask != nullptr && live) ? p.q_filter_mask[q] : -1;
            sh_after_s[q] = (p.q_after_score != nullptr && live) ? p.q_after_score[q] : INFINITY;
            sh_after_i[q] = (p.q_after_id != nullptr && live) ? p.q_after_id[q] : (int64_t)-1;
        }
    }
    __syncthreads();

    // The sample pass's per-workgroup best scores of this wave's 2 x NT queries, as order-preserving keys (0 = none):
    // loaded BEFORE the first tiles so that the selection below runs while those are in flight.
    unsigned best_key[NT][2][kMaxSampleGroups / 64];
    if (MODE == kFlat && p.sample_best != nullptr) {
#pragma unroll
        for (int pq = 0; pq < NT; ++pq)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int j = 0; j < kMaxSampleGroups / 64; ++j) {
                    const int grp = lane + 64 * j;
                    const float v = grp < p.sample_groups ? p.sample_best[(pq * 16 + half * 8 + wid) * kMaxSampleGroups + grp] : -INFINITY;
                    best_key[pq][half][j] = v == -INFINITY ? 0u : score_key(v);
                }
    }

    const int mt_step = 16 * (int)p.row_stride * 4;
    TileRegs<CH> R0, R1;
    ItemSeq seq(bid, G, (G & 1) ? 0 : p.xcd_skew);
    int t = seq.next();  // the item R0 holds; R1 holds the one after it
    const int t1 = seq.next();
    WorkItem W0 = get_work<MODE>(p, t, n_tiles), W1 = get_work<MODE>(p, t1, n_tiles);
    issue_tile_loads<CH>(R0, make_tile_desc(slab_of<MODE>(p, t, n_tiles), p.row_stride, tags_of<MODE>(p, t, n_tiles), W0),
                         voff_lane, mt_step);
    issue_tile_loads<CH>(R1, make_tile_desc(slab_of<MODE>(p, t1, n_tiles), p.row_stride, tags_of<MODE>(p, t1, n_tiles), W1),
                         voff_lane, mt_step);
    __builtin_amdgcn_sched_barrier(0);

    // The sample floor of each of this wave's queries: the k-th largest of the sample pass's per-workgroup bests
    // (those workgroups scanned disjoint rows, so k rows of the slab reach it under the query's filters, and so
    // does the final k-th best; fewer than k finite entries: no floor).  Radix selection over the keys' top
    // kFloorBits bits by ballot counts — the truncation only lowers the floor by < 2^-11 of its value.
    if (MODE == kFlat && p.sample_best != nullptr) {
        constexpr int kFloorBits = 20;
        unsigned T[NT][2] = {};
        // bit by bit, the wave's 2 x NT selections side by side (each is a chain of dependent scalar steps)
#pragma unroll 1
        for (int b = 31; b >= 32 - kFloorBits; --b) {
#pragma unroll
            for (int pq = 0; pq < NT; ++pq)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const unsigned cand = T[pq][half] | (1u << b);
                    int c = 0;
#pragma unroll
                    for (int j = 0; j < kMaxSampleGroups / 64; ++j) c += __popcll(__ballot(best_key[pq][half][j] >= cand));
                    T[pq][half] = c >= p.k ? cand : T[pq][half];
                }
        }
#pragma unroll
        for (int pq = 0; pq < NT; ++pq)
#pragma unroll
            for (int half = 0; half < 2; ++half)
                if (lane == 0) sh_floor[pq * 16 + half * 8 + wid] = T[pq][half] ? key_score(T[pq][half]) : -INFINITY;
    }

    // A tile's 8 K-partials meet in LDS.  Two tiles share ONE barrier: both are dumped (four LDS
    // images: two per loop iteration, alternating between iterations so a fast wave's next dump never
    // lands on an image a slow wave is still reading — it would have to pass the next barrier first),
    // then ranked in ascending row order.
    auto dump_tile = [&](const f32x4 (&acc)[2][NT], int buf) {
        float* P = lds + buf * (kWaves * NQ * kPitch);
        // lane (n=m, g) holds rows 4g..4g+3 of M-tile mt for query nt*16+n
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *reinterpret_cast<f32x4*>(P + (wid * NQ + nt * 16 + m) * kPitch + mt * 16 + 4 * g) = acc[mt][nt];
    };
#ifdef RASS_SCAN_DIAG
    unsigned d_calls = 0, d_any = 0, d_cand = 0, d_rej = 0;
#endif
    // Rank one (tile, query group pq) of a dumped tile: sum the 8 K-partials in fixed order, filter, insert.
    auto rank_part = [&](const WorkItem& w, int buf, int pq) {
        const float* P = lds + buf * (kWaves * NQ * kPitch);
        const int r = lane & 31;
        const int row = w.tile * kTileRows + r;
        const int tag = sh_tags[buf][r];
        const bool row_ok = (r < w.rows) && (tag != -1);
        const int q = pq * 16 + (lane >> 5) * 8 + wid;
        const int qf1 = sh_qfilt[q];
        const float* src = P + q * kPitch + r;
        float s = src[0];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) s += src[wv * NQ * kPitch];
        bool ok;
        if (EXT) {
            ok = row_ok && (qf1 < 0 || qf1 == (tag & sh_qmask[q]));
            // continuation: strictly after (after_s, after_i) in (score desc, id asc)
            const float as = sh_after_s[q];
            ok = ok && (s < as || (s == as && (p.id_base + (int64_t)row) > sh_after_i[q]));
        } else {
            ok = row_ok && (qf1 < 0 || qf1 == tag);
        }
        if (IVF) ok = ok && ((w.mask >> q) & 1u);
        // the sample floor: k rows of the corpus already score >= floor_q, so a row below it cannot be in the
        // query's top-k (ties are kept: the id order decides them in the merge)
        const float floor_q = MODE == kFlat ? sh_floor[q] : -INFINITY;
#ifdef RASS_SCAN_DIAG
        d_rej += __popcll(__ballot(ok && s > tau[pq] && s < floor_q));
#endif
        if (MODE == kFlat) ok = ok && (s >= floor_q);
        s = ok ? s : -INFINITY;
#ifdef RASS_SCAN_DIAG
        {
            const int c = __popcll(__ballot(s > tau[pq]));
            d_calls += 1; d_any += c != 0; d_cand += c;
        }
#endif
        if (mode_is_sample(MODE))
            L[pq].s = fmaxf(L[pq].s, s);  // the sample pass keeps each row slot's best score, nothing else
        else
            insert_candidates(L[pq], tau[pq], s, row, p.k);
    };

    // Main loop.  Iteration i multiplies the tile pair (A_i, B_i) and, between the MFMA chunks, ranks the pair
    // of iteration i-1 out of the LDS images that iteration dumped (2 x NT parts spread over the 2 x CH chunk
    // slots; rows are still ranked in ascending order, which the strict `>` tie rule needs).  One barrier per
    // iteration: it orders this iteration's dumps before the next iteration's reads, and the next iteration's
    // dumps (into the images read now) behind this iteration's reads.
    constexpr int kParts = 2 * NT;
    auto slot_of = [](int part) { return ((part + 1) * 2 * CH) / kParts - 1; };
    WorkItem Pa{0, 0, 0u}, Pb{0, 0, 0u};  // the previous pair (rows = 0: nothing ranks in the first iteration)
    int pair = 0;  // 0 / 2: which two LDS images this iteration dumps into
    while (t < n_tiles) {
        f32x4 acc[2][NT];
        // the pair's row tags go to LDS now (every wave loaded the same 32 per tile): the refill below reuses
        // R.tag for the tile two steps ahead, and the ranking reads them an iteration later
        if (wid == 0 && lane < 32) {
            sh_tags[pair][lane] = R0.tag;
            sh_tags[pair + 1][lane] = R1.tag;
        }
        const WorkItem Wa = W0;
        t = seq.next();
        WorkItem Wn = get_work<MODE>(p, t, n_tiles);
        auto rank_prev = [&](int slot) {
#pragma unroll
            for (int part = 0; part < kParts; ++part)
                if (slot_of(part) == slot) {
                    if (part < NT)
                        rank_part(Pa, pair ^ 2, part);
                    else
                        rank_part(Pb, (pair ^ 2) + 1, part - NT);
                }
        };
        multiply_and_refill<CH, NT>(R0, qf, acc,
                                    make_tile_desc(slab_of<MODE>(p, t, n_tiles), p.row_stride, tags_of<MODE>(p, t, n_tiles), Wn),
                                    voff_lane, mt_step, [&](int j) { rank_prev(j); });
        dump_tile(acc, pair);
        W0 = Wn;
        const WorkItem Wb = W1;
        const int tn = seq.next();
        Wn = get_work<MODE>(p, tn, n_tiles);
        multiply_and_refill<CH, NT>(R1, qf, acc,
                                    make_tile_desc(slab_of<MODE>(p, tn, n_tiles), p.row_stride, tags_of<MODE>(p, tn, n_tiles), Wn),
                                    voff_lane, mt_step, [&](int j) { rank_prev(CH + j); });
        dump_tile(acc, pair + 1);
        W1 = Wn;
        Pa = Wa;
        Pb = Wb;
        __syncthreads();
        pair ^= 2;
    }
    // the last pair
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) rank_part(Pa, pair ^ 2, pq);
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) rank_part(Pb, (pair ^ 2) + 1, pq);

#ifdef RASS_SCAN_DIAG
    if (lane == 0) {
        atomicAdd(&g_scan_diag[0], (unsigned long long)d_calls);
        atomicAdd(&g_scan_diag[1], (unsigned long long)d_any);
        atomicAdd(&g_scan_diag[2], (unsigned long long)d_cand);
        atomicAdd(&g_scan_diag[4], (unsigned long long)d_rej);
    }
#endif
#ifdef RASS_SCAN_CLOCKS
    if (threadIdx.x == 0) g_scan_clocks[2 * blockIdx.x + 1] = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) g_scan_core[1] = clock64();
#endif
    if (mode_is_sample(MODE)) {
        // The sample pass: this workgroup's best score per query -> part_scores[32][kMaxSampleGroups] (-inf: no row of the
        // sample passed the query's filters); the big scan's waves take the k-th largest over the workgroups.
#pragma unroll
        for (int pq = 0; pq < NT; ++pq) {
            float v = L[pq].s;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 32));
            const int q = pq * 16 + (lane >> 5) * 8 + wid;
            if ((lane & 31) == 0) p.part_scores[q * kMaxSampleGroups + bid] = q < p.nq ? v : -INFINITY;
        }
        return;
    }
    // Per-workgroup sorted lists -> [gridDim.x][nq][k]
    const int lpos = lane & 31;
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) {
        const int q = pq * 16 + (lane >> 5) * 8 + wid;
        if (q < p.nq && lpos < p.k) {
            const int64_t o = ((int64_t)bid * p.nq + q) * p.k + lpos;
            const bool filled = L[pq].i != 0x7fffffff;
            p.part_scores[o] = filled ? L[pq].s : -INFINITY;
            p.part_ids[o] = filled ? (p.id_base + (int64_t)L[pq].i) : (int64_t)-1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Wide rows: 1 024 < dim <= 2 048 (row_stride = 256 * {5..8}).  EMBED_DIM is an environment knob of the reference
// (app/main.py:80) and the encoder serves hidden sizes up to 2 048, so the index must take what the encoder emits.
// A wave's K slice (row_stride / 8 columns) no longer fits its registers as one tile image, so it is walked in P
// PANELS of CHP chunks (P = 2 up to 1 792 columns, P = 4 x 4 chunks at 2 048): the two register images R0 / R1 hold
// panels pn and pn + 1 of the current tile; a panel's MFMAs continue the accumulators of the panel before it (the
// fmaf chain of the slice runs on in k order: a score is the same fixed sequence the oracle's emulation restates for
// this stride), and each consumed register is refilled with the panel TWO steps ahead in the (tile, panel) sequence.
// The query fragments of all panels stay in registers (row_stride / 32 VGPRs): that is what limits this kernel to 16
// queries per launch (NT = 1; a 32-query group runs as two launches, api.hip).  Flat scans only (no IVF plan, no
// cross-index work list, no sample floor: with <= 16 queries the scan is HBM-bound and the ranking hides under it).
// Everything else — descriptors, XCD-aware item order, LDS images, one barrier per tile pair, ranking of the previous
// pair between the MFMA chunks, register-only insertion, EXT filters / continuation bound — is the kernel above.
template <int PN, int P, int CHP, typename Rank>
__device__ __forceinline__ void wide_panels(TileRegs<CHP>& R0, TileRegs<CHP>& R1, const f32x4 (&qf)[P][1][CHP],
                                            f32x4 (&acc)[2][1], const TileDesc& dcur, const TileDesc& dnext,
                                            int voff_lane, int mt_step, Rank&& rank) {
    if constexpr (PN < P) {
        constexpr bool to_next = PN + 2 >= P;                      // the refill target lies in the next tile
        constexpr int target = to_next ? PN + 2 - P : PN + 2;      // ... and is this panel of it
        multiply_and_refill<CHP, 1, to_next && target == 0, PN == 0>(
            (PN & 1) ? R1 : R0, qf[PN], acc, to_next ? dnext : dcur, voff_lane, mt_step,
            [&](int j) { if (PN == P - 1 && j == CHP - 1) rank(); }, target * CHP * 1024);
        wide_panels<PN + 1, P, CHP>(R0, R1, qf, acc, dcur, dnext, voff_lane, mt_step, rank);
    }
}

template <int CHP, int P, bool EXT>
__global__ __launch_bounds__(kThreads, 2) void scan_topk_f32_wide_kernel(ScanArgs p) {
    static_assert(P == 2 || P == 4, "R0 holds the even panels, R1 the odd ones");
    constexpr int NQ = 16, CHT = P * CHP;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][kWaves][NQ][kPitch]
    const int lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int n_tiles = (p.n_rows + kTileRows - 1) / kTileRows;
    const int G = gridDim.x;

    f32x4 qf[P][1][CHP];  // [panel]: lane (n = m, g) holds Qn[n][slice + 16 (panel CHP + j) + 4g .. +3]
    {
        const float* qbase = p.q_padded + (int64_t)m * p.row_stride + wid * 16 * CHT + 4 * g;
#pragma unroll
        for (int pn = 0; pn < P; ++pn)
#pragma unroll
            for (int j = 0; j < CHP; ++j) qf[pn][0][j] = *reinterpret_cast<const f32x4*>(qbase + 16 * (pn * CHP + j));
    }
    const int voff_lane = wid * CHT * 1024 + lane * 16;

    TopList L;
    L.s = -INFINITY;
    L.i = 0x7fffffff;
    float tau = -INFINITY;
    __shared__ int sh_qfilt[16];
    __shared__ int sh_tags[4][32];
    __shared__ int sh_qmask[16];
    __shared__ float sh_after_s[16];
    __shared__ int64_t sh_after_i[16];
    if (threadIdx.x < 16) {
        const int q = threadIdx.x;
        const bool live = q < p.nq;
        sh_qfilt[q] = (p.q_filter != nullptr && live) ? p.q_filter[q] : -1;
        if (EXT) {
            sh_qmask[q] = (p.q_filter_mask != nullptr && live) ? p.q_filter_mask[q] : -1;
            sh_after_s[q] = (p.q_after_score != nullptr && live) ? p.q_after_score[q] : INFINITY;
            sh_after_i[q] = (p.q_after_id != nullptr && live) ? p.q_after_id[q] : (int64_t)-1;
        }
    }
    __syncthreads();

    const int mt_step = 16 * (int)p.row_stride * 4;
    TileRegs<CHP> R0, R1;
    ItemSeq seq((int)blockIdx.x, G, (G & 1) ? 0 : p.xcd_skew);
    int t = seq.next();
    WorkItem W0 = get_work<kFlat>(p, t, n_tiles);
    TileDesc D0 = make_tile_desc(p.corpus, p.row_stride, p.row_tag, W0);
    issue_tile_loads<CHP, true>(R0, D0, voff_lane, mt_step, 0);
    issue_tile_loads<CHP, false>(R1, D0, voff_lane, mt_step, CHP * 1024);
    __builtin_amdgcn_sched_barrier(0);

    auto dump_tile = [&](const f32x4 (&acc)[2][1], int buf) {
        float* P_ = lds + buf * (kWaves * NQ * kPitch);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4*>(P_ + (wid * NQ + m) * kPitch + mt * 16 + 4 * g) = acc[mt][0];
    };
    // wave wid ranks queries wid and 8 + wid (the two halves of the wave) of a dumped tile
    auto rank_tile = [&](const WorkItem& w, int buf) {
        const float* P_ = lds + buf * (kWaves * NQ * kPitch);
        const int r = lane & 31;
        const int row = w.tile * kTileRows + r;
        const int tag = sh_tags[buf][r];
        const bool row_ok = (r < w.rows) && (tag != -1);
        const int q = (lane >> 5) * 8 + wid;
        const int qf1 = sh_qfilt[q];
        const float* src = P_ + q * kPitch + r;
        float s = src[0];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) s += src[wv * NQ * kPitch];
        bool ok;
        if (EXT) {
            ok = row_ok && (qf1 < 0 || qf1 == (tag & sh_qmask[q]));
            const float as = sh_after_s[q];
            ok = ok && (s < as || (s == as && (p.id_base + (int64_t)row) > sh_after_i[q]));
        } else {
            ok = row_ok && (qf1 < 0 || qf1 == tag);
        }
        s = ok ? s : -INFINITY;
        insert_candidates(L, tau, s, row, p.k);
    };

    WorkItem Pa{0, 0, 0u}, Pb{0, 0, 0u};
    int pair = 0;
    while (t < n_tiles) {
        f32x4 acc[2][1];
        // ---- tile A = W0 (its first two panels resident); the registers run on into tile B
        if (wid == 0 && lane < 32) sh_tags[pair][lane] = R0.tag;
        const WorkItem Wa = W0;
        const int tb = seq.next();
        const WorkItem Wb = get_work<kFlat>(p, tb, n_tiles);
        const TileDesc Db = make_tile_desc(p.corpus, p.row_stride, p.row_tag, Wb);
        wide_panels<0, P, CHP>(R0, R1, qf, acc, D0, Db, voff_lane, mt_step, [&]() { rank_tile(Pa, pair ^ 2); });
        dump_tile(acc, pair);
        // ---- tile B; the registers run on into tile C
        if (wid == 0 && lane < 32) sh_tags[pair + 1][lane] = R0.tag;
        t = seq.next();
        const WorkItem Wc = get_work<kFlat>(p, t, n_tiles);
        D0 = make_tile_desc(p.corpus, p.row_stride, p.row_tag, Wc);
        wide_panels<0, P, CHP>(R0, R1, qf, acc, Db, D0, voff_lane, mt_step, [&]() { rank_tile(Pb, (pair ^ 2) + 1); });
        dump_tile(acc, pair + 1);
        W0 = Wc;
        Pa = Wa;
        Pb = Wb;
        __syncthreads();
        pair ^= 2;
    }
    rank_tile(Pa, pair ^ 2);
    rank_tile(Pb, (pair ^ 2) + 1);

    const int lpos = lane & 31;
    const int q = (lane >> 5) * 8 + wid;
    if (q < p.nq && lpos < p.k) {
        const int64_t o = ((int64_t)blockIdx.x * p.nq + q) * p.k + lpos;
        const bool filled = L.i != 0x7fffffff;
        p.part_scores[o] = filled ? L.s : -INFINITY;
        p.part_ids[o] = filled ? (p.id_base + (int64_t)L.i) : (int64_t)-1;
    }
}

template <int CHP, int P, bool EXT>
static hipError_t launch_wide_variant(const ScanArgs& a, int grid, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)4 * kWaves * 16 * kPitch * sizeof(float);  // 72 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_topk_f32_wide_kernel<CHP, P, EXT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((scan_topk_f32_wide_kernel<CHP, P, EXT>), dim3(grid), dim3(kThreads), lds_bytes, stream, a);
    return hipGetLastError();
}

template <bool EXT>
static hipError_t launch_wide(int ch_total, const ScanArgs& a, int grid, hipStream_t stream) {
    switch (ch_total) {
        case 10: return launch_wide_variant<5, 2, EXT>(a, grid, stream);
        case 12: return launch_wide_variant<6, 2, EXT>(a, grid, stream);
        case 14: return launch_wide_variant<7, 2, EXT>(a, grid, stream);
        case 16: return launch_wide_variant<4, 4, EXT>(a, grid, stream);   // 2 x 8 chunks spill (256 VGPRs at 2 waves per SIMD)
        default: return hipErrorInvalidValue;
    }
}

template <int CH, int NT, int MODE, bool EXT>
static hipError_t launch_variant(const ScanArgs& a, int grid, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)4 * kWaves * NT * 16 * kPitch * sizeof(float);  // 144 KiB at NT = 2
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_topk_f32_kernel<CH, NT, MODE, EXT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((scan_topk_f32_kernel<CH, NT, MODE, EXT>), dim3(grid), dim3(kThreads), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int NT, int MODE, bool EXT = false>
static hipError_t launch_ch(int ch, const ScanArgs& a, int grid, hipStream_t stream) {
    switch (ch) {
        case 1: return launch_variant<1, NT, MODE, EXT>(a, grid, stream);
        case 2: return launch_variant<2, NT, MODE, EXT>(a, grid, stream);
        case 3: return launch_variant<3, NT, MODE, EXT>(a, grid, stream);
        case 4: return launch_variant<4, NT, MODE, EXT>(a, grid, stream);
        case 5: return launch_variant<5, NT, MODE, EXT>(a, grid, stream);
        case 6: return launch_variant<6, NT, MODE, EXT>(a, grid, stream);
        case 7: return launch_variant<7, NT, MODE, EXT>(a, grid, stream);
        case 8: return launch_variant<8, NT, MODE, EXT>(a, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

bool scan_supported_stride(int64_t row_stride) {
    if (row_stride % 128 != 0) return false;
    const int64_t ch = row_stride / 128;
    return (ch >= 1 && ch <= 8) || (ch >= 10 && ch <= 16 && ch % 2 == 0);  // wide rows: whole 256-column units
}

hipError_t launch_scan_topk_f32(const ScanArgs& a, int grid, hipStream_t stream) {
    if (!scan_supported_stride(a.row_stride)) return hipErrorInvalidValue;
    const int ch = (int)(a.row_stride / 128);
    const bool ext = a.q_filter_mask || a.q_after_score || a.q_after_id;
    if (a.wgs_per_group > 0 && (ext || ch > 8 || a.work_base)) return hipErrorInvalidValue;
    if (a.wgs_per_group > 0 && a.work_tile != nullptr) {  // the fine scans of several IVF launch groups in one launch
        if (!a.work_rows || !a.work_mask || !a.n_work || a.nq_total < 1 || grid % a.wgs_per_group != 0 || a.xcd_skew)
            return hipErrorInvalidValue;
        if (grid / a.wgs_per_group != (a.nq_total + 31) / 32) return hipErrorInvalidValue;
        return launch_ch<2, kIvfGroups>(ch, a, grid, stream);
    }
    if (ch > 8) {  // wide rows: flat scans of <= 16 queries (the caller splits larger groups)
        if (a.sample_pass || a.sample_best || a.work_base || a.work_tile || a.nq > 16) return hipErrorInvalidValue;
        if ((a.q_after_score == nullptr) != (a.q_after_id == nullptr)) return hipErrorInvalidValue;
        if (a.q_filter_mask != nullptr && a.q_filter == nullptr) return hipErrorInvalidValue;
        return ext ? launch_wide<true>(ch, a, grid, stream) : launch_wide<false>(ch, a, grid, stream);
    }
    if (a.sample_pass && a.wgs_per_group > 0) {  // the sample passes of a batch's launch groups in one launch
        if (ext || ch > 8 || a.work_tile || a.work_base || a.nq_total < 1 || grid % a.wgs_per_group != 0 ||
            grid / a.wgs_per_group != (a.nq_total + 31) / 32 || a.wgs_per_group > kMaxSampleGroups)
            return hipErrorInvalidValue;
        return launch_ch<2, kFlatSampleGroups>(ch, a, grid, stream);
    }
    if (a.sample_pass) {  // the score-floor sample of a flat scan with > 16 queries
        if (a.work_base != nullptr || a.work_tile != nullptr || a.nq <= 16) return hipErrorInvalidValue;
        if ((a.q_after_score == nullptr) != (a.q_after_id == nullptr)) return hipErrorInvalidValue;
        if (a.q_filter_mask != nullptr && a.q_filter == nullptr) return hipErrorInvalidValue;
        return ext ? launch_ch<2, kFlatSample, true>(ch, a, grid, stream) : launch_ch<2, kFlatSample>(ch, a, grid, stream);
    }
    if (ext && a.work_base != nullptr) {  // cross-index batch with masked filters
        if (!a.work_tile || !a.work_rows || !a.work_mask || !a.n_work || !a.work_tags) return hipErrorInvalidValue;
        if (a.q_after_score || a.q_after_id) return hipErrorInvalidValue;
        if (a.q_filter_mask != nullptr && a.q_filter == nullptr) return hipErrorInvalidValue;
        if (a.nq <= 16) return launch_ch<1, kMulti, true>(ch, a, grid, stream);
        return launch_ch<2, kMulti, true>(ch, a, grid, stream);
    }
    if (ext && a.work_tile != nullptr) {  // IVF probe with masked filters (no continuation bound: ids are slab positions)
        if (!a.work_rows || !a.work_mask || !a.n_work) return hipErrorInvalidValue;
        if (a.q_after_score || a.q_after_id) return hipErrorInvalidValue;
        if (a.q_filter_mask != nullptr && a.q_filter == nullptr) return hipErrorInvalidValue;
        if (a.nq <= 16) return launch_ch<1, kIvf, true>(ch, a, grid, stream);
        return launch_ch<2, kIvf, true>(ch, a, grid, stream);
    }
    if (ext) {  // masked filters / continuation bound of a flat scan
        if ((a.q_after_score == nullptr) != (a.q_after_id == nullptr)) return hipErrorInvalidValue;
        if (a.q_filter_mask != nullptr && a.q_filter == nullptr) return hipErrorInvalidValue;
        if (a.nq <= 16) return launch_ch<1, kFlat, true>(ch, a, grid, stream);
        return launch_ch<2, kFlat, true>(ch, a, grid, stream);
    }
    if (a.work_base != nullptr) {  // cross-index batch: every item names its slab
        if (!a.work_tile || !a.work_rows || !a.work_mask || !a.n_work || !a.work_tags) return hipErrorInvalidValue;
        if (a.nq <= 16) return launch_ch<1, kMulti>(ch, a, grid, stream);
        return launch_ch<2, kMulti>(ch, a, grid, stream);
    }
    if (a.work_tile != nullptr) {  // IVF probe: iterate the plan instead of every tile
        if (!a.work_rows || !a.work_mask || !a.n_work) return hipErrorInvalidValue;
        if (a.nq <= 16) return launch_ch<1, kIvf>(ch, a, grid, stream);
        return launch_ch<2, kIvf>(ch, a, grid, stream);
    }
    if (a.wgs_per_group > 0) {  // several launch groups of 32 (padded) queries over one slab, wgs_per_group workgroups each
        if (a.nq != 32 || grid % a.wgs_per_group != 0 || a.sample_best || a.xcd_skew) return hipErrorInvalidValue;
        return launch_ch<2, kFlatGroups>(ch, a, grid, stream);
    }
    if (a.nq <= 16) return launch_ch<1, kFlat>(ch, a, grid, stream);
    return launch_ch<2, kFlat>(ch, a, grid, stream);
}

}  // namespace rass

#ifdef RASS_SCAN_DIAG
// read-and-clear the launch counters (diagnostic builds only)
extern "C" int rassdiag_read(unsigned long long* out, int n) {
    unsigned long long h[8] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(rass::g_scan_diag), sizeof(h)) != hipSuccess) return -2;
    for (int i = 0; i < n && i < 8; ++i) out[i] = h[i];
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rass::g_scan_diag), z, sizeof(z)) != hipSuccess) return -3;
    return 0;
}
#endif
