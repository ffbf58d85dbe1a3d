// scan_core.h — device-side building blocks shared by the kernels that stream a tile16 fp32 slab through
// v_mfma_f32_16x16x4_f32 with the K axis split over 8 waves: the fused scan + top-k (scan_topk.hip) and the
// k-means assignment of the IVF build (kmeans.hip).  Not part of any ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rass {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kTileRows = 32;
constexpr int kPitch = 36;  // floats per query row of the LDS partial image (32 rows + pad)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// One tile's A fragments for one wave: 2 M-tiles x CH chunks of 16 B per lane.
template <int CH>
struct TileRegs {
    f32x4 a[2][CH];
    int tag;
};

// Wave-uniform buffer descriptors of one tile: corpus rows and row tags.  The range check
// drops every lane past the end (rows beyond n_rows, or a whole run-ahead tile past the last
// one): zeros come back and no memory request is made.
struct TileDesc {
    __amdgpu_buffer_rsrc_t rows;
    __amdgpu_buffer_rsrc_t tags;
};

// One unit of work of a workgroup: a 32-row slab tile, how many of its rows exist, and which
// queries of the batch may rank its rows (all of them for the flat scan; for IVF the queries
// that probe the list the tile belongs to).
struct WorkItem {
    int tile;
    int rows;
    unsigned mask;
};

// Which work a workgroup iterates over: every tile of one slab (flat), the tiles of a device-built probe plan
// over one slab (IVF), or the tiles of SEVERAL slabs — one per index of a cross-index query batch — each item
// naming its own slab and tag array (MULTI: concurrent users' per-user indices share one launch).
// kFlatSample is kFlat under its own kernel name: the sample pass that finds the score floors (ScanArgs::floors),
// kept apart so that per-kernel profiles do not average the short sample launches into the scan's.
// kFlatGroups: ONE launch scans the same slab for several launch groups of <= 32 queries (ScanArgs::wgs_per_group
// workgroups each): the coarse stage of an IVF batch (4 096 centroids for 1 024 queries in one launch instead of 32).
// kIvfGroups: ONE launch walks the probe plans of several launch groups (each its own work list, item count, queries and
// lists): the fine stage of an IVF batch without a launch boundary — ramp-up, tail, gap — between the groups.
// kFlatSampleGroups: the sample passes of several launch groups in one launch (a batch call's 32 passes of 20 us each).
enum ScanMode { kFlat = 0, kIvf = 1, kMulti = 2, kFlatSample = 3, kFlatGroups = 4, kIvfGroups = 5, kFlatSampleGroups = 6 };
constexpr bool mode_is_flat(int mode) {
    return mode == kFlat || mode == kFlatSample || mode == kFlatGroups || mode == kFlatSampleGroups;
}
constexpr bool mode_is_sample(int mode) { return mode == kFlatSample || mode == kFlatSampleGroups; }

__device__ __forceinline__ TileDesc make_tile_desc(const float* __restrict__ X, int64_t row_stride,
                                                   const int32_t* __restrict__ row_tag, const WorkItem& w) {
    const int rows_here = w.rows;
    const int64_t base_row = (int64_t)w.tile * kTileRows;
    // The descriptor must be PROVABLY wave-uniform or hipcc wraps every buffer op in a
    // waterfall loop: pin its inputs with readfirstlane (guide T20).
    const uint64_t base_u = reinterpret_cast<uint64_t>(X + base_row * row_stride);
    const uint32_t base_lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
    const uint32_t base_hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
    const unsigned blocks_here = (unsigned)(rows_here + 15) >> 4;  // tile16: whole 16-row blocks
    const unsigned bytes = __builtin_amdgcn_readfirstlane(blocks_here * 16u * (unsigned)row_stride * 4u);
    TileDesc d;
    d.rows = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)base_hi << 32) | base_lo),
                                               /*stride*/ 0, (int)bytes, 0x00020000);
    // Row tags: always loaded (straight-line code keeps hipcc's vmcnt counts exact); with no
    // tag array the descriptor has zero records: the load returns 0 and costs no traffic.
    const bool has_tags = row_tag != nullptr;
    const uint64_t tbase_u = has_tags ? reinterpret_cast<uint64_t>(row_tag + base_row) : base_u;
    const uint32_t tlo = __builtin_amdgcn_readfirstlane((uint32_t)tbase_u);
    const uint32_t thi = __builtin_amdgcn_readfirstlane((uint32_t)(tbase_u >> 32));
    const unsigned tbytes = __builtin_amdgcn_readfirstlane(has_tags ? (unsigned)(rows_here * 4) : 0u);
    d.tags = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<int32_t*>(((uint64_t)thi << 32) | tlo), 0,
                                               (int)tbytes, 0x00020000);
    return d;
}

// voff: this lane's byte offset inside a chunk (ONE VGPR for every load of the kernel); soff: the wave-uniform
// part (M-tile and chunk), which goes into the instruction's SGPR / immediate offset fields.  Folding it into
// per-load VGPR offsets, as the first version did, cost ~10 VGPRs the main loop does not have.
__device__ __forceinline__ f32x4 load_chunk(const TileDesc& d, int voff, int soff) {
#ifndef RASS_SCAN_AUX      // cache policy of the corpus stream: 2 = nt (read once); the micro-benchmarks sweep it (-DRASS_SCAN_AUX=<bits>)
#define RASS_SCAN_AUX 2
#endif
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.rows, voff, soff, RASS_SCAN_AUX));
}

__device__ __forceinline__ int load_tag(const TileDesc& d) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(d.tags, (lane_id() & 31) * 4, 0, 0);
}

// Prologue: all of a tile's loads, in the order multiply_and_refill consumes them.  `soff0`: byte offset of the K
// panel these registers hold (0 unless the wave's K slice is walked in two panels: the wide-row kernel, dim > 1024).
template <int CH, bool TAGS = true>
__device__ __forceinline__ void issue_tile_loads(TileRegs<CH>& r, const TileDesc& d, int voff_lane, int mt_step,
                                                 int soff0 = 0) {
    if (TAGS) r.tag = load_tag(d);
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) r.a[mt][j] = load_chunk(d, voff_lane, soff0 + mt * mt_step + j * 1024);
}

// Multiply the resident tile and, chunk by chunk, re-issue each consumed register's load
// for the tile two steps ahead (`next`): about two tiles (32 KiB per wave, 256 KiB per CU)
// stay in flight at all times instead of one tile requested in a burst after the previous
// one has fully arrived.  sched_barrier(0) per chunk pins the load placement (hipcc would
// otherwise sink the loads behind the whole MFMA block).
//
// `between(j)` runs after chunk j's MFMAs were issued: the ranking of the PREVIOUS tile pair is cut into
// parts and placed there, so its VALU / LDS / scalar work issues while this wave's (and its SIMD partner's)
// MFMAs execute on the matrix pipe (a v_mfma_f32_16x16x4_f32 occupies the pipe for 32 cycles but takes only a
// few to issue).  Ranked in a phase of its own behind the barrier — as the first version did — it left the
// matrix pipe idle while all 8 waves ranked in lockstep: 72 us of a 690 us launch at B = 32.
//
// ZERO = false continues the accumulators of the previous call (the second K panel of a wide row: the fmaf chain of a
// wave's slice runs on in k order); `soff0` is the panel's byte offset for the refill loads.
template <int CH, int NT, bool TAGS = true, bool ZERO = true, typename Between>
__device__ __forceinline__ void multiply_and_refill(TileRegs<CH>& r, const f32x4 (&qf)[NT][CH],
                                                    f32x4 (&acc)[2][NT], const TileDesc& next, int voff_lane,
                                                    int mt_step, Between&& between, int soff0 = 0) {
    if (ZERO) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (TAGS) r.tag = load_tag(next);  // the k-means kernel streams centroids: no row tags
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        // The 2 x NT accumulators are advanced ROUND-ROBIN, one k-step at a time: a dependent
        // v_mfma_f32_16x16x4_f32 can issue 40 cycles after its producer but the pipe takes a new one every 32, so
        // back-to-back MFMAs on one accumulator (what hipcc emitted for the second M-tile when left to itself:
        // two chains of four) leave the pipe idle 20 % of the time whenever the SIMD's other wave is not in its
        // MFMA phase.  With >= 2 independent accumulators between a producer and its consumer there is no bubble.
        const f32x4 a0 = r.a[0][j], a1 = r.a[1][j];
#define RASS_KSTEP(comp)                                                                                              \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                               \
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.comp, qf[nt][j].comp, acc[0][nt], 0, 0, 0);              \
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.comp, qf[nt][j].comp, acc[1][nt], 0, 0, 0);              \
    }
        RASS_KSTEP(x)
        RASS_KSTEP(y)
        RASS_KSTEP(z)
        RASS_KSTEP(w)
#undef RASS_KSTEP
        r.a[0][j] = load_chunk(next, voff_lane, soff0 + j * 1024);
        r.a[1][j] = load_chunk(next, voff_lane, soff0 + mt_step + j * 1024);
        __builtin_amdgcn_sched_barrier(0);
        between(j);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Order-preserving u32 key of a finite float score (larger score <-> larger key; every finite score's key is
// >= 0x00800000, so 0 can stand for "no score"): the sample floor's selection counts keys by ballot.
__device__ __forceinline__ unsigned score_key(float s) {
    const unsigned u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// Half-wave sorted list: lanes 0..31 hold query A's best-first top-32, lanes 32..63 query B's.
struct TopList {
    float s;
    int i;
};

// The k-th best score of my half (lanes 0..31 / 32..63): two readlanes + a select, no LDS permute.
__device__ __forceinline__ float bcast_kth(float v, int k) {
    const int a = __builtin_amdgcn_readlane(__float_as_int(v), k - 1);
    const int b = __builtin_amdgcn_readlane(__float_as_int(v), 32 + k - 1);
    return __int_as_float((lane_id() & 32) ? b : a);
}

// Lane i <- lane i-1 across the whole wave (lane 0 keeps its value): one DPP move (wave_shr:1).
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }

// Insert every lane's candidate (cs valid where cs > tau of its half) into the half-wave lists.
// Everything stays in registers / the scalar unit: the first version used __shfl_up / __shfl here, which
// hipcc lowers to ds_bpermute_b32 — three LDS round trips (~400 cycles) per inserted candidate.  A per-workgroup
// list takes ~k ln(n/k) (60 at k = 10, 3 900 rows) insertions per launch and the 8 waves meet at a barrier every
// two tiles, so the slowest wave's insertions were on every workgroup's critical path: with the ranking
// removed the B = 32 kernel ran 72 us (10 %) faster, with only the insertion removed 57 us
// (profiles/r02_scan_phase_experiments.txt).
__device__ __forceinline__ void insert_candidates(TopList& L, float& tau, float s, int row, int k) {
    unsigned long long mask = __ballot(s > tau);
    const int lane = lane_id();
    const int lpos = lane & 31;
    while (mask) {
        const int c = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float cs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), c));
        const int ci = __builtin_amdgcn_readlane(row, c);
        // straight-line body (bitwise predicates, selects): the short-circuit / nested-if form compiled to five
        // exec-mask branches per candidate
        const bool mine = ((lane ^ c) & 32) == 0;
        const bool better = (L.s > cs) | ((L.s == cs) & (L.i < ci));
        const int pos = __builtin_popcountll(__ballot(better & mine));
        const float us = __int_as_float(wave_shr1(__float_as_int(L.s)));
        const int ui = wave_shr1(L.i);
        const bool live = mine & (pos < k);          // a candidate that ranks behind the k kept changes nothing
        const bool take = live & (lpos == pos);
        const bool shift = live & (lpos > pos);
        L.s = take ? cs : (shift ? us : L.s);
        L.i = take ? ci : (shift ? ui : L.i);
        tau = bcast_kth(L.s, k);
        // candidates the raised threshold has just ruled out leave the queue (they would rank behind the k kept: same lists).
        // A workgroup's FIRST tile meets empty lists — all 64 lanes queue up — and on a small slab (an IVF's coarse scan, a
        // fine scan at nprobe 1, a 10 k-row index) that tile is the whole launch: 64 serial steps per part cost 5.5 us of a
        // 15 us launch however small k was (r03, scripts/microbench/scan_small.hip).
        mask &= __ballot(s > tau);
    }
}

// ---- bulk insertion: MANY candidates of one call at once (round 4) ------------------------------------------------------
// insert_candidates takes one candidate per step, ~200 cycles each, serially over the wave.  That is the right shape when a
// tile yields a handful of candidates per query (the flat scan behind a sample floor).  It is the wrong one where a tile is
// DENSE in candidates: an IVF's fine scan (a tile belongs to a list one or two of the batch's queries probe, nearly all of its
// 64 rows beat that workgroup's short list, and the one wave that ranks that query works through them while seven wait at
// the barrier: 6 us per 64-KiB tile of an int8 slab, scan_i8.hip), or any workgroup's first tile.  Here the half-wave's 32
// candidates are SORTED (worst first: a 15-stage bitonic network on DPP / v_permlane16_swap, both halves = both queries of
// the wave at once), compared lane by lane with the running list (best first: list ++ candidates is a bitonic sequence and the
// lane-wise winner its best 32) and re-sorted by the 5-stage bitonic merge: ~21 compare-exchange stages whatever the number of
// candidates — the same list the one-by-one insertion ends with ((score desc, row asc) is a total order).
template <int STRIDE>
__device__ __forceinline__ int half_xor_lane(int v, int lane) {   // the dword of lane ^ STRIDE (STRIDE <= 16: inside a half-wave)
    static_assert(STRIDE == 1 || STRIDE == 2 || STRIDE == 4 || STRIDE == 8 || STRIDE == 16, "stride");
    if constexpr (STRIDE == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
    if constexpr (STRIDE == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
    if constexpr (STRIDE == 4)   // i ^ 4 = half_mirror(quad_reverse(i)): (i ^ 3) ^ 7
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x1B, 0xf, 0xf, true), 0x141, 0xf, 0xf, true);
    if constexpr (STRIDE == 8)   // i ^ 8 = row_mirror(half_mirror(i)): (i ^ 7) ^ 15
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true), 0x140, 0xf, 0xf, true);
    const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 16) ? sw[0] : sw[1]);
}

template <int STRIDE>
__device__ __forceinline__ void top_cmpx(float& s, int& i, int lane, bool keep_better) {
    const float os = __int_as_float(half_xor_lane<STRIDE>(__float_as_int(s), lane));
    const int oi = half_xor_lane<STRIDE>(i, lane);
    const bool mine_better = (s > os) | ((s == os) & (i < oi));
    const bool take = mine_better != keep_better;
    s = take ? os : s;
    i = take ? oi : i;
}

template <int SIZE, int STRIDE>
__device__ __forceinline__ void top_sort_steps(float& s, int& i, int lane) {   // one level of the half-wave bitonic sort
    if constexpr (STRIDE > 0) {
        // blocks of SIZE alternate direction; the last level (SIZE == 32) runs WORST first in both halves
        const bool best_first = SIZE == 32 ? false : (lane & SIZE) == 0;
        const bool lower = (lane & STRIDE) == 0;
        top_cmpx<STRIDE>(s, i, lane, best_first == lower);
        top_sort_steps<SIZE, STRIDE / 2>(s, i, lane);
    }
}

// cs / ci: this lane's candidate, already (-inf, 0x7fffffff) where it may not rank.  Both halves (two queries) at once.
__device__ __forceinline__ void insert_candidates_bulk(TopList& L, float& tau, float cs, int ci, int k) {
    const int lane = lane_id();
    top_sort_steps<2, 1>(cs, ci, lane);
    top_sort_steps<4, 2>(cs, ci, lane);
    top_sort_steps<8, 4>(cs, ci, lane);
    top_sort_steps<16, 8>(cs, ci, lane);
    top_sort_steps<32, 16>(cs, ci, lane);
    // list (best first) ++ candidates (worst first) is bitonic: the lane-wise winners are its best 32, themselves bitonic
    const bool keep = (L.s > cs) | ((L.s == cs) & (L.i < ci));
    float s = keep ? L.s : cs;
    int i = keep ? L.i : ci;
    top_cmpx<16>(s, i, lane, (lane & 16) == 0);
    top_cmpx<8>(s, i, lane, (lane & 8) == 0);
    top_cmpx<4>(s, i, lane, (lane & 4) == 0);
    top_cmpx<2>(s, i, lane, (lane & 2) == 0);
    top_cmpx<1>(s, i, lane, (lane & 1) == 0);
    // entries behind the k kept are dropped, as the one-by-one insertion never stores them
    const bool kept = (lane & 31) < k;
    L.s = kept ? s : -INFINITY;
    L.i = kept ? i : 0x7fffffff;
    tau = bcast_kth(L.s, k);
}

// One call site for both shapes: the network when the call brings at least kBulkMin candidates, one by one below.
constexpr int kBulkMin = 8;
__device__ __forceinline__ void insert_candidates_auto(TopList& L, float& tau, float s, int row, int k) {
    const unsigned long long mask = __ballot(s > tau);
    if (__popcll(mask) >= kBulkMin) {
        const bool in = s > tau;
        insert_candidates_bulk(L, tau, in ? s : -INFINITY, in ? row : 0x7fffffff, k);
    } else if (mask) {
        insert_candidates(L, tau, s, row, k);
    }
}

}  // namespace rass
