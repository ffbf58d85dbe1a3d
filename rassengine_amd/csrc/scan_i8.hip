// scan_i8.hip — SURVEY §8f-4 ("bf16 (or int8)"): int8 candidate scan for the prefilter mode of a flat fp32 index.
// NOT the parity path: the fp32 fused scan (scan_topk.hip) stays the default; this is mode 2 of rass_index_set_prefilter.
//
//   1. quantize_tile16_i8_kernel: every row x of the fp32 tile16 slab becomes 1 byte per component,
//          scale = max|x| / 127,   q[c] = rint(x[c] * (127 / max|x|))   (round half to even; a zero row: q = 0, scale = 0)
//      in a slab of its own ("tile16i", below) plus one fp32 scale per row.  Queries are quantised the same way.
//   2. scan_i8_topk_kernel: the persistent, K-split, register-streaming structure of the bf16 candidate scan over that
//      slab (a QUARTER of the fp32 bytes per pass) with v_mfma_i32_16x16x64_i8: a row's candidate score is
//          (float)(sum_c q_row[c] * q_query[c]) * scale_row
//      — the integer sum is exact in int32 (|sum| <= 2048 * 127^2 < 2^26; up to dim 1 024 it is also below 2^24, so the
//      conversion is exact too; past that it is ONE round-to-nearest-even of an exact integer), so the score is a fixed
//      sequence of correctly rounded fp32 operations on exact operands and the oracle's restatement (numpy integer dot
//      products, the same conversion and multiply) reproduces it BIT FOR BIT; the query's own scale is the same for every
//      row and is left out.  Keeps the 32 best candidates per query and workgroup.
//   3. merge (merge_topk.hip) -> 32 candidates per query; rerank_f32_kernel (scan_bf16.hip) -> their exact fp32 scores in the
//      flat kernel's order and the exact (score desc, id asc) top-k among them.
// The result equals the flat result whenever the true top-k is inside the int8 top-32 — measured (bench.py --prefilter int8:
// recall vs the flat kernel), never assumed.
//
// int8 slab layout ("tile16i"): 16-row blocks of `stride_i8` bytes per row (the fp32 stride rounded up to 512, zero padded; up to
// 2 048 = the widest row an index takes);
// chunk jb (columns 64jb .. 64jb+63) of the 16 rows is one contiguous 1 KiB in MFMA lane order, lane (m = lane&15,
// g = lane>>4) holding X[16b+m][64jb + 16g .. +15]: element (r, c) at
//   (r>>4)*16*stride_i8 + (c>>6)*1024 + (((c>>4)&3)*16 + (r&15))*16 + (c&15)      [bytes]

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "scan_core.h"

namespace rass {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kIWaves = 8;
constexpr int kIThreads = kIWaves * 64;
constexpr int kITileRows = 64;  // 4 blocks of 16 rows
constexpr int kIPitch = 68;     // ints per query row of the LDS partial image (64 rows + pad)

struct IDesc {
    __amdgpu_buffer_rsrc_t rows;
    __amdgpu_buffer_rsrc_t tags;
    __amdgpu_buffer_rsrc_t scales;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(uint64_t addr, unsigned bytes) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)nb, 0x00020000);
}

// tile t of the slab: rows past n_rows read as zero bytes / scale 0 / tag 0 (and are masked by `rows`); a tile past the end of
// the slab has zero rows and a zero-sized descriptor on tile 0
__device__ __forceinline__ IDesc make_idesc(const ScanI8Args& p, int tile, int rows) {
    const int64_t base_row = rows > 0 ? (int64_t)tile * kITileRows : 0;
    IDesc d;
    const unsigned blocks = (unsigned)(rows + 15) >> 4;
    d.rows = make_rsrc(reinterpret_cast<uint64_t>(p.corpus + base_row * p.row_stride), blocks * 16u * (unsigned)p.row_stride);
    d.scales = make_rsrc(reinterpret_cast<uint64_t>(p.row_scale + base_row), (unsigned)(rows * 4));
    const bool has = p.row_tag != nullptr;
    d.tags = make_rsrc(reinterpret_cast<uint64_t>(has ? (const void*)(p.row_tag + base_row) : (const void*)p.corpus),
                       has ? (unsigned)(rows * 4) : 0u);
    return d;
}

template <int CHI>
struct ITile {
    i32x4 a[4][CHI];
    int tag;      // tag of row lane (0..63) of the tile
    float scale;  // scale of that row
};

#ifndef RASS_I8_NBUF
#define RASS_I8_NBUF 2   // LDS images of the partial sums: 1 = two workgroups per CU (a second barrier per tile), 2 = one
#endif
constexpr int kINBuf = RASS_I8_NBUF;

// IVF = true: the work items are the entries of a probe plan (64-row tiles of the slab with per-tile query masks, written by
// plan_probe_kernel earlier on this stream) instead of every tile of the slab; no sample floor.
template <int CHI, int NT, bool IVF>
__global__ __launch_bounds__(kIThreads, kINBuf == 1 ? 4 : 2) void scan_i8_topk_kernel(ScanI8Args p) {
    constexpr int NQ = NT * 16;
    extern __shared__ __attribute__((aligned(16))) int lds_i[];  // [kINBuf][kIWaves][NQ][kIPitch]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int n_tiles = IVF ? __builtin_amdgcn_readfirstlane(*p.n_work) : (p.n_rows + kITileRows - 1) / kITileRows;
    // a grouped launch (flat only): this workgroup's group, its place in the group, the group's queries / filters / lists
    const int G = (!IVF && p.wgs_per_group > 0) ? p.wgs_per_group : (int)gridDim.x;
    const int grp = (!IVF && p.wgs_per_group > 0) ? (int)blockIdx.x / G : 0;
    const int bid = (int)blockIdx.x - grp * G;
    p.q_i8 += (int64_t)grp * p.q_group_stride;
    if (p.q_filter != nullptr) p.q_filter += grp * 32;
    if (p.q_filter_mask != nullptr) p.q_filter_mask += grp * 32;
    p.part_scores += (int64_t)grp * p.part_group_stride;
    if (p.part_ids != nullptr) p.part_ids += (int64_t)grp * p.part_group_stride;

    i32x4 qf[NT][CHI];
    {
        const signed char* qb = p.q_i8 + (int64_t)m * p.row_stride + wid * 64 * CHI + 16 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < CHI; ++j)
                qf[nt][j] = *reinterpret_cast<const i32x4*>(qb + (int64_t)nt * 16 * p.row_stride + 64 * j);
    }
    const int voff_lane = wid * CHI * 1024 + lane * 16;
    const int blk_step = 16 * (int)p.row_stride;

    TopList L[NT];
    float tau[NT];
    int qfilt[NT], qmask[NT];   // a row matches when (tag & qmask) == qfilt (qmask = -1: the exact compare); qfilt < 0: no filter
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) {
        L[pq].s = -INFINITY;
        L[pq].i = 0x7fffffff;
        tau[pq] = -INFINITY;
        const int q = pq * 16 + (lane >> 5) * 8 + wid;
        qfilt[pq] = (p.q_filter != nullptr && q < p.nq) ? p.q_filter[q] : -1;
        qmask[pq] = (p.q_filter_mask != nullptr && q < p.nq) ? p.q_filter_mask[q] : -1;
    }

    // The sample floor (as in the fp32 scan, scan_topk.hip): sample_best[g][q] is the best candidate score workgroup g of a
    // SAMPLE launch (this kernel over the slab's first 64 * sample_groups rows, k = 1) found for query q under q's filter.
    // Those workgroups scanned disjoint rows, so the k-th largest of them is reached by k rows of the slab, hence by the
    // final k-th best: rows scoring below it are dropped before the sorted insertion (rows tying with it are kept) — the
    // lists the merge sees lose only entries that could not have ranked.  With 32 candidates kept per query the insertion
    // was the kernel's critical path (~185 insertions per query and workgroup over 61 tiles; ~8 with the floor).
    float floor_q[NT];
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) floor_q[pq] = -INFINITY;
    if (!IVF && p.sample_best != nullptr) {
        constexpr int kSlots = kMaxSampleGroups / 64;
        unsigned key[NT][2][kSlots];
#pragma unroll
        for (int pq = 0; pq < NT; ++pq)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int j = 0; j < kSlots; ++j) {
                    const int grp = lane + 64 * j, q = pq * 16 + half * 8 + wid;
                    const float v = (grp < p.sample_groups && q < p.nq) ? p.sample_best[(int64_t)grp * p.nq + q] : -INFINITY;
                    key[pq][half][j] = v == -INFINITY ? 0u : score_key(v);
                }
        constexpr int kFloorBits = 20;   // the truncation only lowers the floor
        unsigned T[NT][2] = {};
#pragma unroll 1
        for (int b = 31; b >= 32 - kFloorBits; --b) {
#pragma unroll
            for (int pq = 0; pq < NT; ++pq)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const unsigned cand = T[pq][half] | (1u << b);
                    int c = 0;
#pragma unroll
                    for (int j = 0; j < kSlots; ++j) c += __popcll(__ballot(key[pq][half][j] >= cand));
                    T[pq][half] = c >= p.k ? cand : T[pq][half];
                }
        }
#pragma unroll
        for (int pq = 0; pq < NT; ++pq) {
            const unsigned t = (lane & 32) ? T[pq][1] : T[pq][0];
            floor_q[pq] = t ? key_score(t) : -INFINITY;
        }
    }

    // item t of the launch: (slab tile, its valid rows, the queries that may rank them); past the end: zero rows of tile 0
    struct Item { int tile, rows; unsigned mask; };
    auto item = [&](int t) {
        Item w;
        const bool ok = t < n_tiles;
        if (IVF) {
            w.tile = ok ? p.work_tile[t] : 0;
            w.rows = ok ? p.work_rows[t] : 0;
            w.mask = ok ? p.work_mask[t] : 0u;
        } else {
            const int rows = p.n_rows - t * kITileRows;
            w.tile = ok ? t : 0;
            w.rows = ok ? (rows > kITileRows ? kITileRows : rows) : 0;
            w.mask = 0xffffffffu;
        }
        return w;
    };
    auto issue = [&](ITile<CHI>& r, const IDesc& d) {
        r.tag = (int)__builtin_amdgcn_raw_buffer_load_b32(d.tags, lane * 4, 0, 0);
        r.scale = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d.scales, lane * 4, 0, 0));
#pragma unroll
        for (int j = 0; j < CHI; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                r.a[b][j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(d.rows, voff_lane + b * blk_step + j * 1024, 0, 2));
    };
    auto mul_refill = [&](ITile<CHI>& r, i32x4 (&acc)[4][NT], const IDesc& next) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[b][nt] = i32x4{0, 0, 0, 0};
        r.tag = (int)__builtin_amdgcn_raw_buffer_load_b32(next.tags, lane * 4, 0, 0);
        r.scale = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(next.scales, lane * 4, 0, 0));
#pragma unroll
        for (int j = 0; j < CHI; ++j) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const i32x4 a = r.a[b][j];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[b][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, qf[nt][j], acc[b][nt], 0, 0, 0);
                r.a[b][j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(next.rows, voff_lane + b * blk_step + j * 1024, 0, 2));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto finish = [&](const i32x4 (&acc)[4][NT], const Item& w, int tag, float scale, int buf) {
        int* P = lds_i + (kINBuf == 2 ? buf : 0) * (kIWaves * NQ * kIPitch);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *reinterpret_cast<i32x4*>(P + (wid * NQ + nt * 16 + m) * kIPitch + b * 16 + 4 * g) = acc[b][nt];
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = (lane & 31) + 32 * half;
            const int row = w.tile * kITileRows + r;
            const int rtag = __shfl(tag, r, 64);
            const float rscale = __shfl(scale, r, 64);
            const bool row_ok = (r < w.rows) && (rtag != -1);
#pragma unroll
            for (int pq = 0; pq < NT; ++pq) {
                const int q = pq * 16 + (lane >> 5) * 8 + wid;
                const int* src = P + q * kIPitch + r;
                int s = src[0];
#pragma unroll
                for (int wv = 1; wv < kIWaves; ++wv) s += src[wv * NQ * kIPitch];
                const float sf = (float)s * rscale;
                bool ok = row_ok && (qfilt[pq] < 0 || qfilt[pq] == (rtag & qmask[pq])) && (sf >= floor_q[pq]);
                if (IVF) ok = ok && ((w.mask >> q) & 1u) != 0;   // only the queries that probe this tile's list
                insert_candidates_auto(L[pq], tau[pq], ok ? sf : -INFINITY, row, p.k);
            }
        }
        if (kINBuf == 1) __syncthreads();   // the image is rewritten by the next tile
    };

    ITile<CHI> R0, R1;
    int t = bid;
    // work items are fetched a whole iteration before their descriptors are built: read right before use (as the first version
    // did) every tile pair waited for two dependent scalar loads from global memory (an IVF's work list)
    Item w0 = item(t), w1 = item(t + G), w2 = item(t + 2 * G), w3 = item(t + 3 * G);
    issue(R0, make_idesc(p, w0.tile, w0.rows));
    issue(R1, make_idesc(p, w1.tile, w1.rows));
    __builtin_amdgcn_sched_barrier(0);
    for (; t < n_tiles; t += 2 * G) {
        const Item w4 = item(t + 4 * G), w5 = item(t + 5 * G);
        i32x4 acc[4][NT];
        int tag = R0.tag;
        float scale = R0.scale;
        mul_refill(R0, acc, make_idesc(p, w2.tile, w2.rows));
        finish(acc, w0, tag, scale, 0);
        tag = R1.tag;
        scale = R1.scale;
        mul_refill(R1, acc, make_idesc(p, w3.tile, w3.rows));
        finish(acc, w1, tag, scale, 1);
        w0 = w2;
        w1 = w3;
        w2 = w4;
        w3 = w5;
    }
    const int lpos = lane & 31;
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) {
        const int q = pq * 16 + (lane >> 5) * 8 + wid;
        if (q < p.nq && lpos < p.k) {
            const int64_t o = ((int64_t)bid * p.nq + q) * p.k + lpos;
            const bool filled = L[pq].i != 0x7fffffff;
            p.part_scores[o] = filled ? L[pq].s : -INFINITY;
            if (p.part_ids) p.part_ids[o] = filled ? (int64_t)L[pq].i : (int64_t)-1;   // LOCAL rows (nullptr: a sample launch)
        }
    }
}

template <int CHI, int NT, bool IVF>
static hipError_t launch_ivariant(const ScanI8Args& a, int grid, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)kINBuf * kIWaves * NT * 16 * kIPitch * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_i8_topk_kernel<CHI, NT, IVF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((scan_i8_topk_kernel<CHI, NT, IVF>), dim3(grid), dim3(kIThreads), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_scan_i8_topk(const ScanI8Args& a, int grid, hipStream_t stream) {
    if (a.row_stride % 512 != 0 || a.row_stride < 512 || a.row_stride > 2048) return hipErrorInvalidValue;  // 8 waves x 64-column chunks
    if (a.nq < 1 || a.nq > 32 || a.k < 1 || a.k > 32 || a.n_rows < 0 || grid < 1) return hipErrorInvalidValue;
    if (a.sample_best && (a.sample_groups < 1 || a.sample_groups > kMaxSampleGroups)) return hipErrorInvalidValue;
    if (a.q_filter_mask && !a.q_filter) return hipErrorInvalidValue;
    const bool ivf = a.work_tile != nullptr;
    if (ivf && (!a.work_rows || !a.work_mask || !a.n_work || a.sample_best || a.wgs_per_group)) return hipErrorInvalidValue;
    if (a.wgs_per_group > 0 && (grid % a.wgs_per_group != 0 || a.sample_best)) return hipErrorInvalidValue;
    const bool two = a.nq > 16;
#define RASS_I8_CASE(C)                                                                                                     \
    if (ivf) return two ? launch_ivariant<C, 2, true>(a, grid, stream) : launch_ivariant<C, 1, true>(a, grid, stream);       \
    return two ? launch_ivariant<C, 2, false>(a, grid, stream) : launch_ivariant<C, 1, false>(a, grid, stream);
    if (a.row_stride == 512) { RASS_I8_CASE(1) }
    if (a.row_stride == 1024) { RASS_I8_CASE(2) }
    if (a.row_stride == 1536) { RASS_I8_CASE(3) }   // wide rows (1 024 < dim <= 2 048)
    RASS_I8_CASE(4)
#undef RASS_I8_CASE
}

// fp32 tile16 blocks [b0, b1) -> tile16i blocks + one scale per row; one wave per 16-row block.  Lane (m, g) owns, of row m,
// the 16 columns 64jb + 16g .. +15 of every 64-column chunk jb = fp32 chunk 4jb + g (four f32x4 at lane groups 0..3).
__global__ __launch_bounds__(256) void quantize_tile16_i8_kernel(const float* __restrict__ src, signed char* __restrict__ dst,
                                                                 float* __restrict__ scale, int64_t stride, int64_t stride_i8,
                                                                 int64_t b0, int64_t b1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nch = (int)(stride >> 6);   // 64-column chunks that hold data (stride is a multiple of 128)
    for (int64_t b = b0 + (int64_t)blockIdx.x * 4 + wave; b < b1; b += (int64_t)gridDim.x * 4) {
        const float* sb = src + b * 16 * stride;
        signed char* db = dst + b * 16 * stride_i8;
        float mx = 0.f;
        for (int jb = 0; jb < nch; ++jb) {
            const float* sc = sb + (int64_t)(4 * jb + g) * 256;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sc + (gg * 16 + m) * 4);
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float inv = mx > 0.f ? 127.f / mx : 0.f;
        if (g == 0) scale[b * 16 + m] = mx / 127.f;
        for (int jb = 0; jb < nch; ++jb) {
            const float* sc = sb + (int64_t)(4 * jb + g) * 256;
            unsigned o[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sc + (gg * 16 + m) * 4);
                const int q0 = (int)rintf(v.x * inv), q1 = (int)rintf(v.y * inv), q2 = (int)rintf(v.z * inv), q3 = (int)rintf(v.w * inv);
                o[gg] = (unsigned)(q0 & 0xff) | ((unsigned)(q1 & 0xff) << 8) | ((unsigned)(q2 & 0xff) << 16) | ((unsigned)(q3 & 0xff) << 24);
            }
            *reinterpret_cast<uint4*>(db + (int64_t)jb * 1024 + lane * 16) = uint4{o[0], o[1], o[2], o[3]};
        }
    }
}

hipError_t launch_quantize_tile16_i8(const float* src, void* dst, float* scale, int64_t stride, int64_t stride_i8, int64_t block0,
                                     int64_t block1, hipStream_t stream) {
    if (block1 <= block0) return hipSuccess;
    if (stride % 128 != 0 || stride_i8 % 512 != 0 || stride_i8 < stride) return hipErrorInvalidValue;
    int64_t blocks = (block1 - block0 + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(quantize_tile16_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, static_cast<signed char*>(dst),
                       scale, stride, stride_i8, block0, block1);
    return hipGetLastError();
}

// queries: normalised fp32 [nq_pad][stride] (q_padded of the fp32 path, zero rows past nq) -> int8 [nq_pad][stride_i8] row-major,
// quantised per query like a corpus row (its scale is not needed: it does not change a query's ranking).  One wave per query.
__global__ __launch_bounds__(64) void queries_to_i8_kernel(const float* __restrict__ src, signed char* __restrict__ dst, int64_t stride,
                                                           int64_t stride_i8) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const float* s = src + (int64_t)q * stride;
    float mx = 0.f;
    for (int64_t c = lane; c < stride; c += 64) mx = fmaxf(mx, fabsf(s[c]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float inv = mx > 0.f ? 127.f / mx : 0.f;
    for (int64_t c = lane; c < stride_i8; c += 64) dst[(int64_t)q * stride_i8 + c] = c < stride ? (signed char)(int)rintf(s[c] * inv) : (signed char)0;
}

hipError_t launch_queries_to_i8(const float* src, void* dst, int nq_pad, int64_t stride, int64_t stride_i8, hipStream_t stream) {
    if (nq_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(queries_to_i8_kernel, dim3(nq_pad), dim3(64), 0, stream, src, static_cast<signed char*>(dst), stride, stride_i8);
    return hipGetLastError();
}

}  // namespace rass
