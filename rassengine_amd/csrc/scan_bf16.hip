// scan_bf16.hip — SURVEY §8f-4: bf16 candidate scan + exact fp32 re-rank ("prefilter" mode of
// a flat index).  NOT the parity path: the fp32 fused scan (scan_topk.hip) stays the default.
//
//   1. scan_bf16_topk_kernel: the same persistent, K-split, register-streaming structure as the
//      fp32 scan, over a bf16 copy of the slab (half the HBM bytes per pass) with
//      v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate, so the pass is purely HBM-bound even at
//      32 queries); keeps the 32 best candidates per query and workgroup by bf16 score.
//   2. merge (merge_topk.hip) -> 32 candidates per query.
//   3. rerank_f32_kernel: exact fp32 scores of those candidates from the fp32 slab, in the flat
//      kernel's fmaf order (bit-identical scores for every returned row), then the exact
//      (score desc, id asc) top-k among them.
// The result equals the flat result whenever the true top-k is inside the bf16 top-32 — measured
// (recall vs the flat kernel), never assumed.
//
// bf16 slab layout ("tile16b"): 16-row blocks; chunk jb (columns 32jb..32jb+31) of the 16 rows
// is one contiguous 1 KiB in MFMA lane order, lane (m = lane&15, g = lane>>4) holding
// X[16b+m][32jb + 8g .. +7]: element (r, c) at
//   (r>>4)*16*stride + (c>>5)*512 + ((((c>>3)&3)*16 + (r&15))*8) + (c&7)      [bf16 elements]

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "kernels.h"
#include "scan_core.h"

namespace rass {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int kBWaves = 8;
constexpr int kBThreads = kBWaves * 64;
constexpr int kBTileRows = 64;  // 4 blocks of 16 rows
constexpr int kBPitch = 68;     // floats per query row of the LDS partial image (64 rows + pad)

__device__ __forceinline__ u16 f2bf_s(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<u16*>(&h);
}

struct BDesc {
    __amdgpu_buffer_rsrc_t rows;
    __amdgpu_buffer_rsrc_t tags;
};

// a work item: 64-row tile `tile` of the slab with `rows` valid rows, visible to the queries whose bit is set in `mask`
struct BWork {
    int tile, rows;
    unsigned mask;
};

// item i of the launch: tile i of the slab (flat) or entry i of the probe plan (IVF); past the end: zero rows of tile 0
template <bool IVF>
__device__ __forceinline__ BWork get_bwork(const ScanBf16Args& p, int i, int n_items) {
    BWork w;
    const bool ok = i < n_items;
    if (IVF) {
        w.tile = ok ? p.work_tile[i] : 0;
        w.rows = ok ? p.work_rows[i] : 0;
        w.mask = ok ? p.work_mask[i] : 0u;
    } else {
        int rows = p.n_rows - i * kBTileRows;
        rows = rows < 0 ? 0 : (rows > kBTileRows ? kBTileRows : rows);
        w.tile = ok ? i : 0;
        w.rows = ok ? rows : 0;
        w.mask = 0xffffffffu;
    }
    return w;
}

__device__ __forceinline__ BDesc make_bdesc(const u16* __restrict__ X, int64_t stride, const int32_t* __restrict__ tag,
                                            const BWork& w) {
    const int rows_here = w.rows;
    const int64_t base_row = (int64_t)w.tile * kBTileRows;
    const uint64_t bu = reinterpret_cast<uint64_t>(X + base_row * stride);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bu), hi = __builtin_amdgcn_readfirstlane((uint32_t)(bu >> 32));
    const unsigned blocks = (unsigned)(rows_here + 15) >> 4;
    const unsigned bytes = __builtin_amdgcn_readfirstlane(blocks * 16u * (unsigned)stride * 2u);
    BDesc d;
    d.rows = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u16*>(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
    const bool has = tag != nullptr;
    const uint64_t tu = has ? reinterpret_cast<uint64_t>(tag + base_row) : bu;
    const uint32_t tlo = __builtin_amdgcn_readfirstlane((uint32_t)tu), thi = __builtin_amdgcn_readfirstlane((uint32_t)(tu >> 32));
    const unsigned tb = __builtin_amdgcn_readfirstlane(has ? (unsigned)(rows_here * 4) : 0u);
    d.tags = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<int32_t*>(((uint64_t)thi << 32) | tlo), 0, (int)tb, 0x00020000);
    return d;
}

template <int CHB>
struct BTile {
    bf16x8 a[4][CHB];
    int tag;  // tag of row lane (0..63) of the tile
};

template <int CHB, int NT, bool EXT, bool IVF>
__global__ __launch_bounds__(kBThreads, 2) void scan_bf16_topk_kernel(ScanBf16Args p) {
    constexpr int NQ = NT * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][kBWaves][NQ][kBPitch]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    // number of work items: tiles of the slab (flat) or entries of the probe plan (IVF: written by plan_probe_kernel earlier
    // on this stream)
    const int n_tiles = IVF ? __builtin_amdgcn_readfirstlane(*p.n_work) : (p.n_rows + kBTileRows - 1) / kBTileRows;
    const int G = gridDim.x;

    bf16x8 qf[NT][CHB];
    {
        const u16* qb = p.q_bf16 + (int64_t)m * p.row_stride + wid * 32 * CHB + 8 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < CHB; ++j)
                qf[nt][j] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)nt * 16 * p.row_stride + 32 * j);
    }
    const int voff_lane = wid * CHB * 1024 + lane * 16;
    const int blk_step = 16 * (int)p.row_stride * 2;

    TopList L[NT];
    float tau[NT];
    int qfilt[NT];
    // EXT: masked filters ((tag & mask) == filter) and the continuation bound of a k > 32 pass (only rows strictly
    // after (after_s, after_i) in (score desc, id asc) rank), as in the fp32 kernel
    int qmask[NT];
    float after_s[NT];
    int64_t after_i[NT];
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) {
        L[pq].s = -INFINITY;
        L[pq].i = 0x7fffffff;
        tau[pq] = -INFINITY;
        const int q = pq * 16 + (lane >> 5) * 8 + wid;
        const bool live = q < p.nq;
        qfilt[pq] = (p.q_filter != nullptr && live) ? p.q_filter[q] : -1;
        qmask[pq] = (EXT && p.q_filter_mask != nullptr && live) ? p.q_filter_mask[q] : -1;
        after_s[pq] = (EXT && p.q_after_score != nullptr && live) ? p.q_after_score[q] : INFINITY;
        after_i[pq] = (EXT && p.q_after_id != nullptr && live) ? p.q_after_id[q] : (int64_t)-1;
    }

    // The sample floor (flat scans; see scan_i8.hip): the k-th largest of the sample launch's per-workgroup best scores of a
    // query is reached by k rows of the slab; rows below it skip the sorted insertion (ties kept).
    float floor_q[NT];
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) floor_q[pq] = -INFINITY;
    if (!IVF && !EXT && p.sample_best != nullptr) {
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
        constexpr int kFloorBits = 20;
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

    auto issue = [&](BTile<CHB>& r, const BDesc& d) {
        r.tag = (int)__builtin_amdgcn_raw_buffer_load_b32(d.tags, lane * 4, 0, 0);
#pragma unroll
        for (int j = 0; j < CHB; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                r.a[b][j] = __builtin_bit_cast(
                    bf16x8, __builtin_amdgcn_raw_buffer_load_b128(d.rows, voff_lane + b * blk_step + j * 1024, 0, 2));
    };
    auto mul_refill = [&](BTile<CHB>& r, f32x4 (&acc)[4][NT], const BDesc& next) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[b][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        r.tag = (int)__builtin_amdgcn_raw_buffer_load_b32(next.tags, lane * 4, 0, 0);
#pragma unroll
        for (int j = 0; j < CHB; ++j) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bf16x8 a = r.a[b][j];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[b][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[nt][j], acc[b][nt], 0, 0, 0);
                r.a[b][j] = __builtin_bit_cast(
                    bf16x8, __builtin_amdgcn_raw_buffer_load_b128(next.rows, voff_lane + b * blk_step + j * 1024, 0, 2));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto finish = [&](const f32x4 (&acc)[4][NT], const BWork& w, int tag, int buf) {
        float* P = lds + buf * (kBWaves * NQ * kBPitch);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *reinterpret_cast<f32x4*>(P + (wid * NQ + nt * 16 + m) * kBPitch + b * 16 + 4 * g) = acc[b][nt];
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = (lane & 31) + 32 * half;
            const int row = w.tile * kBTileRows + r;
            const int rtag = __shfl(tag, r, 64);
            const bool row_ok = (r < w.rows) && (rtag != -1);
#pragma unroll
            for (int pq = 0; pq < NT; ++pq) {
                const int q = pq * 16 + (lane >> 5) * 8 + wid;
                const float* src = P + q * kBPitch + r;
                float s = src[0];
#pragma unroll
                for (int w = 1; w < kBWaves; ++w) s += src[w * NQ * kBPitch];
                bool ok;
                if (EXT) {
                    ok = row_ok && (qfilt[pq] < 0 || qfilt[pq] == (rtag & qmask[pq]));
                    ok = ok && (s < after_s[pq] || (s == after_s[pq] && (p.id_base + (int64_t)row) > after_i[pq]));
                } else {
                    ok = row_ok && (qfilt[pq] < 0 || qfilt[pq] == rtag) && (s >= floor_q[pq]);
                }
                if (IVF) ok = ok && ((w.mask >> q) & 1u) != 0;   // only the queries that probe this tile's list
                if (EXT && !IVF) insert_candidates(L[pq], tau[pq], ok ? s : -INFINITY, row, p.k);   // (the network's temporaries spill here)
                else insert_candidates_auto(L[pq], tau[pq], ok ? s : -INFINITY, row, p.k);
            }
        }
    };

    BTile<CHB> R0, R1;
    int t = blockIdx.x;
    BWork w0 = get_bwork<IVF>(p, t, n_tiles), w1 = get_bwork<IVF>(p, t + G, n_tiles);
    issue(R0, make_bdesc(p.corpus, p.row_stride, p.row_tag, w0));
    issue(R1, make_bdesc(p.corpus, p.row_stride, p.row_tag, w1));
    __builtin_amdgcn_sched_barrier(0);
    for (; t < n_tiles; t += 2 * G) {
        f32x4 acc[4][NT];
        int tag = R0.tag;
        const BWork w2 = get_bwork<IVF>(p, t + 2 * G, n_tiles);
        mul_refill(R0, acc, make_bdesc(p.corpus, p.row_stride, p.row_tag, w2));
        finish(acc, w0, tag, 0);
        tag = R1.tag;
        const BWork w3 = get_bwork<IVF>(p, t + 3 * G, n_tiles);
        mul_refill(R1, acc, make_bdesc(p.corpus, p.row_stride, p.row_tag, w3));
        finish(acc, w1, tag, 1);
        w0 = w2;
        w1 = w3;
    }
    const int lpos = lane & 31;
#pragma unroll
    for (int pq = 0; pq < NT; ++pq) {
        const int q = pq * 16 + (lane >> 5) * 8 + wid;
        if (q < p.nq && lpos < p.k) {
            const int64_t o = ((int64_t)blockIdx.x * p.nq + q) * p.k + lpos;
            const bool filled = L[pq].i != 0x7fffffff;
            p.part_scores[o] = filled ? L[pq].s : -INFINITY;
            // id_base = 0 for the candidate scan of the prefilter mode (the re-rank needs LOCAL rows)
            if (p.part_ids) p.part_ids[o] = filled ? (p.id_base + (int64_t)L[pq].i) : (int64_t)-1;   // nullptr: a sample launch
        }
    }
}

template <int CHB, int NT, bool EXT, bool IVF = false>
static hipError_t launch_bvariant(const ScanBf16Args& a, int grid, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)2 * kBWaves * NT * 16 * kBPitch * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bf16_topk_kernel<CHB, NT, EXT, IVF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((scan_bf16_topk_kernel<CHB, NT, EXT, IVF>), dim3(grid), dim3(kBThreads), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_scan_bf16_topk(const ScanBf16Args& a, int grid, hipStream_t stream) {
    if (a.row_stride % 256 != 0) return hipErrorInvalidValue;  // 8 waves x 32-column chunks
    const int chb = (int)(a.row_stride / 256);
    const bool two = a.nq > 16;
    const bool ext = a.q_filter_mask || a.q_after_score || a.q_after_id;
    if (ext) {
        if ((a.q_after_score == nullptr) != (a.q_after_id == nullptr)) return hipErrorInvalidValue;
        if (a.q_filter_mask != nullptr && a.q_filter == nullptr) return hipErrorInvalidValue;
    }
    const bool ivf = a.work_tile != nullptr;
    if (a.sample_best && (a.sample_groups < 1 || a.sample_groups > kMaxSampleGroups || ivf || ext)) return hipErrorInvalidValue;
    // IVF probe: plain or masked filters; no continuation bound (the ids it compares would be slab positions)
    if (ivf && (a.q_after_score || a.q_after_id || !a.work_rows || !a.work_mask || !a.n_work)) return hipErrorInvalidValue;
#define RASS_BF16_CASE(C)                                                                                     \
    case C:                                                                                                   \
        if (ivf && ext) return two ? launch_bvariant<C, 2, true, true>(a, grid, stream) : launch_bvariant<C, 1, true, true>(a, grid, stream); \
        if (ivf) return two ? launch_bvariant<C, 2, false, true>(a, grid, stream) : launch_bvariant<C, 1, false, true>(a, grid, stream); \
        if (ext) return two ? launch_bvariant<C, 2, true>(a, grid, stream) : launch_bvariant<C, 1, true>(a, grid, stream); \
        return two ? launch_bvariant<C, 2, false>(a, grid, stream) : launch_bvariant<C, 1, false>(a, grid, stream);
    switch (chb) {
        RASS_BF16_CASE(1)
        RASS_BF16_CASE(2)
        RASS_BF16_CASE(3)
        RASS_BF16_CASE(4)
        default: return hipErrorInvalidValue;
    }
#undef RASS_BF16_CASE
}

// fp32 tile16 slab -> bf16 tile16b slab, blocks [b0, b1).  bf16 lane (m, g) of chunk jb holds
// columns 32jb + 8g .. +7 = fp32 chunk 2jb + (g>>1), lane groups 2(g&1) and 2(g&1)+1.
// `src_b0`: the source block that lands in destination block b0 (a staging slab starts at block 0); only rows in
// [row_lo, row_hi) of the DESTINATION are written, so a partially filled block keeps its earlier rows.
__global__ __launch_bounds__(256) void convert_tile16_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst,
                                                                  int64_t stride, int64_t b0, int64_t b1, int64_t src_b0,
                                                                  int64_t row_lo, int64_t row_hi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nchb = (int)(stride >> 5);
    for (int64_t b = b0 + (int64_t)blockIdx.x * 4 + wave; b < b1; b += (int64_t)gridDim.x * 4) {
        const float* sb = src + (b - b0 + src_b0) * 16 * stride;
        u16* db = dst + b * 16 * stride;
        const int64_t r = b * 16 + m;
        if (r < row_lo || r >= row_hi) continue;
        for (int jb = 0; jb < nchb; ++jb) {
            const float* sc = sb + (int64_t)(2 * jb + (g >> 1)) * 256;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sc + ((2 * (g & 1)) * 16 + m) * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sc + ((2 * (g & 1) + 1) * 16 + m) * 4);
            uint4 o;
            o.x = (unsigned)f2bf_s(v0.x) | ((unsigned)f2bf_s(v0.y) << 16);
            o.y = (unsigned)f2bf_s(v0.z) | ((unsigned)f2bf_s(v0.w) << 16);
            o.z = (unsigned)f2bf_s(v1.x) | ((unsigned)f2bf_s(v1.y) << 16);
            o.w = (unsigned)f2bf_s(v1.z) | ((unsigned)f2bf_s(v1.w) << 16);
            *reinterpret_cast<uint4*>(db + (int64_t)jb * 512 + lane * 8) = o;
        }
    }
}

hipError_t launch_convert_tile16_bf16(const float* src, void* dst, int64_t stride, int64_t block0, int64_t block1,
                                      hipStream_t stream, int64_t src_block0, int64_t row_lo, int64_t row_hi) {
    if (src_block0 < 0) src_block0 = block0;
    if (row_hi < 0) row_hi = block1 * 16;
    if (block1 <= block0) return hipSuccess;
    if (stride % 32 != 0) return hipErrorInvalidValue;
    int64_t blocks = (block1 - block0 + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(convert_tile16_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src,
                       static_cast<u16*>(dst), stride, block0, block1, src_block0, row_lo, row_hi);
    return hipGetLastError();
}

// bf16 tile16b slab rows [first_row, first_row + n) -> row-major fp32 (exact upcast), dim columns at out_stride
__global__ void unpack_rows_tile16b_kernel(const u16* __restrict__ slab, int64_t stride, int64_t first_row, int64_t n,
                                           int dim, float* __restrict__ out, int64_t out_stride) {
    const int64_t total = n * dim;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / dim;
        const int c = (int)(e - i * dim);
        const int64_t r = first_row + i;
        const int64_t off = (r >> 4) * 16 * stride + (int64_t)(c >> 5) * 512 + ((((c >> 3) & 3) * 16 + (r & 15)) * 8) + (c & 7);
        out[i * out_stride + c] = __uint_as_float((unsigned)slab[off] << 16);
    }
}

hipError_t launch_unpack_rows_tile16b(const void* slab, int64_t stride, int64_t first_row, int64_t n, int dim,
                                      float* out, int64_t out_stride, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n * dim + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unpack_rows_tile16b_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       static_cast<const u16*>(slab), stride, first_row, n, dim, out, out_stride);
    return hipGetLastError();
}

// queries: normalised fp32 [nq_pad][stride] (q_padded of the fp32 path) -> bf16 [nq_pad][stride]
__global__ void queries_to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = f2bf_s(src[i]);
}

hipError_t launch_queries_to_bf16(const float* src, void* dst, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(queries_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, static_cast<u16*>(dst), n);
    return hipGetLastError();
}

// Exact re-rank.  One workgroup per query: thread (c = tid>>3, w = tid&7) recomputes K-slice w of
// candidate c's score in the flat kernel's order (chunk j, component i, lane group g: the fmaf
// chain of v_mfma_f32_16x16x4_f32), the 8 slice partials are added in slice order, then the 32
// exact (score, id) pairs are ranked by counting and the best k are written.  CH = chunks per slice (stride / 128), a
// template parameter so that the slice's 4 CH row loads are all in flight before the first fmaf (with a runtime bound the
// loop paid one gather round trip per chunk: 26 us per launch at 1024 columns, r4).
template <int CH>
__global__ __launch_bounds__(256) void rerank_f32_kernel(const float* __restrict__ slab, int64_t stride,
                                                         const float* __restrict__ q_padded,
                                                         const int64_t* __restrict__ cand_rows, int n_cand, int k,
                                                         int64_t id_base, float* __restrict__ out_scores,
                                                         int64_t* __restrict__ out_ids, int64_t gs, int64_t gi,
                                                         const int64_t* __restrict__ id_map) {
    __shared__ float part[32][9];
    __shared__ float sc[32];
    __shared__ int64_t rw[32];
    const int q = blockIdx.x;
    float* const os = out_scores + (gs > 0 ? (int64_t)(q >> 5) * gs + (int64_t)(q & 31) * k : (int64_t)q * k);
    int64_t* const oi = out_ids + (gi > 0 ? (int64_t)(q >> 5) * gi + (int64_t)(q & 31) * k : (int64_t)q * k);
    const int c = threadIdx.x >> 3, w = threadIdx.x & 7;
    const int64_t row = c < n_cand ? cand_rows[(int64_t)q * n_cand + c] : -1;
    float acc = 0.f;
    if (row >= 0) {
        const float* xb = slab + (row >> 4) * 16 * stride + (int64_t)(w * CH) * 256 + (int)(row & 15) * 4;
        const float* qv = q_padded + (int64_t)q * stride + w * CH * 16;
        f32x4 xv[CH][4];
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) xv[j][g] = *reinterpret_cast<const f32x4*>(xb + j * 256 + g * 64);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            f32x4 qq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) qq[g] = *reinterpret_cast<const f32x4*>(qv + j * 16 + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc = fmaf(xv[j][g][i], qq[g][i], acc);
        }
    }
    part[c][w] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int cc = threadIdx.x;
        float s = part[cc][0];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) s += part[cc][ww];
        const int64_t r = cc < n_cand ? cand_rows[(int64_t)q * n_cand + cc] : -1;
        sc[cc] = r >= 0 ? s : -INFINITY;
        // the id that is reported and that breaks score ties: the slab row itself, or (an IVF's permuted slab) the source row
        rw[cc] = (r >= 0 && id_map != nullptr) ? id_map[r] : r;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int cc = threadIdx.x;
        const float s = sc[cc];
        const int64_t r = rw[cc];
        int rank = 0;
        for (int o = 0; o < 32; ++o) {
            const float so = sc[o];
            const int64_t ro = rw[o];
            const bool o_valid = ro >= 0, me_valid = r >= 0;
            const bool better = o_valid && (!me_valid || so > s || (so == s && ro < r));
            rank += (o != cc && better) ? 1 : 0;
        }
        if (r < 0) rank = 32 + cc;  // invalid entries never land in [0, k)
        if (rank < k) {
            os[rank] = s;
            oi[rank] = id_base + r;
        }
    }
    // slots beyond the number of valid candidates
    if (threadIdx.x < 32) {
        int valid = 0;
        for (int o = 0; o < 32; ++o) valid += rw[o] >= 0 ? 1 : 0;
        const int e = threadIdx.x;
        if (e >= valid && e < k) {
            os[e] = -INFINITY;
            oi[e] = -1;
        }
    }
}

// Wide rows (stride 1 280 .. 2 048: 10 .. 16 chunks per K slice): the same arithmetic with the slice walked in two halves of
// CH chunks each (the loads of a half in flight together), the fmaf chain running on across them in chunk order.
template <int CH>
__global__ __launch_bounds__(256) void rerank_f32_wide_kernel(const float* __restrict__ slab, int64_t stride,
                                                              const float* __restrict__ q_padded,
                                                              const int64_t* __restrict__ cand_rows, int n_cand, int k,
                                                              int64_t id_base, float* __restrict__ out_scores,
                                                              int64_t* __restrict__ out_ids, int64_t gs, int64_t gi,
                                                              const int64_t* __restrict__ id_map) {
    __shared__ float part[32][9];
    __shared__ float sc[32];
    __shared__ int64_t rw[32];
    const int q = blockIdx.x;
    float* const os = out_scores + (gs > 0 ? (int64_t)(q >> 5) * gs + (int64_t)(q & 31) * k : (int64_t)q * k);
    int64_t* const oi = out_ids + (gi > 0 ? (int64_t)(q >> 5) * gi + (int64_t)(q & 31) * k : (int64_t)q * k);
    const int c = threadIdx.x >> 3, w = threadIdx.x & 7;
    const int64_t row = c < n_cand ? cand_rows[(int64_t)q * n_cand + c] : -1;
    float acc = 0.f;
    if (row >= 0) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float* xb = slab + (row >> 4) * 16 * stride + (int64_t)(w * 2 * CH + half * CH) * 256 + (int)(row & 15) * 4;
            const float* qv = q_padded + (int64_t)q * stride + (w * 2 * CH + half * CH) * 16;
            f32x4 xv[CH][4];
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) xv[j][g] = *reinterpret_cast<const f32x4*>(xb + j * 256 + g * 64);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                f32x4 qq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) qq[g] = *reinterpret_cast<const f32x4*>(qv + j * 16 + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc = fmaf(xv[j][g][i], qq[g][i], acc);
            }
        }
    }
    part[c][w] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int cc = threadIdx.x;
        float s = part[cc][0];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) s += part[cc][ww];
        const int64_t r = cc < n_cand ? cand_rows[(int64_t)q * n_cand + cc] : -1;
        sc[cc] = r >= 0 ? s : -INFINITY;
        rw[cc] = (r >= 0 && id_map != nullptr) ? id_map[r] : r;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int cc = threadIdx.x;
        const float s = sc[cc];
        const int64_t r = rw[cc];
        int rank = 0;
        for (int o = 0; o < 32; ++o) {
            const float so = sc[o];
            const int64_t ro = rw[o];
            const bool o_valid = ro >= 0, me_valid = r >= 0;
            const bool better = o_valid && (!me_valid || so > s || (so == s && ro < r));
            rank += (o != cc && better) ? 1 : 0;
        }
        if (r < 0) rank = 32 + cc;
        if (rank < k) {
            os[rank] = s;
            oi[rank] = id_base + r;
        }
        int valid = 0;
        for (int o = 0; o < 32; ++o) valid += rw[o] >= 0 ? 1 : 0;
        if (cc >= valid && cc < k) {
            os[cc] = -INFINITY;
            oi[cc] = -1;
        }
    }
}

hipError_t launch_rerank_f32(const float* slab, int64_t stride, const float* q_padded, const int64_t* cand_rows, int nq,
                             int n_cand, int k, int64_t id_base, float* out_scores, int64_t* out_ids,
                             hipStream_t stream, int64_t out_scores_group_stride, int64_t out_ids_group_stride,
                             const int64_t* id_map) {
    if (nq < 1 || n_cand < 1 || n_cand > 32 || k < 1 || k > n_cand || stride % 128 != 0 || stride > 2048) return hipErrorInvalidValue;
    if (stride > 1024) {   // wide rows: strides of 256 * {5..8} = two halves of 5..8 chunks per K slice
        if (stride % 256 != 0) return hipErrorInvalidValue;
#define RASS_RERANK_WIDE(C)                                                                                                        \
    case C:                                                                                                                        \
        hipLaunchKernelGGL(rerank_f32_wide_kernel<C>, dim3(nq), dim3(256), 0, stream, slab, stride, q_padded, cand_rows, n_cand, k, \
                           id_base, out_scores, out_ids, out_scores_group_stride, out_ids_group_stride, id_map);                   \
        break;
        switch ((int)(stride >> 8)) {
            RASS_RERANK_WIDE(5) RASS_RERANK_WIDE(6) RASS_RERANK_WIDE(7) RASS_RERANK_WIDE(8)
            default: return hipErrorInvalidValue;
        }
#undef RASS_RERANK_WIDE
        return hipGetLastError();
    }
#define RASS_RERANK_CASE(C)                                                                                                   \
    case C:                                                                                                                   \
        hipLaunchKernelGGL(rerank_f32_kernel<C>, dim3(nq), dim3(256), 0, stream, slab, stride, q_padded, cand_rows, n_cand, k, \
                           id_base, out_scores, out_ids, out_scores_group_stride, out_ids_group_stride, id_map);              \
        break;
    switch ((int)(stride >> 7)) {
        RASS_RERANK_CASE(1) RASS_RERANK_CASE(2) RASS_RERANK_CASE(3) RASS_RERANK_CASE(4)
        RASS_RERANK_CASE(5) RASS_RERANK_CASE(6) RASS_RERANK_CASE(7) RASS_RERANK_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef RASS_RERANK_CASE
    return hipGetLastError();
}

}  // namespace rass
