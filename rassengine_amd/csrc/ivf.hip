// ivf.hip — K9 of SURVEY §8a: the device side of the IVF (inverted-file) cosine index that
// stands in for HNSW's sub-linear behaviour at 100 M rows (reference: OpenSearch k-NN HNSW,
// app/main.py:563-572).
//
// Layout: the corpus slab is the same tile16 fp32 layout the flat scan streams, permuted so
// that the rows of inverted list l are contiguous and start on a 32-row tile boundary
// (list_tile0[l], list_len[l]); slab_ids[] maps a slab row back to the caller's row id.
// Probe = (i) coarse: the fused flat scan over the centroid slab, top-nprobe; (ii) plan: turn
// the batch's probed lists into ONE work list of slab tiles, each with the bitmask of queries
// that probe its list; (iii) the SAME fused scan kernel iterating that work list (IVF mode of
// scan_topk.hip).  A tile is fetched once per batch however many queries probe it.

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "kernels.h"

namespace rass {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPlanThreads = 1024;

// One workgroup.  probe_ids: [nq][nprobe] list ids from the coarse scan (-1 = none).
__global__ __launch_bounds__(kPlanThreads) void plan_probe_kernel(const int64_t* __restrict__ probe_ids, int nq,
                                                                  int nprobe, int nlist,
                                                                  const int32_t* __restrict__ list_tile0,
                                                                  const int32_t* __restrict__ list_len,
                                                                  int32_t* __restrict__ work_tile,
                                                                  int32_t* __restrict__ work_rows,
                                                                  uint32_t* __restrict__ work_mask,
                                                                  int32_t* __restrict__ n_work,
                                                                  int64_t* __restrict__ scanned_rows,
                                                                  const uint32_t* __restrict__ preset_mask,
                                                                  int tile_rows) {
    extern __shared__ uint32_t sh[];   // [nlist] query masks, then [kPlanThreads] scan scratch
    uint32_t* mask = sh;
    uint32_t* part = sh + nlist;
    const int tid = threadIdx.x;
    for (int l = tid; l < nlist; l += kPlanThreads) mask[l] = preset_mask ? preset_mask[l] : 0u;
    __syncthreads();
    if (!preset_mask) {
        for (int e = tid; e < nq * nprobe; e += kPlanThreads) {
            const int64_t l = probe_ids[e];
            if (l >= 0 && l < nlist) atomicOr(&mask[(int)l], 1u << (e / nprobe));
        }
    }
    __syncthreads();
    // each thread owns a contiguous chunk of lists; exclusive scan of the chunks' tile counts
    const int per = (nlist + kPlanThreads - 1) / kPlanThreads;
    const int l0 = tid * per, l1 = min(nlist, l0 + per);
    uint32_t cnt = 0;
    uint32_t rows = 0;
    for (int l = l0; l < l1; ++l)
        if (mask[l]) {
            cnt += (uint32_t)((list_len[l] + tile_rows - 1) / tile_rows);
            rows += (uint32_t)list_len[l];
        }
    part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < kPlanThreads; off <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    // Write-out, flat over the OUTPUT positions (round 3): thread tid fills positions tid, tid + 1024, ... — the owner chunk of a
    // position by binary search over the chunks' inclusive prefix, then a walk over that chunk's <= `per` lists.  (Each thread
    // writing its own lists' tiles one after the other made one thread write a whole probed list — ~95 tiles at 3 000 rows —
    // while 1 000 others idled: 17 / 44 / 81 us per plan at nprobe 1 / 4 / 128 in the kernel trace.)
    const uint32_t total = part[kPlanThreads - 1];
    for (uint32_t i = tid; i < total; i += kPlanThreads) {
        int lo = 0, hi = kPlanThreads - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (part[mid] > i) hi = mid; else lo = mid + 1;
        }
        uint32_t base = lo ? part[lo - 1] : 0u;
        int l = lo * per;
        const int lend = min(nlist, l + per);
        uint32_t mk = 0;
        int len = 0;
        for (; l < lend; ++l) {
            mk = mask[l];
            if (!mk) continue;
            len = list_len[l];
            const uint32_t nt = (uint32_t)((len + tile_rows - 1) / tile_rows);
            if (i - base < nt) break;
            base += nt;
        }
        const int t = (int)(i - base);
        work_tile[i] = list_tile0[l] + t;
        work_rows[i] = min(tile_rows, len - tile_rows * t);
        work_mask[i] = mk;
    }
    if (tid == kPlanThreads - 1) *n_work = (int32_t)part[tid];
    // rows the fine scan will touch (recall / bytes bookkeeping)
    __syncthreads();
    part[tid] = rows;
    __syncthreads();
    for (int off = kPlanThreads / 2; off > 0; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0 && scanned_rows) *scanned_rows = (int64_t)part[0];
}

hipError_t launch_plan_probe(const int64_t* probe_ids, int nq, int nprobe, int nlist, const int32_t* list_tile0,
                             const int32_t* list_len, int32_t* work_tile, int32_t* work_rows, uint32_t* work_mask,
                             int32_t* n_work, int64_t* scanned_rows, hipStream_t stream, const uint32_t* preset_mask,
                             int tile_rows) {
    if (nq < 1 || nq > 32 || nprobe < 1 || nlist < 1 || nlist > 32768) return hipErrorInvalidValue;
    if (tile_rows != 32 && tile_rows != 64) return hipErrorInvalidValue;   // the fp32 scan's tile / the bf16 scan's
    const size_t lds = ((size_t)nlist + kPlanThreads) * sizeof(uint32_t);
    static size_t attr = 0;
    if (lds > attr && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_probe_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    hipLaunchKernelGGL(plan_probe_kernel, dim3(1), dim3(kPlanThreads), lds, stream, probe_ids, nq, nprobe, nlist,
                       list_tile0, list_len, work_tile, work_rows, work_mask, n_work, scanned_rows, preset_mask, tile_rows);
    return hipGetLastError();
}

// ---- the plan of a whole BATCH of launch groups, straight from the coarse scan's per-workgroup lists ----------------
// One workgroup per launch group g (gridDim.x groups).  The coarse stage (kFlatGroups) left, per group,
// [n_clists][32][nprobe] (score desc, id asc) candidate lists; a query's probed lists are the nprobe best of its
// n_clists * nprobe candidates under that order — found here by counting, per candidate, the candidates that beat it
// (<= 256 of them, in LDS), so no merge launch and no sorted [nq][nprobe] list in HBM in between.  The rest is
// plan_probe_kernel's: list masks -> tile counts -> prefix scan -> the group's work list (its own slice of the work
// arrays, work_cap entries) + its item count and scanned rows.
__global__ __launch_bounds__(kPlanThreads) void plan_probe_groups_kernel(const float* __restrict__ cpart_scores,
                                                                         const int64_t* __restrict__ cpart_ids,
                                                                         int n_clists, int nprobe, int nq_total,
                                                                         int64_t cpart_group_stride, int nlist,
                                                                         const int32_t* __restrict__ list_tile0,
                                                                         const int32_t* __restrict__ list_len,
                                                                         int32_t* __restrict__ work_tile,
                                                                         int32_t* __restrict__ work_rows,
                                                                         uint32_t* __restrict__ work_mask,
                                                                         int64_t work_cap, int32_t* __restrict__ n_work,
                                                                         int64_t* __restrict__ scanned_rows, int tile_rows) {
    extern __shared__ uint32_t sh[];   // [nlist] masks | [kPlanThreads] scan scratch | [32 * C] scores | [32 * C] ids
    const int g = blockIdx.x;
    const int C = n_clists * nprobe;                       // candidates per query
    uint32_t* mask = sh;
    uint32_t* part = sh + nlist;
    float* cs = reinterpret_cast<float*>(part + kPlanThreads);
    int32_t* ci = reinterpret_cast<int32_t*>(cs + 32 * C);
    const int tid = threadIdx.x;
    const int nq = min(32, nq_total - 32 * g);             // the last group may be short (its padded queries are ignored)
    cpart_scores += (int64_t)g * cpart_group_stride;
    cpart_ids += (int64_t)g * cpart_group_stride;
    work_tile += (int64_t)g * work_cap;
    work_rows += (int64_t)g * work_cap;
    work_mask += (int64_t)g * work_cap;
    for (int l = tid; l < nlist; l += kPlanThreads) mask[l] = 0u;
    for (int e = tid; e < 32 * C; e += kPlanThreads) {     // candidate (q, c): list c / nprobe, position c % nprobe
        const int q = e / C, c = e - q * C;
        const int64_t o = ((int64_t)(c / nprobe) * 32 + q) * nprobe + (c % nprobe);
        cs[e] = cpart_scores[o];
        ci[e] = (int32_t)cpart_ids[o];
    }
    __syncthreads();
    for (int e = tid; e < nq * C; e += kPlanThreads) {
        const int q = e / C;
        const float s = cs[e];
        const int32_t id = ci[e];
        if (id < 0 || id >= nlist) continue;
        int beaten = 0;                                    // candidates of q that rank before this one
        const float* qs = cs + q * C;
        const int32_t* qi = ci + q * C;
        for (int j = 0; j < C; ++j) {
            const float sj = qs[j];
            const int32_t ij = qi[j];
            beaten += (ij >= 0) && (sj > s || (sj == s && ij < id));
        }
        if (beaten < nprobe) atomicOr(&mask[id], 1u << q);
    }
    __syncthreads();
    const int per = (nlist + kPlanThreads - 1) / kPlanThreads;
    const int l0 = tid * per, l1 = min(nlist, l0 + per);
    uint32_t cnt = 0;
    uint32_t rows = 0;
    for (int l = l0; l < l1; ++l)
        if (mask[l]) {
            cnt += (uint32_t)((list_len[l] + tile_rows - 1) / tile_rows);
            rows += (uint32_t)list_len[l];
        }
    part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < kPlanThreads; off <<= 1) {
        const uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const uint32_t total = part[kPlanThreads - 1];
    for (uint32_t i = tid; i < total; i += kPlanThreads) {
        int lo = 0, hi = kPlanThreads - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (part[mid] > i) hi = mid; else lo = mid + 1;
        }
        uint32_t base = lo ? part[lo - 1] : 0u;
        int l = lo * per;
        const int lend = min(nlist, l + per);
        uint32_t mk = 0;
        int len = 0;
        for (; l < lend; ++l) {
            mk = mask[l];
            if (!mk) continue;
            len = list_len[l];
            const uint32_t nt = (uint32_t)((len + tile_rows - 1) / tile_rows);
            if (i - base < nt) break;
            base += nt;
        }
        const int t = (int)(i - base);
        work_tile[i] = list_tile0[l] + t;
        work_rows[i] = min(tile_rows, len - tile_rows * t);
        work_mask[i] = mk;
    }
    if (tid == kPlanThreads - 1) n_work[g] = (int32_t)part[tid];
    __syncthreads();
    part[tid] = rows;
    __syncthreads();
    for (int off = kPlanThreads / 2; off > 0; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0 && scanned_rows) scanned_rows[g] = (int64_t)part[0];
}

hipError_t launch_plan_probe_groups(const float* cpart_scores, const int64_t* cpart_ids, int n_clists, int nprobe, int groups,
                                    int nq_total, int64_t cpart_group_stride, int nlist, const int32_t* list_tile0,
                                    const int32_t* list_len, int32_t* work_tile, int32_t* work_rows, uint32_t* work_mask,
                                    int64_t work_cap, int32_t* n_work, int64_t* scanned_rows, hipStream_t stream, int tile_rows) {
    if (groups < 1 || nprobe < 1 || nprobe > 32 || n_clists < 1 || n_clists * nprobe > 256 || nlist < 1 || nlist > 32768)
        return hipErrorInvalidValue;
    if (tile_rows != 32 && tile_rows != 64) return hipErrorInvalidValue;
    const size_t lds = ((size_t)nlist + kPlanThreads + 2 * 32 * (size_t)n_clists * nprobe) * sizeof(uint32_t);
    static size_t attr = 0;
    if (lds > attr && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_probe_groups_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    hipLaunchKernelGGL(plan_probe_groups_kernel, dim3(groups), dim3(kPlanThreads), lds, stream, cpart_scores, cpart_ids,
                       n_clists, nprobe, nq_total, cpart_group_stride, nlist, list_tile0, list_len, work_tile, work_rows,
                       work_mask, work_cap, n_work, scanned_rows, tile_rows);
    return hipGetLastError();
}

// ---- nprobe > 32: threshold select + plan from the full centroid score matrix ------------------
// The coarse scan is launched with one workgroup per 32-centroid tile and k = 32, so its
// per-workgroup lists [n_ctiles][nq][32] hold EVERY centroid's score.  tau[q] = the nprobe-th
// largest of them (radix select on order-preserving integer keys, 4 passes of 8 bits).
__device__ __forceinline__ uint32_t desc_key(float s, int64_t id) {
    if (id < 0 || !(s > -INFINITY)) return 0xffffffffu;          // never selected
    const uint32_t b = __float_as_uint(s);
    const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending in s
    return ~asc;                                                  // ascending key = descending score
}

__global__ __launch_bounds__(256) void ivf_threshold_kernel(const float* __restrict__ part_scores,
                                                            const int64_t* __restrict__ part_ids, int n_ctiles,
                                                            int nq, int nprobe, uint32_t* __restrict__ tau_key) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_prefix, sel_remaining;
    const int q = blockIdx.x;
    const int n = n_ctiles * 32;
    if (threadIdx.x == 0) {
        sel_prefix = 0;
        sel_remaining = (uint32_t)nprobe;
    }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t prefix = sel_prefix;
        const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
        for (int e = threadIdx.x; e < n; e += 256) {
            const int64_t o = ((int64_t)(e >> 5) * nq + q) * 32 + (e & 31);
            const uint32_t key = desc_key(part_scores[o], part_ids[o]);
            if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t rem = sel_remaining, b = 0;
            for (; b < 255; ++b) {
                if (hist[b] >= rem) break;
                rem -= hist[b];
            }
            sel_prefix = prefix | (b << shift);
            sel_remaining = rem;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) tau_key[q] = sel_prefix;  // keys <= tau are probed (ties may add a few lists)
}

hipError_t launch_ivf_threshold(const float* part_scores, const int64_t* part_ids, int n_ctiles, int nq, int nprobe,
                                uint32_t* tau_key, hipStream_t stream) {
    if (n_ctiles < 1 || nq < 1 || nprobe < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivf_threshold_kernel, dim3(nq), dim3(256), 0, stream, part_scores, part_ids, n_ctiles, nq, nprobe,
                       tau_key);
    return hipGetLastError();
}

// Probe masks from the score matrix + thresholds, written to a global mask array that
// plan_probe_kernel then consumes through its `preset_mask` input.
__global__ void ivf_mask_from_scores_kernel(const float* __restrict__ part_scores,
                                            const int64_t* __restrict__ part_ids, int n_ctiles, int nq, int nlist,
                                            const uint32_t* __restrict__ tau_key, uint32_t* __restrict__ mask) {
    const int64_t total = (int64_t)n_ctiles * nq * 32;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)((o >> 5) % nq);
        const int64_t l = part_ids[o];
        if (l >= 0 && l < nlist && desc_key(part_scores[o], l) <= tau_key[q]) atomicOr(&mask[l], 1u << q);
    }
}

hipError_t launch_ivf_mask_from_scores(const float* part_scores, const int64_t* part_ids, int n_ctiles, int nq,
                                       int nlist, const uint32_t* tau_key, uint32_t* mask, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)nlist * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const int64_t total = (int64_t)n_ctiles * nq * 32;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ivf_mask_from_scores_kernel, dim3(blocks), dim3(256), 0, stream, part_scores, part_ids, n_ctiles,
                       nq, nlist, tau_key, mask);
    return hipGetLastError();
}

// dst slab row d <- src slab row src_of[d] (tile16 both sides); src_of[d] < 0 -> zero row.
// One wave per 16-row destination block: coalesced 1 KiB writes, 16-B gathers on the source.
__global__ __launch_bounds__(256) void permute_rows_tile16_kernel(const float* __restrict__ src,
                                                                  float* __restrict__ dst, int64_t stride,
                                                                  const int64_t* __restrict__ src_of,
                                                                  int64_t dst_rows) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nchunks = (int)(stride >> 4);
    const int64_t nblk = dst_rows >> 4;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < nblk; b += (int64_t)gridDim.x * 4) {
        const int64_t s = src_of[b * 16 + m];
        const float* sp = s >= 0 ? src + (s >> 4) * 16 * stride + ((int64_t)g * 16 + (s & 15)) * 4 : nullptr;
        float* dp = dst + b * 16 * stride + lane * 4;
        for (int j = 0; j < nchunks; ++j) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (sp) v = *reinterpret_cast<const f32x4*>(sp + (int64_t)j * 256);
            *reinterpret_cast<f32x4*>(dp + (int64_t)j * 256) = v;
        }
    }
}

hipError_t launch_permute_rows_tile16(const float* src, float* dst, int64_t stride, const int64_t* src_of,
                                      int64_t dst_rows, hipStream_t stream) {
    if (dst_rows <= 0) return hipSuccess;
    if (dst_rows % 16 != 0) return hipErrorInvalidValue;
    int64_t blocks = (dst_rows / 16 + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(permute_rows_tile16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, stride,
                       src_of, dst_rows);
    return hipGetLastError();
}

// The same permutation into a bf16 tile16b slab (scan_bf16.hip's layout: lane (m, g) of 32-column chunk jb holds
// X[16b + m][32jb + 8g .. +7]), rounding to bf16 on the way (round to nearest even, as convert_tile16_bf16_kernel): the
// IVF over a bf16 slab is built from the fp32 index without an fp32 list-ordered copy in between.
__global__ __launch_bounds__(256) void permute_rows_tile16_bf16_kernel(const float* __restrict__ src,
                                                                       unsigned short* __restrict__ dst, int64_t stride,
                                                                       const int64_t* __restrict__ src_of,
                                                                       int64_t dst_rows) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nchb = (int)(stride >> 5);
    const int64_t nblk = dst_rows >> 4;
    auto bf = [](float f) -> unsigned {   // the conversion convert_tile16_bf16_kernel uses: the two slabs hold the same bits
        __hip_bfloat16 h = __float2bfloat16(f);
        return (unsigned)*reinterpret_cast<unsigned short*>(&h);
    };
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < nblk; b += (int64_t)gridDim.x * 4) {
        const int64_t s = src_of[b * 16 + m];
        const float* sp = s >= 0 ? src + (s >> 4) * 16 * stride + (int64_t)(g >> 1) * 256 + ((2 * (g & 1)) * 16 + (s & 15)) * 4 : nullptr;
        unsigned short* dp = dst + b * 16 * stride + lane * 8;
        for (int jb = 0; jb < nchb; ++jb) {
            uint4 o = uint4{0u, 0u, 0u, 0u};
            if (sp) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp + (int64_t)jb * 512);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + (int64_t)jb * 512 + 64);
                o.x = bf(v0.x) | (bf(v0.y) << 16);
                o.y = bf(v0.z) | (bf(v0.w) << 16);
                o.z = bf(v1.x) | (bf(v1.y) << 16);
                o.w = bf(v1.z) | (bf(v1.w) << 16);
            }
            *reinterpret_cast<uint4*>(dp + (int64_t)jb * 512) = o;
        }
    }
}

hipError_t launch_permute_rows_tile16_bf16(const float* src, void* dst, int64_t stride, const int64_t* src_of,
                                           int64_t dst_rows, hipStream_t stream) {
    if (dst_rows <= 0) return hipSuccess;
    if (dst_rows % 16 != 0 || stride % 32 != 0) return hipErrorInvalidValue;
    int64_t blocks = (dst_rows / 16 + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(permute_rows_tile16_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src,
                       static_cast<unsigned short*>(dst), stride, src_of, dst_rows);
    return hipGetLastError();
}

}  // namespace rass
