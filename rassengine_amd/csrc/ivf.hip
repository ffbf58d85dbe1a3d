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
                                                                  int64_t* __restrict__ scanned_rows) {
    extern __shared__ uint32_t sh[];   // [nlist] query masks, then [kPlanThreads] scan scratch
    uint32_t* mask = sh;
    uint32_t* part = sh + nlist;
    const int tid = threadIdx.x;
    for (int l = tid; l < nlist; l += kPlanThreads) mask[l] = 0u;
    __syncthreads();
    for (int e = tid; e < nq * nprobe; e += kPlanThreads) {
        const int64_t l = probe_ids[e];
        if (l >= 0 && l < nlist) atomicOr(&mask[(int)l], 1u << (e / nprobe));
    }
    __syncthreads();
    // each thread owns a contiguous chunk of lists; exclusive scan of the chunks' tile counts
    const int per = (nlist + kPlanThreads - 1) / kPlanThreads;
    const int l0 = tid * per, l1 = min(nlist, l0 + per);
    uint32_t cnt = 0;
    uint32_t rows = 0;
    for (int l = l0; l < l1; ++l)
        if (mask[l]) {
            cnt += (uint32_t)((list_len[l] + 31) / 32);
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
    uint32_t o = part[tid] - cnt;
    for (int l = l0; l < l1; ++l) {
        const uint32_t mk = mask[l];
        if (!mk) continue;
        const int len = list_len[l];
        const int nt = (len + 31) / 32;
        for (int t = 0; t < nt; ++t) {
            work_tile[o] = list_tile0[l] + t;
            work_rows[o] = min(32, len - 32 * t);
            work_mask[o] = mk;
            ++o;
        }
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
                             int32_t* n_work, int64_t* scanned_rows, hipStream_t stream) {
    if (nq < 1 || nq > 32 || nprobe < 1 || nlist < 1 || nlist > 32768) return hipErrorInvalidValue;
    const size_t lds = ((size_t)nlist + kPlanThreads) * sizeof(uint32_t);
    static size_t attr = 0;
    if (lds > attr && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_probe_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    hipLaunchKernelGGL(plan_probe_kernel, dim3(1), dim3(kPlanThreads), lds, stream, probe_ids, nq, nprobe, nlist,
                       list_tile0, list_len, work_tile, work_rows, work_mask, n_work, scanned_rows);
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

}  // namespace rass
