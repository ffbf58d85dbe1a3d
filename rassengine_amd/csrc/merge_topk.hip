// merge_topk.hip — K2 (second stage) and K3's post-all-gather merge of SURVEY §8a.
//
// Input: n_lists candidate lists per query, [n_lists][nq][k] (score f32, id i64; id < 0 =
// empty slot).  Output: [nq][k], best first under (score desc, id asc) — the same total
// order the scan kernel keeps, so merging per-workgroup lists, or per-GPU lists after the
// RCCL all-gather, gives exactly the single-scan result.  This is the analogue of
// OpenSearch's shard -> coordinator top-k merge (reference SHARD_COUNT, app/main.py:89).
//
// One 1024-thread workgroup per query; the candidates are bitonic-sorted in LDS
// (<= 8192 candidates = 96 KiB).  A few microseconds next to a >= 600 us scan.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace rass {

constexpr int kMergeThreads = 1024;

__device__ __forceinline__ bool cand_better(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

__global__ __launch_bounds__(kMergeThreads) void merge_topk_kernel(const float* __restrict__ scores,
                                                                   const int64_t* __restrict__ ids, int n_lists,
                                                                   int nq, int k, int P,
                                                                   float* __restrict__ out_scores,
                                                                   int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int64_t* sid = reinterpret_cast<int64_t*>(smem);       // [P]
    float* ssc = reinterpret_cast<float*>(smem + (size_t)P * 8);  // [P]
    const int q = blockIdx.x;
    const int n = n_lists * k;
    constexpr int64_t kWorstId = 0x7fffffffffffffffLL;

    for (int e = threadIdx.x; e < P; e += kMergeThreads) {
        float s = -INFINITY;
        int64_t id = kWorstId;
        if (e < n) {
            const int list = e / k, kk = e - list * k;
            const int64_t o = ((int64_t)list * nq + q) * k + kk;
            const int64_t gi = ids[o];
            const float gs = scores[o];
            // NaN and -inf never rank (the scan never emits them; foreign lists might).
            if (gi >= 0 && gs > -INFINITY) {
                s = gs;
                id = gi;
            }
        }
        ssc[e] = s;
        sid[e] = id;
    }

    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (P >> 1); t += kMergeThreads) {
                const int i = 2 * t - (t & (stride - 1));
                const int j = i + stride;
                const bool best_first = (i & size) == 0;
                const float si = ssc[i], sj = ssc[j];
                const int64_t ii = sid[i], ij = sid[j];
                const bool j_better = cand_better(sj, ij, si, ii);
                const bool i_better = cand_better(si, ii, sj, ij);
                if (best_first ? j_better : i_better) {
                    ssc[i] = sj;
                    ssc[j] = si;
                    sid[i] = ij;
                    sid[j] = ii;
                }
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < k) {
        const int e = threadIdx.x;
        const bool filled = (e < P) && sid[e] != kWorstId;
        out_scores[(int64_t)q * k + e] = filled ? ssc[e] : -INFINITY;
        out_ids[(int64_t)q * k + e] = filled ? sid[e] : (int64_t)-1;
    }
}

hipError_t launch_merge_topk(const float* scores, const int64_t* ids, int n_lists, int nq, int k,
                             float* out_scores, int64_t* out_ids, hipStream_t stream) {
    const int n = n_lists * k;
    if (n_lists < 1 || nq < 1 || k < 1 || n > kMergeMaxCandidates) return hipErrorInvalidValue;
    int P = 2;
    while (P < n) P <<= 1;
    if (P < k) {
        while (P < k) P <<= 1;
    }
    const size_t lds_bytes = (size_t)P * 12;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&merge_topk_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kMergeMaxCandidates * 12);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(kMergeThreads), lds_bytes, stream, scores, ids,
                       n_lists, nq, k, P, out_scores, out_ids);
    return hipGetLastError();
}

}  // namespace rass
