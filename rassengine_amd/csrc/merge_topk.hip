// merge_topk.hip — K2 (second stage) and K3's post-all-gather merge of SURVEY §8a.
//
// Input: n_lists candidate lists per query, [n_lists][nq][k] (score f32, id i64; id < 0 =
// empty slot).  Output: [nq][k], best first under (score desc, id asc) — the same total
// order the scan kernel keeps, so merging per-workgroup lists, or per-GPU lists after the
// RCCL all-gather, gives exactly the single-scan result.  This is the analogue of
// OpenSearch's shard -> coordinator top-k merge (reference SHARD_COUNT, app/main.py:89).
//
// Wavefront bitonic top-k: one workgroup per query, 16 waves.  A wave pulls 64 candidates
// at a time, sorts them in registers with a 21-stage shuffle network (no LDS, no barrier),
// and folds them into its running best-32 with a 6-stage bitonic merge (running list in
// lanes 0..31, the new group's best 32 reversed into lanes 32..63).  The 16 per-wave lists
// meet once in LDS and wave 0 folds them the same way.  k <= 32.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace rass {

constexpr int kMergeWaves = 16;
constexpr int kMergeThreads = kMergeWaves * 64;
constexpr int64_t kWorstId = 0x7fffffffffffffffLL;

struct Cand {
    float s;
    int64_t id;
};

__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {
    return (a.s > b.s) || (a.s == b.s && a.id < b.id);
}

__device__ __forceinline__ Cand shfl_xor_cand(const Cand& c, int mask) {
    Cand o;
    o.s = __shfl_xor(c.s, mask, 64);
    const int lo = __shfl_xor((int)(c.id & 0xffffffffLL), mask, 64);
    const int hi = __shfl_xor((int)(c.id >> 32), mask, 64);
    o.id = ((int64_t)hi << 32) | (uint32_t)lo;
    return o;
}

__device__ __forceinline__ Cand shfl_cand(const Cand& c, int src) {
    Cand o;
    o.s = __shfl(c.s, src, 64);
    const int lo = __shfl((int)(c.id & 0xffffffffLL), src, 64);
    const int hi = __shfl((int)(c.id >> 32), src, 64);
    o.id = ((int64_t)hi << 32) | (uint32_t)lo;
    return o;
}

// compare-exchange with lane ^ stride; keep the better one when keep_better
__device__ __forceinline__ void cmpx(Cand& c, int stride, bool keep_better) {
    const Cand o = shfl_xor_cand(c, stride);
    const bool mine_better = cand_better(c, o);
    if (mine_better != keep_better) c = o;
}

// full bitonic sort of the wave's 64 candidates, best first
__device__ __forceinline__ void wave_sort64(Cand& c, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const bool best_first = (lane & size) == 0;  // size == 64: always true
            const bool lower = (lane & stride) == 0;
            cmpx(c, stride, best_first == lower);
        }
    }
}

// lanes 0..31 sorted best-first, lanes 32..63 sorted worst-first (a bitonic sequence)
// -> whole wave sorted best-first
__device__ __forceinline__ void wave_bitonic_merge64(Cand& c, int lane) {
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) cmpx(c, stride, (lane & stride) == 0);
}

// fold a sorted-best-first 64 group (only its best 32 matter) into the running list
__device__ __forceinline__ void fold_group(Cand& run, const Cand& grp_sorted, int lane) {
    // lane 32+i takes the group's element 31-i (reversed), lanes 0..31 keep the running list
    const Cand rev = shfl_cand(grp_sorted, 63 - lane);
    Cand c = (lane < 32) ? run : rev;
    wave_bitonic_merge64(c, lane);
    run = c;
}

__global__ __launch_bounds__(kMergeThreads) void merge_topk_kernel(const float* __restrict__ scores,
                                                                   const int64_t* __restrict__ ids, int n_lists,
                                                                   int /*nq = gridDim.x*/, int k, float* __restrict__ out_scores,
                                                                   int64_t* __restrict__ out_ids,
                                                                   const int64_t* __restrict__ id_map,
                                                                   int64_t score_list_stride, int64_t id_list_stride,
                                                                   MergeGroups grp) {
    __shared__ float sh_s[kMergeWaves * 32];
    __shared__ int64_t sh_i[kMergeWaves * 32];
    // Grouped form (grp.size > 0): blockIdx.x counts the queries of SEVERAL launch groups; group g's lists start
    // grp.*_stride elements after group g-1's and hold min(grp.size, nq_total - g*size) queries each.
    int q = blockIdx.x;
    if (grp.size > 0) {
        const int g = q / grp.size;
        q -= g * grp.size;
        scores += (int64_t)g * grp.score_stride;
        ids += (int64_t)g * grp.id_stride;
        out_scores += (int64_t)g * grp.out_score_stride;
        out_ids += (int64_t)g * grp.out_id_stride;
        if (grp.lists_are_dense) {  // [n_lists][nq_g][k] per group: the list stride follows the group's query count
            const int nq_g = min(grp.size, grp.nq_total - g * grp.size);
            score_list_stride = id_list_stride = (int64_t)nq_g * k;
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = n_lists * k;

    Cand run;
    run.s = -INFINITY;
    run.id = kWorstId;
    // groups of 64 consecutive candidates, dealt round-robin to the waves
    for (int base = wave * 64; base < n; base += kMergeWaves * 64) {
        const int e = base + lane;
        Cand c;
        c.s = -INFINITY;
        c.id = kWorstId;
        if (e < n) {
            const int list = e / k, kk = e - list * k;
            // list strides in elements: nq*k for dense [n_lists][nq][k] arrays, larger when each
            // list is one rank's packed (scores | ids) record of the single all-gather
            const int64_t o = (int64_t)q * k + kk;
            const int64_t gi = ids[(int64_t)list * id_list_stride + o];
            const float gs = scores[(int64_t)list * score_list_stride + o];
            // NaN and -inf never rank (the scan never emits them; foreign lists might)
            if (gi >= 0 && gs > -INFINITY) {
                c.s = gs;
                c.id = gi;
            }
        }
        wave_sort64(c, lane);
        fold_group(run, c, lane);
    }
    // tree over the 16 per-wave lists (each sorted best-first in lanes 0..31): in round r the
    // waves with wave % (2<<r) == 0 fold in the list of wave + (1<<r), handed over through LDS
    Cand acc = run;
#pragma unroll 1
    for (int step = 1; step < kMergeWaves; step <<= 1) {
        const bool sender = (wave & (2 * step - 1)) == step;
        const bool receiver = (wave & (2 * step - 1)) == 0;
        if (sender && lane < 32) {
            sh_s[wave * 32 + lane] = acc.s;
            sh_i[wave * 32 + lane] = acc.id;
        }
        __syncthreads();
        if (receiver) {
            // lanes 32..63 take the partner's list reversed: lane 32+i <- element 31-i
            Cand c;
            c.s = sh_s[(wave + step) * 32 + (31 - (lane & 31))];
            c.id = sh_i[(wave + step) * 32 + (31 - (lane & 31))];
            if (lane < 32) c = acc;
            wave_bitonic_merge64(c, lane);
            acc = c;
        }
    }
    if (wave != 0) return;
    if (lane < k) {
        const bool filled = acc.id != kWorstId;
        out_scores[(int64_t)q * k + lane] = filled ? acc.s : -INFINITY;
        // id_map (IVF): candidates carry slab positions, callers get their own row ids
        out_ids[(int64_t)q * k + lane] = filled ? (id_map ? id_map[acc.id] : acc.id) : (int64_t)-1;
    }
}

hipError_t launch_merge_topk(const float* scores, const int64_t* ids, int n_lists, int nq, int k,
                             float* out_scores, int64_t* out_ids, hipStream_t stream, const int64_t* id_map,
                             int64_t score_list_stride, int64_t id_list_stride, const MergeGroups* groups) {
    const int64_t n = (int64_t)n_lists * k;
    if (n_lists < 1 || nq < 1 || k < 1 || k > 32 || n > kMergeMaxCandidates) return hipErrorInvalidValue;
    if (score_list_stride <= 0) score_list_stride = (int64_t)nq * k;
    if (id_list_stride <= 0) id_list_stride = (int64_t)nq * k;
    MergeGroups grp;
    if (groups) {
        grp = *groups;
        if (grp.size < 1 || grp.nq_total != nq) return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(kMergeThreads), 0, stream, scores, ids, n_lists, nq, k,
                       out_scores, out_ids, id_map, score_list_stride, id_list_stride, grp);
    return hipGetLastError();
}

}  // namespace rass
