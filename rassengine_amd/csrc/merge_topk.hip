// merge_topk.hip — K2 (second stage) and K3's post-all-gather merge of SURVEY §8a.
//
// Input: n_lists candidate lists per query, [n_lists][nq][k] (score f32, id i64; id < 0 =
// empty slot).  Output: [nq][k], best first under (score desc, id asc) — the same total
// order the scan kernel keeps, so merging per-workgroup lists, or per-GPU lists after the
// RCCL all-gather, gives exactly the single-scan result.  This is the analogue of
// OpenSearch's shard -> coordinator top-k merge (reference SHARD_COUNT, app/main.py:89).
//
// Wavefront bitonic top-k: one workgroup per query, 16 waves.  A wave pulls 64 candidates
// at a time, sorts them in registers with a 21-stage exchange network (DPP inside a row of 16 lanes, v_permlane16/32_swap
// across rows: no LDS, no barrier; round 3: 16.9 -> 13.9 us per launch against the ds_bpermute form),
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

// One dword of lane (lane ^ STRIDE), in registers: DPP moves inside a row of 16 lanes (quad permutes for 1 and 2; 4 and 8 as a
// mirror of a mirror: half_mirror(i) = i ^ 7, quad_reverse(i) = i ^ 3, row_mirror(i) = i ^ 15), v_permlane16_swap /
// v_permlane32_swap across rows (swap(x, x) leaves the even rows / the lower half of x in every row of the first result and
// the odd rows / the upper half in the second: a lane's partner value is in the result its own row does not name).  Round 3:
// the 34 compare-exchange stages of a merge were 102 ds_bpermute round trips through the LDS crossbar.
template <int STRIDE>
__device__ __forceinline__ int xor_lane(int v, int lane) {
    static_assert(STRIDE == 1 || STRIDE == 2 || STRIDE == 4 || STRIDE == 8 || STRIDE == 16 || STRIDE == 32, "stride");
    if constexpr (STRIDE == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
    if constexpr (STRIDE == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
    if constexpr (STRIDE == 4)
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x1B, 0xf, 0xf, true), 0x141, 0xf, 0xf, true);
    if constexpr (STRIDE == 8)
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true), 0x140, 0xf, 0xf, true);
    if constexpr (STRIDE == 16) {
        const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((lane & 16) ? sw[0] : sw[1]);
    }
    if constexpr (STRIDE == 32) {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((lane & 32) ? sw[0] : sw[1]);
    }
    return v;
}

template <int STRIDE>
__device__ __forceinline__ Cand xor_cand(const Cand& c, int lane) {
    Cand o;
    o.s = __int_as_float(xor_lane<STRIDE>(__float_as_int(c.s), lane));
    const int lo = xor_lane<STRIDE>((int)(c.id & 0xffffffffLL), lane);
    const int hi = xor_lane<STRIDE>((int)(c.id >> 32), lane);
    o.id = ((int64_t)hi << 32) | (uint32_t)lo;
    return o;
}

// lane (63 - lane) = lane ^ 63: across the halves, across the rows, mirrored inside the row
__device__ __forceinline__ Cand reverse_cand(const Cand& c, int lane) {
    Cand o = xor_cand<32>(c, lane);
    o = xor_cand<16>(o, lane);
    o.s = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(o.s), 0x140, 0xf, 0xf, true));
    const int lo = __builtin_amdgcn_mov_dpp((int)(o.id & 0xffffffffLL), 0x140, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(o.id >> 32), 0x140, 0xf, 0xf, true);
    o.id = ((int64_t)hi << 32) | (uint32_t)lo;
    return o;
}

// compare-exchange with lane ^ STRIDE; keep the better one when keep_better
template <int STRIDE>
__device__ __forceinline__ void cmpx(Cand& c, int lane, bool keep_better) {
    const Cand o = xor_cand<STRIDE>(c, lane);
    const bool mine_better = cand_better(c, o);
    if (mine_better != keep_better) c = o;
}

template <int SIZE, int STRIDE>
__device__ __forceinline__ void sort_steps(Cand& c, int lane) {
    if constexpr (STRIDE > 0) {
        const bool best_first = (lane & SIZE) == 0;  // SIZE == 64: always true
        const bool lower = (lane & STRIDE) == 0;
        cmpx<STRIDE>(c, lane, best_first == lower);
        sort_steps<SIZE, STRIDE / 2>(c, lane);
    }
}

// full bitonic sort of the wave's 64 candidates, best first
__device__ __forceinline__ void wave_sort64(Cand& c, int lane) {
    sort_steps<2, 1>(c, lane);
    sort_steps<4, 2>(c, lane);
    sort_steps<8, 4>(c, lane);
    sort_steps<16, 8>(c, lane);
    sort_steps<32, 16>(c, lane);
    sort_steps<64, 32>(c, lane);
}

// lanes 0..31 sorted best-first, lanes 32..63 sorted worst-first (a bitonic sequence)
// -> whole wave sorted best-first
template <int STRIDE>
__device__ __forceinline__ void merge_steps(Cand& c, int lane) {
    if constexpr (STRIDE > 0) {
        cmpx<STRIDE>(c, lane, (lane & STRIDE) == 0);
        merge_steps<STRIDE / 2>(c, lane);
    }
}
__device__ __forceinline__ void wave_bitonic_merge64(Cand& c, int lane) { merge_steps<32>(c, lane); }

// fold a sorted-best-first 64 group (only its best 32 matter) into the running list
__device__ __forceinline__ void fold_group(Cand& run, const Cand& grp_sorted, int lane) {
    // lane 32+i takes the group's element 31-i (reversed), lanes 0..31 keep the running list
    const Cand rev = reverse_cand(grp_sorted, lane);
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
    // groups of 64 consecutive candidates, dealt round-robin to the waves.  (Issuing a group's scattered loads a group ahead of
    // its sort changed nothing, 14.3 against 13.9 us per launch: with the exchanges in registers the kernel is paced by the
    // vector issue of its sixteen waves on one CU, ~1 900 instructions each.)
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
