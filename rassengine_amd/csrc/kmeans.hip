// kmeans.hip — the device side of the IVF build (K9(i) of SURVEY §8a): spherical k-means over rows that are
// already resident in an index's tile16 slab.
//
// Stands in for what the reference delegates to the k-NN plugin at index time (HNSW graph construction,
// app/main.py:563-572): here the coarse structure is nlist centroids + one inverted list per centroid.
//
//   kmeans_assign_f32_kernel   row -> arg max_c <row, centroid_c>, exact fp32 on v_mfma_f32_16x16x4_f32.
//       A GEMM X[n,1024] . C^T[1024,nlist] with a row-argmax epilogue, built from the scan kernel's blocks
//       (scan_core.h): a workgroup takes 32 rows as the resident operand (K split over its 8 waves, fragments
//       loaded STRAIGHT from the tile16 slab — its lane order is the MFMA operand order for either side), streams
//       the centroid slab (nlist x 1024 fp32 = 16.8 MB at 4096 lists: L2 / Infinity-Cache resident) through the
//       same two-register-tile pipeline, meets the 8 K-partials in LDS and keeps ONE running (best score, list)
//       per lane — no top-k list, no insertion; a 32-lane reduction per 32 rows at the end.  Persistent over
//       row blocks.  Bound: fp32 MFMA (2 * nlist * 1024 FLOP per row: 8.6 TFLOP per million rows at 4096 lists).
//   kmeans_accumulate_kernel   sums[list[r]] += row r, counts[list[r]] += 1 (fp32 atomics into the 16.8 MB
//       accumulator, which lives in L2): one pass over the rows, HBM-bound.
//
// The all-reduce of (sums, counts) across ranks, the normalisation of the new centroids and the re-seeding of empty
// lists stay in the Python layer (rassengine_amd/ivf.py): they are O(nlist x dim), not O(rows).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "scan_core.h"

namespace rass {

template <int CH>
__global__ __launch_bounds__(kThreads, 2) void kmeans_assign_f32_kernel(AssignArgs p) {
    constexpr int NT = 2;
    constexpr int NQ = NT * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][kWaves][NQ][kPitch]

    const int lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int n_ct = (p.nlist + kTileRows - 1) / kTileRows;
    const int voff_lane = wid * CH * 1024 + lane * 16;
    const int mt_step = 16 * (int)p.row_stride * 4;

    auto centroid_tile = [&](int t) {
        WorkItem w;
        int rows = p.nlist - t * kTileRows;
        rows = rows < 0 ? 0 : (rows > kTileRows ? kTileRows : rows);
        // wave-uniform by construction: pin them to SGPRs (hipcc otherwise clamps with v_med3 and carries the
        // six live WorkItems of the pipeline in VGPRs the main loop does not have)
        w.tile = __builtin_amdgcn_readfirstlane(t < n_ct ? t : 0);
        w.rows = __builtin_amdgcn_readfirstlane(t < n_ct ? rows : 0);
        w.mask = 0xffffffffu;
        return w;
    };
    auto dump_tile = [&](const f32x4 (&acc)[2][NT], int buf) {
        float* P = lds + buf * (kWaves * NQ * kPitch);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *reinterpret_cast<f32x4*>(P + (wid * NQ + nt * 16 + m) * kPitch + mt * 16 + 4 * g) = acc[mt][nt];
    };

    int pair = 0;
    for (int b = blockIdx.x; b < p.n_blocks; b += gridDim.x) {
        // the 32 rows of this block as the resident B operand: tile16 chunk (block, chunk) IS the fragment
        const int64_t blk = p.first_block + (int64_t)b * p.block_step;
        f32x4 qf[NT][CH];
        {
            const float* base = p.rows + blk * 2 * 16 * p.row_stride + (int64_t)wid * CH * 256 + lane * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool in_slab = (blk * 2 + nt + 1) * 16 <= p.slab_rows;  // a slab holds whole 16-row blocks
#pragma unroll
                for (int j = 0; j < CH; ++j)
                    qf[nt][j] = in_slab ? *reinterpret_cast<const f32x4*>(base + (int64_t)nt * 16 * p.row_stride + j * 256)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // running best of (row q, centroid position r of the tile): pass pq handles row pq*16 + (lane>>5)*8 + wid
        float bs[NT];
        int bi[NT];
#pragma unroll
        for (int pq = 0; pq < NT; ++pq) {
            bs[pq] = -INFINITY;
            bi[pq] = 0x7fffffff;
        }
        auto rank_tile = [&](const WorkItem& w, int buf) {
            const float* P = lds + buf * (kWaves * NQ * kPitch);
            const int r = lane & 31;
            const bool ok = r < w.rows;
#pragma unroll
            for (int pq = 0; pq < NT; ++pq) {
                const int q = pq * 16 + (lane >> 5) * 8 + wid;
                const float* src = P + q * kPitch + r;
                float s = src[0];
#pragma unroll
                for (int wv = 1; wv < kWaves; ++wv) s += src[wv * NQ * kPitch];  // the scan's summation order
                const bool better = ok & (s > bs[pq]);  // tiles ascend: strict > keeps the lowest list on ties
                bs[pq] = better ? s : bs[pq];
                bi[pq] = better ? w.tile * kTileRows + r : bi[pq];
            }
        };

        TileRegs<CH> R0, R1;
        int t = 0;
        WorkItem W0 = centroid_tile(0), W1 = centroid_tile(1);
        issue_tile_loads<CH, false>(R0, make_tile_desc(p.centroids, p.row_stride, nullptr, W0), voff_lane, mt_step);
        issue_tile_loads<CH, false>(R1, make_tile_desc(p.centroids, p.row_stride, nullptr, W1), voff_lane, mt_step);
        __builtin_amdgcn_sched_barrier(0);
        WorkItem Pa{0, 0, 0u}, Pb{0, 0, 0u};
        while (t < n_ct) {
            f32x4 acc[2][NT];
            const WorkItem Wa = W0;
            WorkItem Wn = centroid_tile(t + 2);
            auto rank_prev = [&](int slot) {  // the previous pair is ranked between this pair's MFMA chunks
                if (slot == (CH - 1) / 2) rank_tile(Pa, pair ^ 2);
                if (slot == CH + (CH - 1) / 2) rank_tile(Pb, (pair ^ 2) + 1);
            };
            multiply_and_refill<CH, NT, false>(R0, qf, acc, make_tile_desc(p.centroids, p.row_stride, nullptr, Wn), voff_lane,
                                        mt_step, [&](int j) { rank_prev(j); });
            dump_tile(acc, pair);
            W0 = Wn;
            const WorkItem Wb = W1;
            Wn = centroid_tile(t + 3);
            multiply_and_refill<CH, NT, false>(R1, qf, acc, make_tile_desc(p.centroids, p.row_stride, nullptr, Wn), voff_lane,
                                        mt_step, [&](int j) { rank_prev(CH + j); });
            dump_tile(acc, pair + 1);
            W1 = Wn;
            Pa = Wa;
            Pb = Wb;
            t += 2;
            __syncthreads();
            pair ^= 2;
        }
        rank_tile(Pa, pair ^ 2);
        rank_tile(Pb, (pair ^ 2) + 1);
        // arg max over the 32 centroid positions of each half (score desc, list asc), then one lane writes
#pragma unroll
        for (int pq = 0; pq < NT; ++pq) {
            float s = bs[pq];
            int i = bi[pq];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                const float os = __shfl_xor(s, off, 64);
                const int oi = __shfl_xor(i, off, 64);
                const bool take = (os > s) | ((os == s) & (oi < i));
                s = take ? os : s;
                i = take ? oi : i;
            }
            if ((lane & 31) == 0) {
                const int q = pq * 16 + (lane >> 5) * 8 + wid;
                const int64_t o = (int64_t)b * kTileRows + q;
                p.assign[o] = i == 0x7fffffff ? 0 : i;
                if (p.best) p.best[o] = s;
            }
        }
    }
}

template <int CH>
static hipError_t launch_assign_ch(const AssignArgs& a, int grid, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)4 * kWaves * 2 * 16 * kPitch * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_assign_f32_kernel<CH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((kmeans_assign_f32_kernel<CH>), dim3(grid), dim3(kThreads), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_kmeans_assign_f32(const AssignArgs& a, int n_cus, hipStream_t stream) {
    if (a.row_stride % 128 != 0 || a.row_stride < 128 || a.row_stride > 1024) return hipErrorInvalidValue;
    if (a.n_blocks <= 0) return hipSuccess;
    if (a.nlist < 1 || a.block_step < 1 || a.first_block < 0) return hipErrorInvalidValue;
    const int grid = a.n_blocks < n_cus ? a.n_blocks : n_cus;
    switch ((int)(a.row_stride / 128)) {
        case 1: return launch_assign_ch<1>(a, grid, stream);
        case 2: return launch_assign_ch<2>(a, grid, stream);
        case 3: return launch_assign_ch<3>(a, grid, stream);
        case 4: return launch_assign_ch<4>(a, grid, stream);
        case 5: return launch_assign_ch<5>(a, grid, stream);
        case 6: return launch_assign_ch<6>(a, grid, stream);
        case 7: return launch_assign_ch<7>(a, grid, stream);
        case 8: return launch_assign_ch<8>(a, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

// sums[assign[r]][c] += X[r][c], counts[assign[r]] += 1 over the processed blocks.  One wave per 16-row tile16
// block: a 16-B load per lane is 4 columns of one row; the 4 adds go to that row's list.
__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float* __restrict__ rows, int64_t stride,
                                                                int64_t first_block, int64_t block_step, int n_blocks,
                                                                int64_t n_valid, const int32_t* __restrict__ assign,
                                                                float* __restrict__ sums, float* __restrict__ counts,
                                                                int dim, int nlist) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nchunks = (int)(stride >> 4);
    const int64_t n_half = (int64_t)n_blocks * 2;  // 16-row halves of the processed 32-row blocks
    for (int64_t h = (int64_t)blockIdx.x * 4 + wave; h < n_half; h += (int64_t)gridDim.x * 4) {
        const int64_t b = h >> 1;
        const int64_t out = b * 32 + (h & 1) * 16 + m;          // position in assign[]
        const int64_t blk16 = (first_block + b * block_step) * 2 + (h & 1);
        const int64_t row = blk16 * 16 + m;                     // slab row
        const bool ok = row < n_valid;
        const int list = ok ? assign[out] : 0;
        if (ok && list >= 0 && list < nlist) {
            const float* src = rows + blk16 * 16 * stride + lane * 4;
            float* dst = sums + (int64_t)list * dim + 4 * g;
            for (int j = 0; j < nchunks; ++j) {
                const int c = 16 * j + 4 * g;
                if (c >= dim) break;
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)j * 256);
                atomicAdd(dst + 16 * j + 0, v.x);
                if (c + 1 < dim) atomicAdd(dst + 16 * j + 1, v.y);
                if (c + 2 < dim) atomicAdd(dst + 16 * j + 2, v.z);
                if (c + 3 < dim) atomicAdd(dst + 16 * j + 3, v.w);
            }
            if (g == 0) atomicAdd(counts + list, 1.0f);
        }
    }
}

hipError_t launch_kmeans_accumulate(const float* rows, int64_t stride, int64_t first_block, int64_t block_step,
                                    int n_blocks, int64_t n_valid, const int32_t* assign, float* sums, float* counts,
                                    int dim, int nlist, hipStream_t stream) {
    if (n_blocks <= 0) return hipSuccess;
    int64_t blocks = ((int64_t)n_blocks * 2 + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows, stride, first_block,
                       block_step, n_blocks, n_valid, assign, sums, counts, dim, nlist);
    return hipGetLastError();
}

}  // namespace rass
