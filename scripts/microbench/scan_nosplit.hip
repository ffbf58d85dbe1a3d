// scan_nosplit.hip — TIMING PROTOTYPE (not product code, no top-k): how fast could the B = 32 exact
// fp32 scan stream if the K axis were NOT split over the waves of a workgroup?
//
// Product kernel (csrc/scan_topk.hip): 8 waves split K, every tile needs partial dumps + a barrier,
// the query fragments live in registers (64 VGPRs) and each wave's loads are re-issued as its MFMAs
// free registers.  Measured: HBM 76 %, MFMA pipe 66 % busy — neither saturated.
// Prototype: the normalised queries live in LDS in B-fragment order (128 KiB); every wave streams
// whole 16-row blocks (64 KiB contiguous in the tile16 layout) through a register ring of RING
// 1-KiB chunks, reads the two B fragments of a chunk from LDS, runs the 8 MFMAs of the chunk as one
// fmaf chain over k (two accumulators), and keeps a running max per lane as a stand-in for top-k.
// No barrier, no partial dump, no cross-wave reduction inside the loop.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scan_nosplit.bin scan_nosplit.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kWaves = 8;
constexpr int kDim = 1024;
constexpr int kChunks = kDim / 16;  // 64 chunks of 16 columns per 16-row block

__global__ void fill_kernel(float* x, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 2654435761u; h ^= h >> 16;
        x[i] = ((float)(h >> 8) * (1.f / 8388608.f) - 1.f) * 0.03125f;  // 24 random mantissa bits
    }
}

template <int RING, int NT>
__global__ __launch_bounds__(kWaves * 64, 2) void scan_nosplit_kernel(const float* __restrict__ X, int64_t n_blocks,
                                                                    const float* __restrict__ Qfrag,
                                                                    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [kChunks][NT][64 lanes] f32x4
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < kChunks * NT * 64; e += kWaves * 64)
        reinterpret_cast<f32x4*>(lds)[e] = reinterpret_cast<const f32x4*>(Qfrag)[e];
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * kWaves + wave, stride = (int64_t)gridDim.x * kWaves;
    float best = -1e30f;
    f32x4 ring[RING];
    // stream position: (block, chunk) advances chunk-major inside a block, then to the wave's next block
    int64_t ld_block = gw;
    int ld_chunk = 0;
    auto issue = [&](int slot) {
        // unconditional (straight-line code keeps hipcc's vmcnt counts exact): the run-ahead past the
        // wave's last block re-reads that block, the values are never used
        const int64_t blk = ld_block < n_blocks ? ld_block : n_blocks - 1;
        const float* src = X + blk * (16 * kDim) + ld_chunk * 256 + lane * 4;
        ring[slot] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
        if (++ld_chunk == kChunks) {
            ld_chunk = 0;
            ld_block += stride;
        }
    };
#pragma unroll
    for (int s = 0; s < RING; ++s) issue(s);

    for (int64_t b = gw; b < n_blocks; b += stride) {
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < kChunks; j0 += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                const int j = j0 + s;
                const f32x4* qf = reinterpret_cast<const f32x4*>(lds) + (j * NT) * 64 + lane;
                f32x4 q[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) q[nt] = qf[nt * 64];
                const f32x4 a = ring[s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 c = acc[nt];
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, q[nt].x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, q[nt].y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, q[nt].z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, q[nt].w, c, 0, 0, 0);
                    acc[nt] = c;
                }
                issue(s);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            best = fmaxf(best, fmaxf(fmaxf(acc[nt].x, acc[nt].y), fmaxf(acc[nt].z, acc[nt].w)));
    }
    out[blockIdx.x * (kWaves * 64) + threadIdx.x] = best;
}

template <int RING, int NT>
static int run(const float* X, int64_t n_rows, const float* Q, float* out, int grid) {
    const size_t lds_bytes = (size_t)kChunks * NT * 64 * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_nosplit_kernel<RING, NT>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r) scan_nosplit_kernel<RING, NT><<<grid, kWaves * 64, lds_bytes>>>(X, n_rows / 16, Q, out);
    CK(hipDeviceSynchronize());
    const int reps = 50;  // back to back, like the bench
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) scan_nosplit_kernel<RING, NT><<<grid, kWaves * 64, lds_bytes>>>(X, n_rows / 16, Q, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("RING=%2d chunks (%2d KiB/wave in flight) NT=%d: %.1f us per scan = %.2f TB/s = %.1f %% of 8 TB/s\n", RING, RING, NT, us,
           (double)n_rows * kDim * 4 / (us * 1e-6) / 1e12, (double)n_rows * kDim * 4 / (us * 1e-6) / 8e12 * 100);
    return 0;
}

int main(int argc, char** argv) {
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 1000000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    float *X, *Q, *out;
    CK(hipMalloc(&X, (size_t)(n_rows + 16) * kDim * 4));
    CK(hipMalloc(&Q, (size_t)kChunks * 2 * 64 * 16));
    CK(hipMalloc(&out, (size_t)grid * kWaves * 64 * 4));
    fill_kernel<<<4096, 256>>>(X, (size_t)(n_rows + 16) * kDim, 1u);
    fill_kernel<<<64, 256>>>(Q, (size_t)kChunks * 2 * 64 * 4, 7u);
    CK(hipDeviceSynchronize());
    if (run<16, 2>(X, n_rows, Q, out, grid)) return 1;
    if (run<32, 2>(X, n_rows, Q, out, grid)) return 1;
    if (run<32, 1>(X, n_rows, Q, out, grid)) return 1;
    if (run<16, 1>(X, n_rows, Q, out, grid)) return 1;
    return 0;
}
