// gemm_phases.hip — where does a 256x256 tile of the ring GEMM spend its time?  Includes the ring kernels as
// source (round 1's kernels, retired from the product in round 3: gemm_retired_kernels.hip; RASS_GEMM_CLOCKS adds four
// wall-clock stamps per block: start, after the pipeline prologue, after the K loop, after the epilogue stores have
// drained) and prints the mean phase lengths.  RASS_GEMM_VARIANT=ring | pring (default pring) picks the kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_phases.bin gemm_phases.hip ../../rassengine_amd/csrc/encoder_misc.hip
#define RASS_GEMM_CLOCKS 1
#define RASS_RETIRED_NO_MAIN 1
#include "gemm_retired_kernels.hip"

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_bf16(unsigned short* x, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((float)(h & 0xffff) - 32768.f) * (1.f / 32768.f);
        x[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}

int main(int argc, char**) {
    const int M = 131072;
    // the encoder's four GEMMs, then (with any argument) the epilogue / N cross terms
    const int shapes[8][3] = {{3072, 1024, 0}, {1024, 1024, 1}, {4096, 1024, 2}, {1024, 4096, 1},
                              {1024, 1024, 0}, {1024, 1024, 2}, {4096, 1024, 1}, {4096, 1024, 0}};
    const int n_shapes = argc > 1 ? 8 : 4;
    for (int si = 0; si < n_shapes; ++si) {
        const int* sh = shapes[si];
        const int N = sh[0], K = sh[1], epi = sh[2];
        unsigned short *X, *W, *R, *Y; float* b;
        CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2));
        CK(hipMalloc(&R, (size_t)M * N * 2)); CK(hipMalloc(&Y, (size_t)M * N * 2)); CK(hipMalloc(&b, N * 4));
        fill_bf16<<<4096, 256>>>(X, (size_t)M * K, 1); fill_bf16<<<1024, 256>>>(W, (size_t)N * K, 2);
        fill_bf16<<<4096, 256>>>(R, (size_t)M * N, 3); CK(hipMemset(b, 0, N * 4));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            CK(rass::launch_retired(getenv("RASS_GEMM_VARIANT") ? getenv("RASS_GEMM_VARIANT") : "pring", X, W, b, R, Y, M, M, N, K, epi, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const int grid = (N / 256) * (M / 256);
        std::vector<unsigned long long> c(4 * (size_t)grid);
        CK(hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(rass::g_gemm_clocks), c.size() * 8));
        unsigned long long core[64];
        CK(hipMemcpyFromSymbol(core, HIP_SYMBOL(rass::g_gemm_core_cycles), sizeof(core)));
        double cyc = 0, lp = 0;
        for (int i = 0; i < 64; ++i) { cyc += (double)core[i]; lp += (c[4 * i + 2] - c[4 * i + 1]) * 0.01; }
        printf("  K loop of blocks 0-63: %.0f s_memtime ticks in %.1f us -> %.0f MHz; per K step %.0f ticks\n", cyc / 64, lp / 64,
               cyc / lp, cyc / 64 / (K / 32));
#ifdef RASS_GEMM_PHASE_TIMERS
        {
            unsigned long long ph[64 * 2 * 4];
            CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(rass::g_gemm_phase_cycles), sizeof(ph)));
            for (int g = 0; g < 2; ++g) {
                double v[4] = {0, 0, 0, 0};
                for (int b = 0; b < 64; ++b) for (int x = 0; x < 4; ++x) v[x] += (double)ph[(b * 2 + g) * 4 + x];
                const double steps = 64.0 * (K / 32);
                printf("  group %c per K step (s_memtime ticks, timers perturb): load %.0f  barrier1 %.0f  compute+wait %.0f  barrier2 %.0f\n",
                       g ? 'B' : 'A', v[0] / steps, v[1] / steps, v[2] / steps, v[3] / steps);
            }
        }
#endif
        double pro = 0, loop = 0, ep = 0, tot = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < grid; ++i) {
            pro += (c[4 * i + 1] - c[4 * i]) * 0.01; loop += (c[4 * i + 2] - c[4 * i + 1]) * 0.01;
            ep += (c[4 * i + 3] - c[4 * i + 2]) * 0.01; tot += (c[4 * i + 3] - c[4 * i]) * 0.01;
            t0 = c[4 * i] < t0 ? c[4 * i] : t0; t1 = c[4 * i + 3] > t1 ? c[4 * i + 3] : t1;
        }
        printf("N=%d K=%d epi=%d: %.0f us (%.0f TF/s), %d tiles, %.1f rounds; per tile: prologue %.1f us, K loop %.1f us, epilogue %.1f us, "
               "total %.1f us; wall/rounds %.1f us\n", N, K, epi, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, grid, grid / 256.0,
               pro / grid, loop / grid, ep / grid, tot / grid, (t1 - t0) * 0.01 / (grid / 256.0));
        (void)hipFree(X); (void)hipFree(W); (void)hipFree(R); (void)hipFree(Y); (void)hipFree(b);
    }
    return 0;
}
