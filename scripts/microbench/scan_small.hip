// scan_small.hip — what does ONE launch of the fused scan cost when the slab is small (an IVF's coarse scan over nlist centroids,
// a fine scan at nprobe 1)?  The product kernel included as source with RASS_SCAN_CLOCKS (per-workgroup start / end wall clocks):
// event time of a launch against the in-kernel time of its slowest workgroup, for a few slab sizes and 16 / 32 queries.
// Build: hipcc -O3 --offload-arch=gfx950 -o scan_small.bin scan_small.hip
#define RASS_SCAN_CLOCKS 1
#include "../../rassengine_amd/csrc/scan_topk.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(float* x, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 2654435761u; h ^= h >> 16;
        x[i] = ((float)(h >> 8) * (1.f / 8388608.f) - 1.f) * 0.03125f;
    }
}

int main() {
    const int64_t stride = 1024;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    const int max_rows = 65536;
    float *X, *Q, *ps; int64_t* pi;
    CK(hipMalloc(&X, (size_t)max_rows * stride * 4));
    CK(hipMalloc(&Q, 32 * stride * 4));
    CK(hipMalloc(&ps, (size_t)n_cus * 32 * 32 * 4));
    CK(hipMalloc(&pi, (size_t)n_cus * 32 * 32 * 8));
    fill_kernel<<<4096, 256>>>(X, (size_t)max_rows * stride, 1u);
    fill_kernel<<<64, 256>>>(Q, (size_t)32 * stride, 7u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> clk(2 * 1024);
    for (int n_rows : {32, 64, 4096, 8192, 65536})
        for (int nq : {16, 32})
            for (int k : {1, 8}) {
                rass::ScanArgs a{};
                a.corpus = X; a.q_padded = Q; a.part_scores = ps; a.part_ids = pi; a.row_stride = stride;
                a.n_rows = n_rows; a.nq = nq; a.k = k;
                const int grid = std::min(n_cus, (n_rows + 31) / 32);
                double ev = 0, in_k = 0;
                const int reps = 20;
                for (int r = 0; r < reps + 3; ++r) {
                    CK(hipEventRecord(e0, 0));
                    CK(rass::launch_scan_topk_f32(a, grid, 0));
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r < 3) continue;
                    CK(hipMemcpyFromSymbol(clk.data(), HIP_SYMBOL(rass::g_scan_clocks), (size_t)2 * grid * 8));
                    unsigned long long s0 = ~0ull, e = 0;
                    for (int b = 0; b < grid; ++b) { s0 = std::min(s0, clk[2 * b]); e = std::max(e, clk[2 * b + 1]); }
                    ev += ms * 1e3; in_k += (e - s0) * 0.01;
                }
                printf("rows %6d  nq %2d  k %d  grid %3d: launch (events) %6.1f us, first start -> last end inside the kernel %6.1f us\n", n_rows, nq, k, grid,
                       ev / reps, in_k / reps);
            }
    return 0;
}
