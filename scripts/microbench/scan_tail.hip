// scan_tail.hip — how much of a scan launch is the tail?  Runs the product scan kernel (included
// as source, RASS_SCAN_CLOCKS adds per-workgroup start/end wall clocks) over a synthetic slab and
// prints the spread of workgroup end times, overall and per XCD (blockIdx % 8).
//
// Round-1 finding (MI355X): with the static round-robin tile order the slowest workgroup ends
// ~3 % (B=32) / ~8 % (B=16) after the mean.  Dynamic tile claiming (one device-scope atomicAdd per
// workgroup and tile, issued two tiles ahead) removed the spread (5 us) but made every step
// slower: 680 vs 660 us at B=32, 662 vs 594 us at B=16 — the claim sits in the wave's in-order
// vmcnt queue and 50 same-address atomics/us are slow, so it holds back the loads behind it
// ("pay for the claims, keep the static order": 725 us).  The product kernel keeps the static
// order.  Build: hipcc -O3 --offload-arch=gfx950 -o scan_tail.bin scan_tail.hip
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
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        h *= 2654435761u; h ^= h >> 16;
        // 24 random mantissa bits: data entropy sets the power draw, hence the clock (a first version
        // with 16-bit values ran the chip faster than the bench's N(0,1) rows and over-promised)
        x[i] = ((float)(h >> 8) * (1.f / 8388608.f) - 1.f) * 0.03125f;
    }
}

int main(int argc, char** argv) {
    const int n_rows = argc > 1 ? atoi(argv[1]) : 1000000;
    const int nq = argc > 2 ? atoi(argv[2]) : 32;
    const int reps = 20;
    const int64_t stride = 1024;
    const int k = 10;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    const int64_t rows_alloc = ((int64_t)n_rows + 15) / 16 * 16;
    float *X, *Q, *ps; int64_t* pi;
    CK(hipMalloc(&X, rows_alloc * stride * 4));
    CK(hipMalloc(&Q, 32 * stride * 4));
    CK(hipMalloc(&ps, (size_t)grid * 32 * 32 * 4));
    CK(hipMalloc(&pi, (size_t)grid * 32 * 32 * 8));
    fill_kernel<<<4096, 256>>>(X, (size_t)rows_alloc * stride, 1u);
    fill_kernel<<<64, 256>>>(Q, (size_t)32 * stride, 7u);
    CK(hipDeviceSynchronize());
    rass::ScanArgs a{};
    a.corpus = X; a.q_padded = Q; a.part_scores = ps; a.part_ids = pi; a.row_stride = stride;
    a.n_rows = n_rows; a.nq = nq; a.k = k;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> clk(2 * grid);
    const int skews[] = {0, 5, 6, 7, 8, 10, 14, 30};
    for (int mode = 0; mode < (int)(sizeof(skews) / sizeof(int)); ++mode) {
        a.xcd_skew = skews[mode];
        double best = 1e30, sum = 0;
        std::vector<double> spread_mean, spread_max;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipEventRecord(e0, 0));
            CK(rass::launch_scan_topk_f32(a, grid, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r < 3) continue;
            best = std::min(best, (double)ms); sum += ms;
            CK(hipMemcpyFromSymbol(clk.data(), HIP_SYMBOL(rass::g_scan_clocks), clk.size() * 8));
            if (r == reps + 2) {
                unsigned long long core[2];
                CK(hipMemcpyFromSymbol(core, HIP_SYMBOL(rass::g_scan_core), sizeof(core)));
                printf("  block 0: %llu core ticks in %.1f us -> %.0f MHz\n", core[1] - core[0],
                       (clk[1] - clk[0]) * 0.01, (double)(core[1] - core[0]) / ((clk[1] - clk[0]) * 0.01));
            }
            unsigned long long s0 = ~0ull, s1 = 0, emax = 0; double emean = 0;
            for (int b = 0; b < grid; ++b) { s0 = std::min(s0, clk[2 * b]); s1 = std::max(s1, clk[2 * b]); emax = std::max(emax, clk[2 * b + 1]); }
            std::vector<double> ends;
            for (int b = 0; b < grid; ++b) { ends.push_back((clk[2 * b + 1] - s0) * 0.01); emean += ends.back(); }
            emean /= grid;
            std::sort(ends.begin(), ends.end());
            if (r == reps + 2) {
                for (int x = 0; x < 8; ++x) {
                    double m = 0; int c = 0;
                    for (int b = x; b < grid; b += 8) { m += (clk[2 * b + 1] - s0) * 0.01; ++c; }
                    printf("  xcd %d: mean wg end %.1f us\n", x, m / c);
                }
            }
            if (r == reps + 2)
                printf("  last rep: wg end times (us after first start): min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f; start spread %.1f\n",
                       ends.front(), ends[grid / 10], ends[grid / 2], ends[grid * 9 / 10], ends.back(),
                       (s1 - s0) * 0.01);
            spread_mean.push_back((emax - s0) * 0.01 - emean);
        }
        double sm = 0; for (double v : spread_mean) sm += v;
        printf("skew %2d %s nq=%d rows=%d: avg %.1f us best %.1f us (%.2f TB/s best); max-mean end spread avg %.1f us\n",
               skews[mode], "static", nq, n_rows, sum / reps * 1e3, best * 1e3,
               (double)n_rows * stride * 4 / (best * 1e-3) / 1e12, sm / spread_mean.size());
    }
    return 0;
}
