// What one gfx950 SIMD issues per cycle for the vector instructions of the attention inner loop, measured: a wave64 runs
// a long loop of 16 independent instructions of ONE kind; one or two waves per SIMD; cycles per instruction from
// s_memtime (constant 100 MHz reference) against v_fma_f32 in the same run, and as a ratio to the wall time.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rates scripts/microbench/valu_rates.hip && /tmp/valu_rates
//
// Modes: 0 v_fma_f32, 1 v_exp_f32, 2 v_pk_fma_f32, 3 v_cvt_pk_bf16_f32, 4 v_max_f32, 5 v_add_f32, 6 v_pk_add_f32,
//        7 v_pk_mul_f32, 8 v_rcp_f32, 9 v_mfma_f32_32x32x16_bf16 (independent accumulators)
// A launch gives waves 0-3 of each block (one per SIMD) mode A and waves 4-7 mode B (-1 = those waves exit at once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__device__ __forceinline__ float body(int iters, float seed) {
    float r[16];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{seed + i, seed - i};
    if constexpr (MODE == 9) {
        f32x16 acc[4];
        bf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)seed; b[i] = (__bf16)(seed + 1.f); }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
        return s;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
            if constexpr (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i & 7]));
            if constexpr (MODE == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
            if constexpr (MODE == 4) asm volatile("v_max_f32 %0, %0, %0" : "+v"(r[i]));
            if constexpr (MODE == 5) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
            if constexpr (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i & 7]));
            if constexpr (MODE == 7) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i & 7]));
            if constexpr (MODE == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    return s;
}

// One MFMA followed by NF v_fma_f32 and NE v_exp_f32 (all independent of the MFMA's result), four such groups per iteration on
// four accumulators; CHAIN: the four MFMAs of an iteration accumulate into ONE tile (the QK^T chain of the attention block).
template <int NF, int NE, bool CHAIN>
__device__ __forceinline__ float mixed(int iters, float seed) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed + i;
    f32x16 acc[4];
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)seed; b[i] = (__bf16)(seed + 1.f); }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            constexpr int kZero = 0;
            const int t = CHAIN ? kZero : j;
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[(j * 4 + i) & 15]));
#pragma unroll
            for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[(j * 4 + NF + i) & 15]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    return s;
}

__device__ __forceinline__ float run_mode(int mode, int iters, float seed) {
    switch (mode) {
        case 0: return body<0>(iters, seed);
        case 1: return body<1>(iters, seed);
        case 2: return body<2>(iters, seed);
        case 3: return body<3>(iters, seed);
        case 4: return body<4>(iters, seed);
        case 5: return body<5>(iters, seed);
        case 6: return body<6>(iters, seed);
        case 7: return body<7>(iters, seed);
        case 8: return body<8>(iters, seed);
        case 9: return body<9>(iters, seed);
        case 10: return mixed<0, 0, false>(iters, seed);
        case 11: return mixed<2, 0, false>(iters, seed);
        case 12: return mixed<4, 0, false>(iters, seed);
        case 13: return mixed<6, 0, false>(iters, seed);
        case 14: return mixed<8, 0, false>(iters, seed);
        case 15: return mixed<12, 0, false>(iters, seed);
        case 16: return mixed<6, 2, false>(iters, seed);
        case 17: return mixed<0, 0, true>(iters, seed);
        case 18: return mixed<6, 0, true>(iters, seed);
        case 19: return mixed<6, 2, true>(iters, seed);
        case 20: return mixed<9, 2, false>(iters, seed);
    }
    return 0.f;
}

__global__ __launch_bounds__(1024) void rates_kernel(int mode_a, int mode_b, int iters, float* sink, long long* clocks, long long* all16) {
    const int wave = threadIdx.x >> 6;
    const int mode = wave < 4 ? mode_a : mode_b;
    if (mode < 0) return;
    const long long t0 = wall_clock64(), c0 = clock64();
    const float s = run_mode(mode, iters, 1.0f + threadIdx.x * 1e-9f);
    const long long t1 = wall_clock64(), c1 = clock64();
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && wave < 8) clocks[blockIdx.x * 8 + wave] = t1 - t0;
    if ((threadIdx.x & 63) == 0) all16[blockIdx.x * 16 + wave] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) clocks[gridDim.x * 8] = c1 - c0;   // s_memtime ticks of one wave, for the clock ratio
}

// Sustained MFMA-only streams with pseudo-random bf16 operands that differ per lane and per instruction (toggling operand
// buses like a real GEMM does): which clock does the chip hold for each MFMA shape?  SHAPE 0: v_mfma_f32_16x16x32_bf16 on 16
// accumulator tiles, SHAPE 1: v_mfma_f32_32x32x16_bf16 on 4 (the same flops per iteration: 16 x 16 384 = 8 x 32 768).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_power_kernel(int iters, float* sink) {
    bf16x8 a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            a[i][j] = (__bf16)(((float)(h & 0xffff) - 32768.f) * (1.f / 32768.f));
            b[i][j] = (__bf16)(((float)(h >> 16) - 32768.f) * (1.f / 32768.f));
        }
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(i + 2 * u) & 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0];
    }
    if (s == 123.456f) sink[0] = s;
}

static const char* kNames[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_max_f32", "v_add_f32",
                               "v_pk_add_f32", "v_pk_mul_f32", "v_rcp_f32", "v_mfma_f32_32x32x16_bf16"};

int main() {
    const int blocks = 256, iters = 20000;
    float* sink; long long* d_clocks; long long* d_all16;
    hipMalloc(&d_all16, 256 * 16 * sizeof(long long));
    hipMalloc(&sink, 4); hipMalloc(&d_clocks, (blocks * 8 + 1) * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double memtime_per_wall = 0;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(rates_kernel, dim3(blocks), dim3(512), 0, 0, 0, 0, iters, sink, d_clocks, d_all16);   // ~0.4 s: clocks up
    hipDeviceSynchronize();
    auto launch = [&](int a, int b, double* wave_a_ticks, double* wave_b_ticks) {
        hipMemset(d_clocks, 0, blocks * 8 * sizeof(long long));
        hipLaunchKernelGGL(rates_kernel, dim3(blocks), dim3(512), 0, 0, a, b, iters, sink, d_clocks, d_all16);   // warm
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rates_kernel, dim3(blocks), dim3(512), 0, 0, a, b, iters, sink, d_clocks, d_all16);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 8 + 1);
        hipMemcpy(h.data(), d_clocks, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double sa = 0, sb = 0;
        for (int i = 0; i < blocks; ++i) { for (int w = 0; w < 4; ++w) sa += h[i * 8 + w]; for (int w = 4; w < 8; ++w) sb += h[i * 8 + w]; }
        *wave_a_ticks = sa / (blocks * 4); *wave_b_ticks = sb / (blocks * 4);
        memtime_per_wall = (double)h[blocks * 8] / (double)h[0];
        return (double)ms;
    };
    // the wall clock ticks at 100 MHz; the core clock is unknown and load-dependent, so everything is quoted against
    // v_fma_f32 alone on a SIMD (4 cycles per wave64 instruction on a 16-lane SIMD)
    double ta, tb;
    const double ms_fma = launch(0, -1, &ta, &tb);
    const double ref = ta;
    const double insts = 16.0 * iters;
    printf("one wave per SIMD, %d x 16 independent instructions; cycles per instruction assume v_fma_f32 = 4.00\n", iters);
    printf("  (v_fma_f32: %.3f ms per launch, %.2f ns per instruction => core clock %.2f GHz if 4 cycles; s_memtime ticks per 10 ns: %.3f)\n", ms_fma,
           ref * 10.0 / insts, 4.0 / (ref * 10.0 / insts), memtime_per_wall);
    for (int m = 0; m < 10; ++m) {
        launch(m, -1, &ta, &tb);
        printf("  %-26s %6.2f cycles per instruction\n", kNames[m], 4.0 * ta / ref);
    }
    printf("two waves per SIMD (A on waves 0-3, B on waves 4-7), cycles per instruction PAIR (one A + one B):\n");
    const int pairs[][2] = {{0, 0}, {1, 1}, {0, 1}, {2, 1}, {3, 1}, {9, 0}, {9, 1}, {9, 9}, {2, 2}};
    for (auto& pr : pairs) {
        launch(pr[0], pr[1], &ta, &tb);
        printf("  %-26s + %-26s  A %6.2f  B %6.2f  (the slower of the two paces the pair)\n", kNames[pr[0]], kNames[pr[1]], 4.0 * ta / ref,
               4.0 * tb / ref);
    }
    printf("SIMD throughput against the number of waves per SIMD (all waves the same stream): cycles per instruction (group) per SIMD\n");
    for (int m : {0, 1, 2, 9, 13, 16, 20}) {
        printf("  %-40s", m < 10 ? kNames[m] : (m == 13 ? "mfma + 6 fma" : m == 16 ? "mfma + 6 fma + 2 exp" : "mfma + 9 fma + 2 exp"));
        for (int wps : {1, 2, 3, 4}) {
            hipMemset(d_all16, 0, blocks * 16 * sizeof(long long));
            hipLaunchKernelGGL(rates_kernel, dim3(blocks), dim3(256 * wps), 0, 0, m, wps > 1 ? m : -1, iters, sink, d_clocks, d_all16);
            hipDeviceSynchronize();
            std::vector<long long> h(blocks * 16);
            hipMemcpy(h.data(), d_all16, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            // the slowest wave of a block = when its SIMDs had finished everything (all waves start together)
            double worst = 0;
            for (int i = 0; i < blocks; ++i) { long long mx = 0; for (int w = 0; w < 16; ++w) mx = h[i * 16 + w] > mx ? h[i * 16 + w] : mx; worst += mx; }
            worst /= blocks;
            const double per = (m < 10 ? 4.0 : 16.0) * worst / ref / wps;
            printf("  %dw %6.2f", wps, per);
        }
        printf("\n");
    }
    {   // sustained, every CU: nothing but back-to-back MFMAs (two waves per SIMD) for ~0.3 s: what clock does the chip hold?
        const int n = 40;
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(rates_kernel, dim3(blocks), dim3(512), 0, 0, 9, 9, iters, sink, d_clocks, d_all16);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)n * blocks * 8 * 16.0 * iters;
        const double tflops = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
        // a SIMD retires one 32x32x16 MFMA per 32 cycles: clock = MFMAs per SIMD / time * 32
        const double ghz = (double)n * 2 * 16.0 * iters * 32.0 / (ms * 1e-3) / 1e9;
        printf("sustained MFMA only, %d CUs x 8 waves, %.0f ms: %.0f TFLOP/s bf16 = a core clock of %.2f GHz (spec peak 2 500 TFLOP/s = 2.4 GHz)\n",
               blocks, ms, tflops, ghz);
    }
    for (int shape = 0; shape < 2; ++shape) {
        const int n = 40, it2 = 2 * iters;
        for (int rep = 0; rep < 2; ++rep) {   // second pass is the one reported (first warms the power state)
            hipEventRecord(e0, 0);
            for (int i = 0; i < n; ++i) {
                if (shape == 0) hipLaunchKernelGGL(mfma_power_kernel<0>, dim3(blocks), dim3(512), 0, 0, it2, sink);
                else hipLaunchKernelGGL(mfma_power_kernel<1>, dim3(blocks), dim3(512), 0, 0, it2, sink);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)n * blocks * 8 * (double)it2 * 16.0 * 16384.0;
        printf("sustained %s, pseudo-random operands, %d CUs x 8 waves, %.0f ms: %.0f TFLOP/s = %.2f GHz\n",
               shape == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", blocks, ms, flops / (ms * 1e-3) / 1e12,
               flops / (ms * 1e-3) / 1e12 / 2500.0 * 2.4);
    }
    printf("one MFMA + NF v_fma_f32 + NE v_exp_f32 per group (asm order kept), cycles per GROUP; one wave per SIMD | two waves per SIMD (each)\n");
    const char* mixed_names[] = {"mfma", "mfma + 2 fma", "mfma + 4 fma", "mfma + 6 fma", "mfma + 8 fma", "mfma + 12 fma", "mfma + 6 fma + 2 exp",
                                 "mfma (one accumulator)", "mfma (one accumulator) + 6 fma", "mfma (one accumulator) + 6 fma + 2 exp",
                                 "mfma + 9 fma + 2 exp"};
    for (int m = 10; m <= 20; ++m) {
        double a1, b1, a2, b2;
        launch(m, -1, &a1, &b1);
        launch(m, m, &a2, &b2);
        // a group is a quarter of an iteration: 4 groups x iters per launch against 16 x iters instructions of the reference
        printf("  %-40s %6.1f | %6.1f %6.1f\n", mixed_names[m - 10], 16.0 * a1 / ref, 16.0 * a2 / ref, 16.0 * b2 / ref);
    }
    return 0;
}
