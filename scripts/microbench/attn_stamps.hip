// attn_stamps.hip — where do an item's cycles go in attention64_kernel?  Includes the product kernel as source with
// RASS_ATTN_STAMPS (s_memtime stamps at item start, first step, every boundary, last step, item end; per wave) and prints
// the mean split at the cfg-3 shape (256 sequences x 512 tokens x 16 heads).  The stamps cost an lgkmcnt(0) each, placed
// only where the kernel waits anyway.  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -o attn_stamps.bin attn_stamps.hip
#ifndef RASS_ATTN_NO_STAMPS   // -DRASS_ATTN_NO_STAMPS: plain timing of the kernel as built (with any RASS_ATTN_EXP_* macro)
#define RASS_ATTN_STAMPS 1
#endif
#include "../../rassengine_amd/csrc/encoder_attn.hip"

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_bf16_gauss(unsigned short* x, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        float acc = 0.f;
        for (int k = 0; k < 4; ++k) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; acc += (float)(h & 0xffff) * (1.f / 65536.f) - 0.5f; }
        const float v = acc * 1.7320508f;   // ~N(0, 1)
        x[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}

int main(int argc, char** argv) {
    const int nseq = 256, S = 512, heads = 16, hidden = 1024;
    const int T = nseq * S;
    unsigned short *qkv, *ctx; int32_t* cu;
    CK(hipMalloc(&qkv, (size_t)T * 3 * hidden * 2)); CK(hipMalloc(&ctx, (size_t)T * hidden * 2)); CK(hipMalloc(&cu, (nseq + 1) * 4));
    std::vector<int32_t> h(nseq + 1);
    for (int i = 0; i <= nseq; ++i) h[i] = i * S;
    CK(hipMemcpy(cu, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    fill_bf16_gauss<<<4096, 256>>>(qkv, (size_t)T * 3 * hidden, 1);
    CK(hipDeviceSynchronize());
    for (const char* variant : {"w8", "w8f", "w8", "w8f"}) {
        setenv("RASS_ATTN_VARIANT", variant, 1);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0;
        for (int r = 0; r < 60; ++r) CK(rass::launch_attention(qkv, cu, nseq, T, S, hidden, heads, ctx, 0));   // clocks settle (~25 ms)
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 20; ++r) CK(rass::launch_attention(qkv, cu, nseq, T, S, hidden, heads, ctx, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 20;
#ifndef RASS_ATTN_STAMPS
        printf("%s: %.1f us per launch (mean of 20, no stamps)\n", variant, ms * 1e3);
#else
        std::vector<unsigned long long> st(256 * 8 * 9);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(rass::g_attn_stamps), st.size() * 8));
        double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < st.size(); ++i) sum[i % 9] += (double)st[i];
        const double items = sum[5];
        for (int w = 0; w < 8; ++w) {
            double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int b = 0; b < 256; ++b) for (int k = 0; k < 9; ++k) a[k] += (double)st[((size_t)b * 8 + w) * 9 + k];
            printf("   wave %d: total %.0f  prologue %.0f  steps %.0f  (boundaries %.0f)  epilogue %.0f = last products %.0f + vmcnt(0) %.0f + barrier %.0f + store %.0f\n",
                   w, a[0] / a[5], a[1] / a[5], a[2] / a[5], a[3] / a[5], a[4] / a[5], a[6] / a[5], a[7] / a[5], a[8] / a[5],
                   (a[4] - a[6] - a[7] - a[8]) / a[5]);
        }
        printf("%s: %.1f us per launch (with stamps); per item and wave, s_memtime ticks (core clock): total %.1f = prologue %.1f + steps %.1f "
               "(of which boundaries %.1f) + epilogue %.1f;  items per wave %.1f\n", variant, ms * 1e3, sum[0] / items, sum[1] / items,
               sum[2] / items, sum[3] / items, sum[4] / items, items / (256 * 8));
#endif
    }
    return 0;
}
