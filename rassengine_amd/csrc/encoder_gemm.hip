// encoder_gemm.hip — K5 of SURVEY §8a: bf16 MFMA GEMM with fused epilogues for the
// sentence encoder (replaces llama.cpp's matmuls behind Ollama's /embeddings,
// reference app/main.py:225-237).
//
//   Y[M, N] = epilogue( X[M, K] (bf16, tokens x features) * W[N, K]^T (bf16, nn.Linear layout) + bias[N] )
//   epilogue: 0 = bias            (QKV projection)
//             1 = bias + residual (attention-out, FFN-down; the sum is formed in fp32)
//             2 = bias + GELU(erf) (FFN-up)
//
// Structure (guide §5, LDS-staged, both operands K-contiguous):
//   * 128 (N) x 128 (M) x 64 (K) block tile, 256 threads = 2x2 waves, each wave 64 x 64 =
//     4 x 4 tiles of v_mfma_f32_16x16x32_bf16; fp32 accumulators (64 VGPRs)
//   * W is the MFMA A operand (rows = output features), X the B operand (cols = tokens): the
//     accumulator then holds 4 CONSECUTIVE output features of one token per tile, so the
//     epilogue reads bias / residual and writes Y as 8-byte pieces along N
//   * staging by global_load_lds (16 B per lane, 1 KiB per wave instruction: 8 rows x 128 B),
//     two LDS buffers, one barrier per K step; the LDS image is lane-linear, the bank-conflict
//     swizzle (16-B chunk c of row r stored at chunk c ^ ((r>>1)&7)) is applied to the global
//     SOURCE address and to the ds_read_b128 address (guide rule 21)
//   * M is padded to 128 by the caller (activations workspace); rows are independent, so
//     padding rows only ever produce padding rows.

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "encoder_kernels.h"

namespace rass {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int kGemmThreads = 256;
constexpr int kTileBytes = 128 * GBK * 2;  // one operand tile: 128 rows x 64 bf16 = 16 KiB

__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {
    // round-to-nearest-even; NaN stays NaN through the plain conversion instruction
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<u16*>(&h);
}

// GELU(x) = x/2 * (1 + erf(x/sqrt2)) with erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7 in
// exact arithmetic, 5e-7 in fp32: two orders below bf16 resolution of the output).  libm's
// erff costs ~3x the epilogue budget: the FFN-up GEMM ran at 510 TF/s with it vs 780 without.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = fmaf(-p, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// Stage one 128 x 64 bf16 operand tile (rows row0.., columns k0..k0+63 of a [rows][ld] matrix)
// into LDS: 16 wave-instructions of 1 KiB; wave w issues pieces w, w+4, w+8, w+12.
__device__ __forceinline__ void stage_tile(const u16* __restrict__ g, int64_t ld, int row0, int k0,
                                           unsigned char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int piece = wave + 4 * p;           // 8 rows per piece
        const int r = piece * 8 + (lane >> 3);    // tile row this lane fills
        const int c_store = lane & 7;             // chunk position in the LDS row (lane-linear)
        const int c_src = c_store ^ ((r >> 1) & 7);
        const u16* src = g + (int64_t)(row0 + r) * ld + k0 + c_src * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const unsigned char* lds_tile, int row, int chunk) {
    const int c = chunk ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + c * 16);
}

template <int EPI>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_kernel(const u16* __restrict__ X, const u16* __restrict__ W,
                                                                   const float* __restrict__ bias,
                                                                   const u16* __restrict__ residual,
                                                                   u16* __restrict__ Y, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2 buf][W tile | X tile]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    // XCD-aware remap: blocks b and b+8 share an L2, so give each XCD a contiguous run of
    // token tiles that re-use the same weight panel (guide T1, bijective form)
    const int nblk = gridDim.x;
    const int orig = blockIdx.x;
    const int q = nblk / 8, rr = nblk % 8, xcd = orig % 8;
    const int bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    const int tiles_n = N / GBN;
    const int bn = bid % tiles_n, bm = bid / tiles_n;
    const int n0 = bn * GBN, m0 = bm * GBM;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / GBK;
    stage_tile(W, K, n0, 0, lds, wave, lane);
    stage_tile(X, K, m0, 0, lds + kTileBytes, wave, lane);
    __syncthreads();  // hipcc drains the pending LDS-DMA (vmcnt(0)) at the barrier
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        unsigned char* buf = lds + cur * 2 * kTileBytes;
        if (t + 1 < nk) {
            unsigned char* nxt = lds + (cur ^ 1) * 2 * kTileBytes;
            stage_tile(W, K, n0, (t + 1) * GBK, nxt, wave, lane);
            stage_tile(X, K, m0, (t + 1) * GBK, nxt + kTileBytes, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = read_frag(buf, wn * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = read_frag(buf + kTileBytes, wm * 64 + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // Epilogue.  acc[i][j]: token m = m0 + wm*64 + j*16 + (lane&15); features
    // n = n0 + wn*64 + i*16 + (lane>>4)*4 + {0,1,2,3}.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
            f32x4 v = acc[i][j] + bv;
            if (EPI == 1) {
                const uint2 r = *reinterpret_cast<const uint2*>(residual + (int64_t)m * N + n);
                v.x += bf16_to_f32((u16)(r.x & 0xffff));
                v.y += bf16_to_f32((u16)(r.x >> 16));
                v.z += bf16_to_f32((u16)(r.y & 0xffff));
                v.w += bf16_to_f32((u16)(r.y >> 16));
            }
            if (EPI == 2) {
                v.x = gelu_erf(v.x);
                v.y = gelu_erf(v.y);
                v.z = gelu_erf(v.z);
                v.w = gelu_erf(v.w);
            }
            uint2 o;
            o.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
            o.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
            *reinterpret_cast<uint2*>(Y + (int64_t)m * N + n) = o;
        }
    }
}

template <int EPI>
static hipError_t launch_epi(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M,
                             int M_pad, int N, int K, hipStream_t stream) {
    constexpr int lds_bytes = 4 * kTileBytes;  // 64 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (N / GBN) * (M_pad / GBM);
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(grid), dim3(kGemmThreads), lds_bytes, stream, X, W, bias,
                       residual, Y, M, N, K);
    return hipGetLastError();
}

hipError_t launch_gemm_bf16(const void* X, const void* W, const float* bias, const void* residual, void* Y, int M,
                            int M_pad, int N, int K, int epilogue, hipStream_t stream) {
    if (M < 0 || M_pad < M || M_pad % GBM != 0 || N % GBN != 0 || K % GBK != 0 || N <= 0 || K <= 0)
        return hipErrorInvalidValue;
    if (M == 0) return hipSuccess;
    const u16* x = static_cast<const u16*>(X);
    const u16* w = static_cast<const u16*>(W);
    const u16* r = static_cast<const u16*>(residual);
    u16* y = static_cast<u16*>(Y);
    switch (epilogue) {
        case 0: return launch_epi<0>(x, w, bias, r, y, M, M_pad, N, K, stream);
        case 1: return r ? launch_epi<1>(x, w, bias, r, y, M, M_pad, N, K, stream) : hipErrorInvalidValue;
        case 2: return launch_epi<2>(x, w, bias, r, y, M, M_pad, N, K, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rass
