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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "encoder_kernels.h"

namespace rass {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int kGemmThreads = 256;
constexpr int kTileBytes = 128 * GBK * 2;  // one operand tile: 128 rows x 64 bf16 = 16 KiB

__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {
    // round-to-nearest-even; NaN stays NaN through the plain conversion instruction
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<u16*>(&h);
}

// GELU(x) = x/2 * (1 + erf(x/sqrt2)) (the erf form of BERT's "gelu").  libm's erff costs ~3x the epilogue budget (the FFN-up
// GEMM ran at 510 TF/s with it vs 780 without); rounds 1-3 used Abramowitz-Stegun 7.1.26 (a reciprocal and an exponential).
__device__ __forceinline__ float gelu_erf(float x) {
    // With h = |x|/2 and z = |x|/sqrt2:  GELU(x) = max(x, 0) - h * erfc(z).  Round 4: erfc(z) = exp2(Q(h)), Q the degree-7
    // least-squares fit of log2(erfc(h sqrt2)) on z in [0, 5] (|rel. error| of erfc < 1.2e-5, so |error| of GELU < 1.5e-6
    // everywhere, two orders below the bf16 resolution of the output; beyond z = 5 erfc < 2e-12 and h is clamped): one
    // transcendental and 12 plain instructions per element, all of them packable — the Abramowitz-Stegun form it replaces
    // (|error| 2e-7) took 12 + a reciprocal + an exponential, and the GELU epilogue of FFN-up is VALU-bound (8-11 us per
    // 256 x 256 tile, profiles/r04_gemm_epilogue_experiments.txt).  scripts/fit_gelu_poly.py derives and checks the constants.
#ifdef RASS_GELU_AS   // rounds 1-3 (the A/B build): Abramowitz-Stegun 7.1.26, 1 - erf(z) = p(t) exp(-z^2), t = 1 / (1 + 0.3275911 z)
    {
        const float h = 0.5f * fabsf(x);
        const float t = __builtin_amdgcn_rcpf(fmaf(0.46328375849f, h, 1.0f));
        float p = fmaf(1.061405429f, t, -1.453152027f);
        p = fmaf(p, t, 1.421413741f);
        p = fmaf(p, t, -0.284496736f);
        p = fmaf(p, t, 0.254829592f);
        p *= t;
        const float zz = h * 1.69864357838f;
        return fmaf(-h, p * __builtin_amdgcn_exp2f(-zz * zz), fmaxf(x, 0.0f));
    }
#endif
    const float h = 0.5f * fabsf(x);
    const float hc = fminf(h, 3.5355339f);
    float q = -2.0300099e-04f;
    q = fmaf(q, hc, 3.5955482e-03f);
    q = fmaf(q, hc, -2.8301010e-02f);
    q = fmaf(q, hc, 1.3302942e-01f);
    q = fmaf(q, hc, -4.2836797e-01f);
    q = fmaf(q, hc, -1.8355303e+00f);
    q = fmaf(q, hc, -2.3021889e+00f);
    q = fmaf(q, hc, -4.7392123e-06f);
    const float e = __builtin_amdgcn_exp2f(q);
    return fmaf(-h, e, fmaxf(x, 0.0f));
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

// ------------------------------------------------------------------------------------------
// "mid" (round 4): gemm_bf16_kernel's 128 x 128 x 64 tile with a FOUR-stage LDS-DMA ring instead of two buffers behind a
// draining barrier.  Shapes too small for the persistent kernels (65 .. ~2 000 rows: the embed micro-batcher's coalesced
// queries, small uploads) are latency-bound, not bandwidth-bound: the two-buffer kernel takes ~1.2 us per 64-deep step (one
// operand tile in flight, its global -> LDS latency exposed every step), and the split-K pair that replaced it in round 2
// (more workgroups, fewer steps each) pays an fp32 partial tile per slice plus a second launch — 12.8 + 5.3 us for the QKV
// projection of 384 tokens.  With three tiles in flight a step is its 32 MFMAs per wave plus one LDS round trip (~0.4 us),
// the epilogue is fused, and K <= 1 024 needs no split: one launch of ~9 us.  Counted waits (vmcnt) and asm fragment reads as
// in p5 (hipcc would drain the DMA queue before every LDS read it can see).
#define RASS_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
constexpr int kMidStages = 4;
constexpr int kMidLdsBytes = kMidStages * 2 * kTileBytes;   // 128 KiB

typedef int mid_i32x4 __attribute__((ext_vector_type(4)));
#define MID_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// The K loop is hand-scheduled like p4's (one wave per SIMD issues in order: whatever is not an MFMA goes, one instruction at
// a time, into the gaps between the MFMAs): a 64-deep step is two sub-steps of 16 MFMAs per wave; under sub-step u run the 8
// fragment reads of sub-step u + 1 and four of the wave's eight DMA pieces of a tile three to four steps ahead (buffer_load
// ... lds on whole-matrix descriptors: an SGPR offset per piece, one VGPR for the lane part); one s_barrier per step, between
// its sub-steps (tile t + 1 is published there and tile t's stage is free from there on).
// BM = token rows per tile (128 or 64): per-workgroup operand traffic (BM + 128) x K x 2 B moves through a latency-bound pipe
// (~4 tiles in flight per CU), so a mid-size batch wants MORE, smaller tiles than CUs it would otherwise leave idle.
template <int EPI, int BM>
__global__ __launch_bounds__(kGemmThreads, 1) void gemm_bf16_mid_kernel(const u16* __restrict__ X, const u16* __restrict__ W,
                                                                       const float* __restrict__ bias,
                                                                       const u16* __restrict__ residual,
                                                                       u16* __restrict__ Y, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [stage][W tile | X tile]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int nblk = gridDim.x;
    const int orig = blockIdx.x;
    const int q = nblk / 8, rr = nblk % 8, xcd = orig % 8;
    const int bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    const int tiles_n = N / GBN;
    const int bn = bid % tiles_n, bm = bid / tiles_n;
    constexpr int NJ = BM / 32;              // 16-token MFMA tiles per wave (the wave's tokens: wm * BM/2 ..)
    constexpr int kXTile = BM * 128;         // bytes of an X tile
    constexpr int kStage = kTileBytes + kXTile;
    const int n0 = bn * GBN, m0 = bm * BM;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nk = K / GBK;   // >= 4 (launcher)

    // operand delivery: a tile = 16 W pieces + 16 X pieces of 8 rows x 128 B; this wave moves pieces wave + 4p, p = 0..3, of each
    auto make_desc = [](const void* base, unsigned bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        mid_i32x4 d;
        d[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        d[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
        d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        d[3] = 0x00020000;
        return d;
    };
    const mid_i32x4 wdesc = make_desc(W, (unsigned)N * (unsigned)K * 2u);
    const mid_i32x4 xdesc = make_desc(X, (unsigned)(gridDim.x / tiles_n * BM) * (unsigned)K * 2u);   // the row tiles launched are allocated
    const int dma_voff = ((lane >> 3) * K + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8)) * 2;
    const unsigned piece_step = (unsigned)K * 64u;   // 32 rows of K bf16
    const unsigned soW0 = __builtin_amdgcn_readfirstlane(((unsigned)(n0 + wave * 8) * (unsigned)K) * 2u);
    const unsigned soX0 = __builtin_amdgcn_readfirstlane(((unsigned)(m0 + wave * 8) * (unsigned)K) * 2u);
    const unsigned mbase = lds_base + wave * 1024;
    // piece `which` (0..3 W, 4..7 X; a 64-row X tile has two per wave: 4, 5) of tile t into stage t % 4
    auto dma = [&](int t, int which) {
        if (which >= 4 + NJ) return;
        const unsigned m0v = mbase + (t & (kMidStages - 1)) * kStage + (which < 4 ? 0 : kTileBytes) + (which & 3) * 4096;
        const unsigned so = (which < 4 ? soW0 : soX0) + (which & 3) * piece_step + (unsigned)t * 128u;
        asm volatile("s_mov_b32 m0, %0" ::"s"(m0v));
        if (which < 4) asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dma_voff), "s"(wdesc), "s"(so) : "memory");
        else asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dma_voff), "s"(xdesc), "s"(so) : "memory");
    };
    // the epilogue's operands leave FIRST: a workgroup has one tile, so loads issued after the K loop are a dependent L2 / HBM
    // round trip at the end of every launch (~1 us of 15).  They are older than every tile piece and loads complete in order,
    // so the counted vmcnt waits below mean what they meant.
    f32x4 ebias[4];
    uint2 eres[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
        ebias[i] = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            eres[i][j] = make_uint2(0, 0);
            if (EPI == 1) {
                const int m = m0 + wm * (BM / 2) + j * 16 + (lane & 15);
                eres[i][j] = *reinterpret_cast<const uint2*>(residual + (int64_t)(m < M ? m : M - 1) * N + n);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // prologue: tiles 0, 1, 2 and the W pieces of tile 3
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) dma(t, w8);
#pragma unroll
    for (int w8 = 0; w8 < 4; ++w8) dma(3, w8);
    if (NJ == 4) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");     // tile 0 landed (tiles 1, 2 and the W half of 3 may fly)
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment addresses inside a stage: row r of a tile at r * 128, 16-B chunk c at c ^ ((r >> 1) & 7)
    const int fr = lane & 15, sw = (fr >> 1) & 7;
    unsigned offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = (ks * 4 + (lane >> 4)) ^ sw;
        offA[ks] = (wn * 64 + fr) * 128 + ch * 16;
        offB[ks] = kTileBytes + (wm * (BM / 2) + fr) * 128 + ch * 16;
    }
    f32x4 acc[4][4];   // [i][j]: j < NJ used
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[4], b0[4], a1[4], b1[4];
    {
        const unsigned aa = lds_base + offA[0], bb = lds_base + offB[0];
        RASS_DS_READ_B128(a0[0], aa, 0); RASS_DS_READ_B128(a0[1], aa, 2048); RASS_DS_READ_B128(a0[2], aa, 4096); RASS_DS_READ_B128(a0[3], aa, 6144);
        RASS_DS_READ_B128(b0[0], bb, 0); RASS_DS_READ_B128(b0[1], bb, 2048);
        if (NJ == 4) { RASS_DS_READ_B128(b0[2], bb, 4096); RASS_DS_READ_B128(b0[3], bb, 6144); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int t = 0; t < nk; ++t) {
        const unsigned sb = lds_base + (t & (kMidStages - 1)) * kStage;
        const unsigned sn = lds_base + ((t + 1) & (kMidStages - 1)) * kStage;
        const bool x3 = t + 3 < nk, w4 = t + 4 < nk;
        // ---- sub-step 0: (t, 0) out of a0 / b0; reads (t, 1) into a1 / b1; the X pieces of tile t + 3
        {
            const unsigned aa = sb + offA[1], bb = sb + offB[1];
#define MID_SUB(AC, BC, AN, BN, DMA_T, DMA_BASE, DMA_ON)                                                                        \
    MID_MFMA(acc[0][0], AC[0], BC[0]); RASS_DS_READ_B128(AN[0], aa, 0);                                                          \
    MID_MFMA(acc[0][1], AC[0], BC[1]); RASS_DS_READ_B128(BN[0], bb, 0);                                                          \
    if (NJ == 4) { MID_MFMA(acc[0][2], AC[0], BC[2]); }                                                                          \
    if (DMA_ON) dma(DMA_T, DMA_BASE);                                                                                            \
    if (NJ == 4) { MID_MFMA(acc[0][3], AC[0], BC[3]); }                                                                          \
    RASS_DS_READ_B128(AN[1], aa, 2048);                                                                                          \
    MID_MFMA(acc[1][0], AC[1], BC[0]); RASS_DS_READ_B128(BN[1], bb, 2048);                                                       \
    MID_MFMA(acc[1][1], AC[1], BC[1]);                                                                                           \
    if (NJ == 4) { MID_MFMA(acc[1][2], AC[1], BC[2]); }                                                                          \
    if (DMA_ON) dma(DMA_T, DMA_BASE + 1);                                                                                        \
    if (NJ == 4) { MID_MFMA(acc[1][3], AC[1], BC[3]); }                                                                          \
    RASS_DS_READ_B128(AN[2], aa, 4096);                                                                                          \
    MID_MFMA(acc[2][0], AC[2], BC[0]); if (NJ == 4) { RASS_DS_READ_B128(BN[2], bb, 4096); }                                      \
    MID_MFMA(acc[2][1], AC[2], BC[1]);                                                                                           \
    if (NJ == 4) { MID_MFMA(acc[2][2], AC[2], BC[2]); }                                                                          \
    if (DMA_ON) dma(DMA_T, DMA_BASE + 2);                                                                                        \
    if (NJ == 4) { MID_MFMA(acc[2][3], AC[2], BC[3]); }                                                                          \
    RASS_DS_READ_B128(AN[3], aa, 6144);                                                                                          \
    MID_MFMA(acc[3][0], AC[3], BC[0]); if (NJ == 4) { RASS_DS_READ_B128(BN[3], bb, 6144); }                                      \
    MID_MFMA(acc[3][1], AC[3], BC[1]);                                                                                           \
    if (NJ == 4) { MID_MFMA(acc[3][2], AC[3], BC[2]); }                                                                          \
    if (DMA_ON) dma(DMA_T, DMA_BASE + 3);                                                                                        \
    if (NJ == 4) { MID_MFMA(acc[3][3], AC[3], BC[3]); }
            MID_SUB(a0, b0, a1, b1, t + 3, 4, x3)
        }
        // ---- the mid-step barrier: this wave's reads of tile t are done, its pieces of tile t + 1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (outstanding behind tile t + 1: tile t + 2 and both halves of tile t + 3 = 2 x (4 + NJ) instructions)
        if (x3 && NJ == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (x3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-step 1: (t, 1) out of a1 / b1; reads (t + 1, 0) into a0 / b0; the W pieces of tile t + 4
        {
            const unsigned aa = sn + offA[0], bb = sn + offB[0];
            MID_SUB(a1, b1, a0, b0, t + 4, 0, w4)
#undef MID_SUB
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // Epilogue (as gemm_bf16_kernel).  acc[i][j]: token m = m0 + wm*64 + j*16 + (lane&15); features
    // n = n0 + wn*64 + i*16 + (lane>>4)*4 + {0,1,2,3}.
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = m0 + wm * (BM / 2) + j * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
            f32x4 v = acc[i][j] + ebias[i];
            if (EPI == 1) {
                const uint2 r = eres[i][j];
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

// Where it pays (scripts/probe_encoder_shapes.py, whole forwards, same box): 129 .. 1 024 rows.  The K loop is not what bounds it —
// hand-scheduling it changed nothing: a workgroup keeps ~4 operand tiles (128 KiB) in flight against ~2 us of global -> LDS
// latency, i.e. ~60 GB/s per CU, and a 128 x 128 tile moves 512 KiB for K = 1 024 (15 us per GEMM on the 72-96 CUs such a
// batch occupies).  64-ROW tiles (while they still fit one per CU) put twice the CUs to work on 3/4 of the bytes each:
// 32 x 12 tokens 1.887 -> 1.51 ms per forward, 16 x 12: 1.610 -> 1.415; 64 x 12 (128-row tiles: 144 workgroups) 2.144 -> 1.99,
// 32 x 32: 2.228 -> 2.04.  Below 129 rows the split-K pair's workgroups win (8 x 12: 1.333 vs 1.376), from 1 536 rows on the
// two-buffer kernel's two workgroups per CU (48 x 32: 2.515 vs 2.59).
static bool mid_enabled(int M = 512) {
    const char* v = rass_env("RASS_GEMM_MID");   // 0: round 3's paths (two-buffer kernel / split-K pair); 2: every shape (the A/Bs)
    if (v != nullptr && atoi(v) == 0) return false;
    if (v != nullptr && atoi(v) == 2) return true;
    return M > 96 && M <= 8 * GBM;   // (from 129 rows until the end of round 4; 97 .. 128 rows: 120 tokens 1.44 -> 1.31 ms per forward)
}

// ------------------------------------------------------------------------------------------
// Split-K form of the 128 x 128 kernel for FEW rows (a query or a handful of chunks: embed_query / ollama_embed_text,
// reference app/main.py:225-237, 266-274).  With M <= 256 the plain kernel launches N/128 x M/128 = 8-32 workgroups, each
// walking all of K behind one barrier per 64-deep step: FFN-down (K = 4096) took 60 us, attn-out 13 us, a one-query
// forward 2.9 ms of which 60 % were these two (profiles/r02_encoder_b1_s16_kernel_stats.csv).  Here the K range is cut
// into S slices so that >= ~128 workgroups stream the weights; every slice writes its fp32 partial tile (rows < M
// only) to a scratch [S][M_pad][N], and splitk_epilogue_kernel sums the slices IN FIXED ORDER (deterministic: no
// atomics), adds bias / residual, applies GELU and rounds to bf16 — the same arithmetic as the fused epilogue up to
// the order of the fp32 partial sums.
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_splitk_kernel(const u16* __restrict__ X,
                                                                          const u16* __restrict__ W,
                                                                          float* __restrict__ partial, int M, int M_pad,
                                                                          int N, int K, int k_per_slice) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2 buf][W tile | X tile]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int tiles_n = N / GBN;
    const int bn = blockIdx.x % tiles_n, bm = blockIdx.x / tiles_n;
    const int slice = blockIdx.y;
    const int n0 = bn * GBN, m0 = bm * GBM, k_lo = slice * k_per_slice;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = k_per_slice / GBK;
    stage_tile(W, K, n0, k_lo, lds, wave, lane);
    stage_tile(X, K, m0, k_lo, lds + kTileBytes, wave, lane);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        unsigned char* buf = lds + cur * 2 * kTileBytes;
        if (t + 1 < nk) {
            unsigned char* nxt = lds + (cur ^ 1) * 2 * kTileBytes;
            stage_tile(W, K, n0, k_lo + (t + 1) * GBK, nxt, wave, lane);
            stage_tile(X, K, m0, k_lo + (t + 1) * GBK, nxt + kTileBytes, wave, lane);
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
    // acc[i][j]: token m = m0 + wm*64 + j*16 + (lane&15); features n0 + wn*64 + i*16 + (lane>>4)*4 + {0..3}
    float* P = partial + (int64_t)slice * M_pad * N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(P + (int64_t)m * N + n0 + wn * 64 + i * 16 + (lane >> 4) * 4) = acc[i][j];
    }
}

// y[m][n..n+3] = epi(sum over the S slices (ascending) + bias [+ residual]); one thread per 4 features
template <int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ partial, int S, int M, int M_pad,
                                                              int N, const float* __restrict__ bias,
                                                              const u16* __restrict__ residual, u16* __restrict__ Y) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over M * N/4
    const int n4 = N / 4;
    if (idx >= (int64_t)M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
    // the slices' loads go out together (S <= 16), the sum runs in ascending slice order
    f32x4 pv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        pv[s] = s < S ? *reinterpret_cast<const f32x4*>(partial + ((int64_t)s * M_pad + m) * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 v = pv[0];
#pragma unroll
    for (int s = 1; s < 16; ++s)
        if (s < S) v += pv[s];
    v += *reinterpret_cast<const f32x4*>(bias + n);
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

// ------------------------------------------------------------------------------------------
// A few rows against a WIDE weight matrix (one query: QKV, N = 3072, and FFN-up, N = 4096, at K = 1024): no split-K and
// no second kernel.  One wave per 16 output features walks all of K straight from global memory / L2 — its 16 weight
// rows are 32 KiB, read once, 16 B per lane and MFMA (A = W rows, B = X rows: D[feature][token]) with 8 loads in
// flight — and applies the epilogue itself; N / 16 >= 128 waves stream the matrix.  Every launch of a one-query forward
// costs ~5 us whatever it does (a hipGraph replay does not change that), so the two launches saved per layer are a
// fifth of the forward.  ROWS = number of 16-token blocks (tokens <= 64).
template <int EPI, int ROWS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_bf16_fewrows_kernel(const u16* __restrict__ X, const u16* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                const u16* __restrict__ residual, u16* __restrict__ Y,
                                                                int M, int N, int K, float* __restrict__ partial,
                                                                int rows_pad) {
    // EPI = -1: K is also cut over gridDim.y workgroups; each writes its fp32 partial tile [slice][rows_pad][N] and the
    // fused reduce + residual + LayerNorm kernel follows (FFN-down: K = 4096 needs more than 64 workgroups)
    // a workgroup = 16 output features; its WAVES (4, or 16 for K >= 4096) waves take an equal share of K each (8 weight
    // loads of 16 B per lane in flight per trip), then wave 0 adds the partial tiles in wave order
    __shared__ f32x4 part[WAVES][ROWS][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int g = lane >> 4, i = lane & 15;
    const int kq = K / (WAVES * (int)gridDim.y), k_lo = ((int)blockIdx.y * WAVES + wave) * kq;
    const u16* wrow = W + (int64_t)(n0 + i) * K + k_lo + 8 * g;   // A operand: W[n0 + i][k_lo + 32 ks + 8 g .. +7]
    const u16* xrow = X + (int64_t)i * K + k_lo + 8 * g;          // B operand: X[16 rb + i][..] (rows < M_pad exist)
    // wave 0's epilogue operands leave with the first weight loads, not after the barrier (a dependent L2 / HBM round trip
    // at the very end of a kernel whose whole duration is 4-5 us)
    const int n = n0 + 4 * g;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 rres[ROWS];
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) rres[rb] = make_uint2(0, 0);
    if (EPI >= 0 && wave == 0) {
        bv = *reinterpret_cast<const f32x4*>(bias + n);
        if (EPI == 1) {
#pragma unroll
            for (int rb = 0; rb < ROWS; ++rb) {
                const int m = 16 * rb + i;
                rres[rb] = *reinterpret_cast<const uint2*>(residual + (int64_t)(m < M ? m : 0) * N + n);
            }
        }
    }
    f32x4 acc[ROWS];
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int k0 = 0; k0 < kq; k0 += 32 * U) {   // one trip at K = 1024 (4 waves) and 4096 (16 waves)
        bf16x8 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const bf16x8*>(wrow + k0 + 32 * u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int rb = 0; rb < ROWS; ++rb) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(xrow + (int64_t)rb * 16 * K + k0 + 32 * u);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b, acc[rb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) part[wave][rb][lane] = acc[rb];
    __syncthreads();
    if (wave != 0) return;
    // token m = 16 rb + i, features n0 + 4 g + {0..3}
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) {
        const int m = 16 * rb + i;
        if (m >= M) continue;
        f32x4 v = part[0][rb][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += part[w][rb][lane];
        if constexpr (EPI < 0) {
            *reinterpret_cast<f32x4*>(partial + ((int64_t)blockIdx.y * rows_pad + m) * N + n) = v;
            continue;
        }
        v += bv;
        if (EPI == 1) {
            const uint2 r = rres[rb];
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

// The few-rows GEMM whose input is LayerNorm(Yin), recomputed by EVERY workgroup into its LDS X tile (<= 16 rows of
// K = 1024: 32 KiB of L2 reads, issued behind the weight loads already in flight) instead of a LayerNorm launch in
// front (WAVES = 16, the default: a wave normalises ONE row of 16 and owns a 64-deep slice of K — normalising four rows took a
// 4-wave workgroup ~1.5 us of vector issue, in every workgroup; RASS_GEMM_LNIN_WAVES=4 keeps that form): a launch costs ~4 us here whatever it does.  Workgroup 0 also stores the normalised rows (x_out: the next
// residual).  The row arithmetic is layernorm_kernel's (wave per row, lane = 8 columns + 512 s, fp32 two-pass,
// xor-shuffle sums), so x_out has the bits the separate launch would have written.
template <int EPI, int ROWS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_bf16_lnin_kernel(const u16* __restrict__ Yin, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             u16* __restrict__ x_out, const u16* __restrict__ W,
                                                             const float* __restrict__ bias, u16* __restrict__ Y, int M,
                                                             int N) {
    constexpr int K = 1024, kPitch = K + 8;   // + 16 B: the 16 rows of a B fragment fall on different banks
    extern __shared__ __attribute__((aligned(16))) unsigned char lnin_lds[];
    u16 (*xs)[kPitch] = reinterpret_cast<u16 (*)[kPitch]>(lnin_lds);                       // [16 ROWS][kPitch]
    f32x4 (*part)[ROWS][64] = reinterpret_cast<f32x4 (*)[ROWS][64]>(lnin_lds + (size_t)16 * ROWS * kPitch * 2);  // [WAVES]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int g = lane >> 4, i = lane & 15;
    constexpr int UW = 32 / WAVES;   // 32-deep MFMA steps of a wave's K slice (K / WAVES)
    const int k_lo = wave * (K / WAVES);
    const u16* wrow = W + (int64_t)(n0 + i) * K + k_lo + 8 * g;
    bf16x8 a[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u) a[u] = *reinterpret_cast<const bf16x8*>(wrow + 32 * u);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + 4 * g);   // (wave 0's epilogue: not a round trip at the end)
    // rows wave, wave + WAVES, ...: all their loads first
    constexpr int RPW = 16 * ROWS / WAVES;   // rows per wave
    constexpr int G = RPW < 4 ? RPW : 4;     // rows reduced side by side
    uint4 raw[RPW][2];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = wave + WAVES * j;
        const int rc = r < M ? r : 0;
#pragma unroll
        for (int st = 0; st < 2; ++st)
            raw[j][st] = *reinterpret_cast<const uint4*>(Yin + (int64_t)rc * K + lane * 8 + 512 * st);
    }
    f32x4 gm[2][2], bt[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int c = lane * 8 + 512 * st;
        gm[st][0] = *reinterpret_cast<const f32x4*>(gamma + c);
        gm[st][1] = *reinterpret_cast<const f32x4*>(gamma + c + 4);
        bt[st][0] = *reinterpret_cast<const f32x4*>(beta + c);
        bt[st][1] = *reinterpret_cast<const f32x4*>(beta + c + 4);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load above is issued before the first wait
    // four rows at a time, their wave reductions side by side: a row's arithmetic and its order are layernorm_kernel's, but
    // the 12 dependent cross-lane steps of a row (2 sums x 6 butterfly steps) overlap with the other rows' instead of running 48
    // deep, and they are DPP / permlane-swap moves, not ds_bpermute round trips (encoder_kernels.h; round 4: 8.2 -> ~5 us per launch)
#pragma unroll
    for (int j0 = 0; j0 < RPW; j0 += G) {
        float x[G][2][8], sum[G], mean[G], sq[G], rstd[G];
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint4 w = raw[j0 + jj][st];
                x[jj][st][0] = bf16_to_f32((u16)(w.x & 0xffff)); x[jj][st][1] = bf16_to_f32((u16)(w.x >> 16));
                x[jj][st][2] = bf16_to_f32((u16)(w.y & 0xffff)); x[jj][st][3] = bf16_to_f32((u16)(w.y >> 16));
                x[jj][st][4] = bf16_to_f32((u16)(w.z & 0xffff)); x[jj][st][5] = bf16_to_f32((u16)(w.z >> 16));
                x[jj][st][6] = bf16_to_f32((u16)(w.w & 0xffff)); x[jj][st][7] = bf16_to_f32((u16)(w.w >> 16));
            }
            sum[jj] = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[jj] += x[jj][st][e];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int jj = 0; jj < G; ++jj) sum[jj] += wave_xor_partner_dpp(sum[jj], lane, off);
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            mean[jj] = sum[jj] / (float)K;
            sq[jj] = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = x[jj][st][e] - mean[jj];
                    sq[jj] = fmaf(d, d, sq[jj]);
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int jj = 0; jj < G; ++jj) sq[jj] += wave_xor_partner_dpp(sq[jj], lane, off);
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            rstd[jj] = rsqrtf(sq[jj] / (float)K + eps);
            const int r = wave + WAVES * (j0 + jj);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int c = lane * 8 + 512 * st;
                float o[8];
                o[0] = (x[jj][st][0] - mean[jj]) * rstd[jj] * gm[st][0].x + bt[st][0].x;
                o[1] = (x[jj][st][1] - mean[jj]) * rstd[jj] * gm[st][0].y + bt[st][0].y;
                o[2] = (x[jj][st][2] - mean[jj]) * rstd[jj] * gm[st][0].z + bt[st][0].z;
                o[3] = (x[jj][st][3] - mean[jj]) * rstd[jj] * gm[st][0].w + bt[st][0].w;
                o[4] = (x[jj][st][4] - mean[jj]) * rstd[jj] * gm[st][1].x + bt[st][1].x;
                o[5] = (x[jj][st][5] - mean[jj]) * rstd[jj] * gm[st][1].y + bt[st][1].y;
                o[6] = (x[jj][st][6] - mean[jj]) * rstd[jj] * gm[st][1].z + bt[st][1].z;
                o[7] = (x[jj][st][7] - mean[jj]) * rstd[jj] * gm[st][1].w + bt[st][1].w;
                uint4 pk;
                pk.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
                pk.z = (unsigned)f32_to_bf16(o[4]) | ((unsigned)f32_to_bf16(o[5]) << 16);
                pk.w = (unsigned)f32_to_bf16(o[6]) | ((unsigned)f32_to_bf16(o[7]) << 16);
#ifdef RASS_ELIM_LN   // elimination build (timing only, wrong results): the raw row instead of the normalised one
                pk = raw[j0 + jj][st];
#endif
                if (r >= M) pk = make_uint4(0, 0, 0, 0);   // rows past the batch: finite zeros in the operand tile
                *reinterpret_cast<uint4*>(&xs[r][c]) = pk;
                if (blockIdx.x == 0 && r < M) *reinterpret_cast<uint4*>(x_out + (int64_t)r * K + c) = pk;
            }
        }
    }
    __syncthreads();
    f32x4 acc[ROWS];
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < UW; ++u) {
#pragma unroll
        for (int rb = 0; rb < ROWS; ++rb) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&xs[16 * rb + i][k_lo + 32 * u + 8 * g]);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b, acc[rb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) part[wave][rb][lane] = acc[rb];
    __syncthreads();
    if (wave != 0) return;
    const int n = n0 + 4 * g;
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) {
        const int m = 16 * rb + i;
        if (m >= M) continue;
        f32x4 v = part[0][rb][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += part[w][rb][lane];   // in wave order
        v += bv;
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

// tokens <= 64; K = whole 256-deep trips per wave: 4 waves per workgroup (K <= 3072), 16 for whole multiples of 4096
// RASS_GEMM_FEWROWS_MAX=<rows> (A/B; read per launch): the one-launch kernel up to that many rows where its partial tiles fit
// (4 waves: K <= 3072); default 128 (r03: 96 tokens 1.405 -> 1.337 ms per forward, 128 tokens 1.539 -> 1.495)
static int fewrows_max_rows() {
    const char* v = rass_env("RASS_GEMM_FEWROWS_MAX");
    const int m = v ? atoi(v) : 96;   // 128 until the end of round 4: from 97 rows the four-stage kernel (mid_enabled) is faster
    return m < 16 ? 16 : (m > 128 ? 128 : m);
}

static int fewrows_waves(int M, int N, int K) {
    if (M < 1 || N % 16 != 0 || N < 1024) return 0;
    if (K % 1024 == 0 && K <= 3072) return M <= fewrows_max_rows() ? 4 : 0;
    if (M > 64) return 0;                        // 16 waves x 8 row blocks of partial tiles would not fit the static LDS
    if (K % 4096 == 0 && K <= 8192) return 16;
    return 0;
}

static bool fewrows_enabled() {  // RASS_GEMM_FEWROWS=0: the split-K pair instead (A/B; read per launch)
    const char* v = rass_env("RASS_GEMM_FEWROWS");
    return !(v && v[0] == '0');
}

// the residual GEMMs (N = hidden) take the one-launch kernel only for the fewest rows: from 3 row blocks on the split-K
// pair is faster (measured at 48 and 64 tokens); RASS_GEMM_FEWROWS_RES=<rows> moves the limit (A/B)
static int fewrows_residual_max_rows() {
    const char* v = rass_env("RASS_GEMM_FEWROWS_RES");
    return v ? atoi(v) : 64;   // 32 until the end of round 4 (see the comment at launch_gemm_bf16_residual_layernorm)
}

template <int EPI, int WAVES>
static hipError_t launch_fewrows_w(const u16* x, const u16* w, const float* bias, const u16* r, u16* y, int M, int N, int K,
                                   hipStream_t stream, float* partial = nullptr, int rows_pad = 0, int slices = 1) {
    const dim3 grid(N / 16, slices), block(64 * WAVES);
    switch ((M + 15) / 16) {
        case 1: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 1, WAVES>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
        case 2: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 2, WAVES>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
        case 3: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 3, WAVES>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
        case 4: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 4, WAVES>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
        default:
            if constexpr (WAVES == 4) {   // 65 .. 128 rows: 4-wave workgroups only (fewrows_waves)
                switch ((M + 15) / 16) {
                    case 5: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 5, 4>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
                    case 6: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 6, 4>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
                    case 7: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 7, 4>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
                    default: hipLaunchKernelGGL((gemm_bf16_fewrows_kernel<EPI, 8, 4>), grid, block, 0, stream, x, w, bias, r, y, M, N, K, partial, rows_pad); break;
                }
            } else {
                return hipErrorInvalidValue;
            }
            break;
    }
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_fewrows(const u16* x, const u16* w, const float* bias, const u16* r, u16* y, int M, int N, int K,
                                 int waves, hipStream_t stream) {
    return waves == 16 ? launch_fewrows_w<EPI, 16>(x, w, bias, r, y, M, N, K, stream)
                       : launch_fewrows_w<EPI, 4>(x, w, bias, r, y, M, N, K, stream);
}

// Number of K slices for a GEMM with few output tiles (0 = do not split), whole 64-deep steps per slice, a scratch of
// S * M_pad * N floats that fits.  Measured on MI355X (scripts/probe_gemm_mid.py, profiles/r03_gemm_mid_sweep.txt; every
// combination of the four encoder GEMMs x 128 .. 3 072 rows x S): what bounds these kernels is the rate at which ONE CU
// can fill its LDS (one 128^2 workgroup takes ~0.9 us per 64-deep step however deep its prefetch ring is — a four-slot
// ring with counted waits measured the SAME times as this two-buffer loop and was removed), so a short K (1 024) wants
// >= 128 workgroups of >= 4 steps and a long K (4 096) up to 512 workgroups of >= 16 steps; beyond that the fp32 partials
// cost more than the split wins.  The rule was then settled on whole forwards (cold weights: scripts/sweep_splitk_rule.sh,
// same file), where more workgroups pull harder on HBM than the warm micro-benchmark shows.  Round 2 split only below 96
// tiles and aimed at 128 workgroups: FFN-down ran 64 serial steps at 96+ tiles (1 024 rows 29 -> 22 us, 1 536 rows
// 45 -> 28, 2 048 rows 46 -> 34).
static int splitk_slices(int M_pad, int N, int K, size_t ws_bytes) {
    const int tiles = (N / GBN) * (M_pad / GBM), steps = K / GBK;
    if (const char* v = rass_env("RASS_GEMM_SPLITK_S")) {   // sweeps (scripts/probe_gemm_mid.py)
        int S = atoi(v);
        if (S < 2 || S > 16 || steps % S != 0 || (size_t)S * M_pad * N * sizeof(float) > ws_bytes) return 0;
        return S;
    }
    if (steps < 2) return 0;
    int S = 1;
    if (K < 2048) {
        // short K (16 steps): the smallest split that gives >= 128 workgroups, slices of >= 4 steps
        while (S < 4 && tiles * S < 128 && steps % (2 * S) == 0) S *= 2;
    } else {
        // long K (64 steps): the largest split that stays within 512 workgroups (two resident per CU); slices of >= 16 steps
        // from 48 tiles on, >= 8 below, >= 4 for a single row of tiles
        const int cap = tiles <= 8 ? 16 : tiles < 48 ? 8 : 4;
        while (S < cap && tiles * S * 2 <= 512 && steps % (2 * S) == 0) S *= 2;
    }
    while (S > 1 && (size_t)S * M_pad * N * sizeof(float) > ws_bytes) S /= 2;
    return S > 1 ? S : 0;
}

// ------------------------------------------------------------------------------------------
// Large shapes (>= 192 tiles of 256 x 256): the persistent kernel below ("p5").  Its predecessors — the one-tile-per-block
// 3-slot ring kernel, its persistent form (pring), the two-slot 64-deep form (p64) and the 4-wave 128x128-per-wave kernel
// (w4l), each measured slower than p5 (profiles/r01_gemm_*.txt, profiles/r02_gemm_w4_experiments.txt) — were retired from
// the product library in round 3 and live on as an archive that still builds: scripts/microbench/gemm_retired_kernels.hip.
#ifdef RASS_GEMM_CLOCKS  // scripts/microbench builds only
__device__ unsigned long long g_gemm_clocks[4 * 16384];
__device__ unsigned long long g_gemm_core_cycles[64];
#ifdef RASS_GEMM_PHASE_TIMERS
__device__ unsigned long long g_gemm_phase_cycles[64 * 2 * 4];
#endif
#endif
#ifdef RASS_GEMM_STAMPS   // diagnostic builds only (scripts/probe_gemm_stamps.py): wall-clock (100 MHz) stamps of workgroups 0..7
__device__ unsigned long long g_p5_stamps[8 * 64 * 4];   // [block][tile][K loop start, K loop end, epilogue stores issued, tile end]
extern "C" int rassdiag_gemm_stamps(unsigned long long* out, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned long long h[8 * 64 * 4];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p5_stamps), sizeof(h)) != hipSuccess) return -2;
    for (int i = 0; i < n && i < 8 * 64 * 4; ++i) out[i] = h[i];
    return 0;
}
#define P5_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (threadIdx.x == 0 && orig < 8 && tile_no < 64) g_p5_stamps[(orig * 64 + tile_no) * 4 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define P5_STAMP(slot) do {} while (0)
#endif
constexpr int RBM = 256, RBN = 256;       // tile of the persistent kernel
constexpr int kRingThreads = 512;         // 2 (N) x 4 (M) waves, each 128 x 64 = 8 x 4 MFMA tiles
constexpr int kPStageTokens = 32;         // tokens per epilogue staging chunk (8 704 B per wave at a pitch of 68 floats)

// K-loop fragment read as opaque asm: hipcc's waitcnt pass orders every LDS access it can see after
// the LDS-DMA (global_load_lds) ops still in flight — in two of the three epilogue variants of the
// persistent kernel it put a vmcnt(0) in front of the fragment reads of EVERY K step (K loop 45 us
// instead of 28).  The DMA / read ordering is this kernel's own protocol (counted vmcnt + barrier).

// ------------------------------------------------------------------------------------------
// "p5": gemm_bf16_p64_kernel's whole-line operand stream with a ring of FIVE 32-KiB HALF-slots instead of two 64-KiB
// slots.  Half-load q = 2T + h holds rows 128h .. 128h+127 of both operand tiles of the 64-deep step T (W half at
// +0, X half at +16 KiB, rows of 128 B, swizzled as in p64) and lives in half-slot (q0 + q) % 5.  A wave's A
// fragments come from half wn of the step, its B fragments from half wm>>1.  While step T is multiplied (two
// half-slots), (T+1, 0) and (T+1, 1) and (T+2, 0) are in flight or landed: 1.5 steps ahead where two whole slots
// allowed one, and the DMA issues spread evenly — every load phase issues one half of a half-load (4 pieces per
// wave, as in the 32-deep kernel): step T issues (T+1, 1) in its first load phase and (T+2, 0) in its second, into
// the half-slots step T-1 was read from.  The stream runs on across tiles; only the next tile's third half-load
// waits for the epilogue to end (its 8 staging areas need three free half-slots).  Exactly 160 KiB of LDS.
constexpr int kP5HalfBytes = 32768;
constexpr int kP5LdsBytes = 5 * kP5HalfBytes;

// POL = cache policy of the three streams, one decimal digit each (x w y): operand loads 0 = default, 2 = nt (streaming), 1 = sc0,
// 3 = sc0 nt (the aux bits of global_load_lds); output stores 0 = default, 1 = nontemporal.  Measured in round 3
// (profiles/r03_gemm_power_limit.txt §6, RASS_P5_POLICY): nt on either operand stream costs 1-8 %, nontemporal OUTPUT stores
// win 3 % on the wide-output shapes (QKV 790 -> 763 us, FFN-up 1 115 -> 1 080; the 0.8-1.1 GB of output no longer push the
// operands out of L2) and nothing on the N = 1024 ones: POL = 1 is the default, RASS_P5_POLICY=0 the A/B.
// ---- LayerNorm folded into the GEMMs around it (round 4; EPI 3 / 4 / 5) ------------------------------------------------
// The post-LN encoder layer is  h1 = LN1(x + attn(x) Wo),  h2 = LN2(h1 + gelu(h1 Wup) Wdown).  The stand-alone LayerNorm kernel
// is HBM-bound (read + write of [T, 1024] bf16 at 5.9 TB/s = 90.7 us, twice per layer = 5.2 % of the forward) and a "thin"
// normalise pass would move the same bytes; what removes the pass is algebra:
//     LN(r) W^T = rstd * (r W'^T  -  mu * colsum(W'))  +  (beta W^T + b),      W' = W diag(gamma)  (bf16, prepared at load)
// so the CONSUMER GEMM (QKV / FFN-up) runs on the raw, un-normalised sums r with pre-scaled weights and applies the row's
// (mu, rstd) and a rank-1 correction in its epilogue (EPI 4: + bias', EPI 5: + bias' + GELU), and the RESIDUAL GEMM
// (attn-out / FFN-down, EPI 3) rebuilds the normalised residual LN_prev(r_prev) element by element from (r_prev, mu, rstd,
// gamma, beta) on the fly, writes the raw sum r (bf16) and, per row and 128-column chunk, the partial sums (S r, S r^2) of the
// ROUNDED values — no atomics: [row][chunk][2] floats, summed in fixed order by ln_stats_finalize_kernel into (mu, rstd).
struct LnFold {
    const float* mr = nullptr;        // EPI 3 / 4 / 5: [rows][2] (mean, rstd) of the rows of `residual` (EPI 3) or of X (EPI 4 / 5)
    const float* gamma = nullptr;     // EPI 3: gamma / beta of the LayerNorm that produced the residual, [N]
    const float* beta = nullptr;
    float* stats = nullptr;           // EPI 3: out, [rows][N / 128][2] partial (sum, sum of squares) of the stored bf16 values
    const float* colsum = nullptr;    // EPI 4 / 5: [N] column sums of W' (fp32 sums of its bf16 values)
};

// xor-reductions inside groups of 8 consecutive lanes on DPP (quad_perm [1,0,3,2], [2,3,0,1], then row_half_mirror)
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    return v;
}

template <int EPI, int POL = 1>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_bf16_p5_kernel(const u16* __restrict__ X,
                                                                      const u16* __restrict__ W,
                                                                      const float* __restrict__ bias,
                                                                      const u16* __restrict__ residual,
                                                                      u16* __restrict__ Y, int M, int N, int K,
                                                                      int tiles_total, LnFold fold) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 2, wm = wave & 3;
    const int G = gridDim.x, orig = blockIdx.x;
    const int pos = (G % 8 == 0) ? (orig % 8) * (G / 8) + orig / 8 : orig;
    int tile = pos;
    if (tile >= tiles_total) return;
    const int tiles_n = N / RBN;
    const int nk = K / 64;   // >= 2 (launcher)

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // fragment offsets inside a half-slot (rows of 128 B; chunk c of row r at c ^ ((r>>1)&7); sub-step s = chunks 4s..4s+3)
    unsigned offA[2], offB[2];
    {
        const int sw = ((lane & 15) >> 1) & 7;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ch = (sub * 4 + (lane >> 4)) ^ sw;
            offA[sub] = (lane & 15) * 128 + ch * 16;
            offB[sub] = 16384 + ((wm & 1) * 64 + (lane & 15)) * 128 + ch * 16;
        }
    }
    const int hA = wn, hB = wm >> 1;  // which half of a step this wave's A / B fragments live in
    // DMA: a half-load is 16 W pieces + 16 X pieces of 8 rows x 128 B; this wave moves pieces wave and wave + 8 of each
    const u16* srcW[2][2];
    const u16* srcX[2][2];
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    auto point_half = [&](int t, auto h_c) {   // sources of half h of tile t's step 0
        constexpr int h = decltype(h_c)::value;
        const int tn0 = (t % tiles_n) * RBN, tm0 = (t / tiles_n) * RBM;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = (wave + 8 * p) * 8 + (lane >> 3);   // row inside the half
            const int c_src = (lane & 7) ^ ((r >> 1) & 7);
            srcW[h][p] = W + (int64_t)(tn0 + 128 * h + r) * K + c_src * 8;
            srcX[h][p] = X + (int64_t)(tm0 + 128 * h + r) * K + c_src * 8;
        }
    };
    auto stage_half = [&](int hs, auto h_c) {   // hs: half-slot index 0..4
        constexpr int h = decltype(h_c)::value;
#ifdef RASS_GEMM_EXP_NO_DMA      // timing experiment: no operand delivery at all (stale LDS)
        (void)hs;
        return;
#endif
        unsigned char* base = lds + hs * kP5HalfBytes;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcX[h][p],
                                             (__attribute__((address_space(3))) void*)(base + 16384 + (wave + 8 * p) * 1024),
                                             16, 0, (POL / 100) % 10);
            srcX[h][p] += 64;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcW[h][p],
                                             (__attribute__((address_space(3))) void*)(base + (wave + 8 * p) * 1024), 16, 0,
                                             (POL / 10) % 10);
            srcW[h][p] += 64;
        }
    };
    auto mod5 = [](int v) { return v >= 5 ? v - 5 : v; };
    const bool grpB = wave >= 4;

    // half-slot of the current tile's half-load 0; every tile advances it by 2 * nk (mod 5)
    int q0 = 0;
    const int tile_adv = (2 * nk) % 5;
    point_half(tile, H0{});
    point_half(tile, H1{});
    stage_half(0, H0{});
    stage_half(1, H1{});
    stage_half(2, H0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int tile_no = 0;
    (void)tile_no;
    for (;;) {
        const int n0 = (tile % tiles_n) * RBN, m0 = (tile / tiles_n) * RBM;
        const int next = tile + G;
        const bool has_next = next < tiles_total;
        P5_STAMP(0);
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (grpB) __builtin_amdgcn_s_barrier();   // group B runs one phase behind group A
        int hs0 = q0;                              // half-slot of (t, 0)
        for (int t = 0; t < nk; ++t) {
            const int hs1 = mod5(hs0 + 1), hs2 = mod5(hs0 + 2), hs3 = mod5(hs0 + 3), hs4 = mod5(hs0 + 4);
            // half-loads this step issues: q = 2t+3 = (t+1, 1) into hs3 and q = 2t+4 = (t+2, 0) into hs4; beyond the
            // tile they are the next tile's (whose third half-load waits for the epilogue)
            const bool iss1 = (t + 1 < nk) || has_next;
            const bool iss2 = (t + 2 < nk) || (has_next && t + 2 == nk);
            const unsigned aslot = lds_base + (hA ? hs1 : hs0) * kP5HalfBytes;
            const unsigned bslot = lds_base + (hB ? hs1 : hs0) * kP5HalfBytes;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                // ---- load phase
                if (sub == 0 && iss1) {
                    if (t + 1 == nk) point_half(next, H1{});   // the half-load is the next tile's (0, 1)
                    stage_half(hs3, H1{});
                }
                if (sub == 1 && iss2) {
                    if (t + 2 == nk) point_half(next, H0{});   // the next tile's (0, 0)
                    stage_half(hs4, H0{});
                }
                bf16x8 a[8], b[4];
#ifdef RASS_GEMM_EXP_NO_MFMA    // timing experiment: the operand stream alone (DMA + waits + barriers)
                for (int i = 0; i < 8; ++i) a[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                for (int j = 0; j < 4; ++j) b[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                (void)aslot; (void)bslot;
#else
                {
                    const unsigned ab = aslot + offA[sub];
                    const unsigned bb = bslot + offB[sub];
                    RASS_DS_READ_B128(b[0], bb, 0);
                    RASS_DS_READ_B128(b[1], bb, 2048);
                    RASS_DS_READ_B128(b[2], bb, 4096);
                    RASS_DS_READ_B128(b[3], bb, 6144);
                    RASS_DS_READ_B128(a[0], ab, 0);
                    RASS_DS_READ_B128(a[1], ab, 2048);
                    RASS_DS_READ_B128(a[2], ab, 4096);
                    RASS_DS_READ_B128(a[3], ab, 6144);
                    RASS_DS_READ_B128(a[4], ab, 8192);
                    RASS_DS_READ_B128(a[5], ab, 10240);
                    RASS_DS_READ_B128(a[6], ab, 12288);
                    RASS_DS_READ_B128(a[7], ab, 14336);
                }
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // before the barrier that ends the step: this wave's pieces of (t+1, 1) have landed; (t+2, 0)'s four may fly
                if (sub == 1 && grpB) {
                    if (iss2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- compute phase
                __builtin_amdgcn_s_setprio(1);
#ifndef RASS_GEMM_EXP_NO_MFMA
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#endif
                __builtin_amdgcn_s_setprio(0);
                if (sub == 1 && !grpB) {
                    if (iss2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            hs0 = hs2;
        }
        if (!grpB) __builtin_amdgcn_s_barrier();  // groups re-aligned: every ring read of this tile is done
        P5_STAMP(1);
        // the next tile's half-loads 0 and 1 are landing in hs0, hs0+1 (= its q0); staging: three free half-slots
        q0 = mod5(q0 + tile_adv);
        float* const stg = reinterpret_cast<float*>(lds + mod5(q0 + 2 + wave / 3) * kP5HalfBytes + (wave % 3) * 8704);

        // ---- epilogue (see gemm_bf16_ring_kernel): LDS transpose per wave, coalesced 16-B stores
        {
            constexpr int kPitchF = 68;
            const int tl = lane >> 3, nq = lane & 7;
            // Bias through opaque asm loads, retired by the explicit vmcnt(0) below: a load hipcc can
            // see stays "possibly pending" on its destination registers across the tile loop, and
            // when the K loop's fragment reads get the same registers the waitcnt pass protects them
            // with a vmcnt(0) in EVERY K step (seen in two of the three epilogue variants).
            f32x4 bv[2][2];
            // Output / residual rows go through BUFFER ops on per-tile descriptors (base = the tile's first row, size = its
            // rows below M): rows past M are dropped / read as zero by the bounds check instead of by a branch.  With
            // `if (m < M)` around every global load and store hipcc's waitcnt pass lost count at the block boundaries and put
            // an s_waitcnt vmcnt(0) in front of EVERY store of the residual epilogues: 16 store round trips per tile,
            // 9-10 us against 3.4 for the bias-only epilogue (scripts/probe_gemm_stamps.py, ISA).
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const int rows_here = M - m0 < RBM ? (M - m0 > 0 ? M - m0 : 0) : RBM;
            const unsigned tile_bytes = __builtin_amdgcn_readfirstlane((unsigned)rows_here * (unsigned)N * 2u);
            auto tile_desc = [&](const u16* base) {
                const uint64_t bu = reinterpret_cast<uint64_t>(base + (int64_t)m0 * N);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bu), hi = __builtin_amdgcn_readfirstlane((uint32_t)(bu >> 32));
                return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u16*>(((uint64_t)hi << 32) | lo), 0, (int)tile_bytes, 0x00020000);
            };
            const __amdgpu_buffer_rsrc_t ydesc = tile_desc(Y);
            const __amdgpu_buffer_rsrc_t rdesc = tile_desc((EPI == 1 || EPI == 3) ? residual : Y);
            // LN fold: per-column vectors of the lane's 2 x 8 columns (EPI 3: gamma / beta of the residual's LayerNorm;
            // EPI 4 / 5: colsum(W')), loaded like the bias
            // (the LN fold's per-column vectors — EPI 3: gamma / beta, EPI 4 / 5: colsum(W') — are loaded per 64-column chunk
            // inside the loop: held across the whole epilogue like the bias they spilled)
#pragma unroll
            for (int ic = 0; ic < 2; ++ic) {
                const float* bp = bias + n0 + wn * 128 + ic * 64 + nq * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[ic][0]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv[ic][1]) : "v"(bp));
            }
            // LN fold: this wave's per-token (mean, rstd) pairs (its 64 tokens) and per-column vectors (its 128 columns: EPI 3
            // gamma / beta, EPI 4 / 5 colsum(W')) are fetched ONCE per tile — one 8-byte piece per lane and array, opaque
            // loads like the bias, retired by the same vmcnt(0) — and parked in the 2 KiB of LDS behind the wave's staging
            // area: as loads inside the (jc, ic) loop they put a memory round trip into each of the tile's four iterations
            // (QKV + 54 us, FFN-up + 89 us per call).
            float* const aux = reinterpret_cast<float*>(lds + mod5(q0 + 2 + wave / 3) * kP5HalfBytes + 26112 + (wave % 3) * 2048);
            float2 aux_mr = float2{0.f, 1.f}, aux_c0 = float2{0.f, 0.f}, aux_c1 = float2{0.f, 0.f};
            if constexpr (EPI >= 3) {
                const int mt = m0 + wm * 64 + lane;
                const float* mp = fold.mr + 2 * (int64_t)(mt < M ? mt : 0);
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(aux_mr) : "v"(mp));
                const float* c0 = (EPI == 3 ? fold.gamma : fold.colsum) + n0 + wn * 128 + 2 * lane;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(aux_c0) : "v"(c0));
                if constexpr (EPI == 3) {
                    const float* c1 = fold.beta + n0 + wn * 128 + 2 * lane;
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(aux_c1) : "v"(c1));
                }
            }
            uint4 resbuf[2][4];   // residual rows of the current and of the next (jc, ic) iteration
#pragma unroll
            for (int jc = 0; jc < 64 / kPStageTokens; ++jc) {
                float st_s[kPStageTokens / 8], st_q[kPStageTokens / 8];   // EPI 3: this wave's 128-column partial sums per token
#pragma unroll
                for (int pass = 0; pass < kPStageTokens / 8; ++pass) st_s[pass] = st_q[pass] = 0.f;
#pragma unroll
                for (int ic = 0; ic < 2; ++ic) {
                    const int nbase = n0 + wn * 128 + ic * 64 + nq * 8;
                    float mu[kPStageTokens / 8], rs[kPStageTokens / 8];   // EPI >= 3: (mean, rstd) of the token's row
                    f32x4 cv[2];          // EPI 4 / 5: colsum(W') of this chunk's 8 columns
                    f32x4 gv[2], ev[2];   // EPI 3: gamma / beta of the residual's LayerNorm for this chunk's 8 columns
                    // Residual rows (EPI 1 / 3): iteration it's 4 x 16 B per lane are loaded one iteration AHEAD, right after
                    // iteration it - 1's transposes (their accumulators are dead by then) and BEFORE its stores: a wait for
                    // them then leaves those stores in flight (vmcnt counts both, in order).  Loaded at the top of their
                    // own iteration they queued behind the previous iteration's stores and every iteration paid a store
                    // round trip: 9-10 us per tile against 3.4 for the bias-only epilogue (scripts/probe_gemm_stamps.py).
                    const int it = jc * 2 + ic;
                    auto load_res = [&](int jc2, int ic2, uint4 (&dst)[4]) {
                        const int nb2 = n0 + wn * 128 + ic2 * 64 + nq * 8;   // (column inside the row; the descriptor starts at row m0)
#pragma unroll
                        for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                            const int row = wm * 64 + jc2 * kPStageTokens + pass * 8 + tl;
                            // read once, like the output: nontemporal keeps it out of the operands' way (POL; always for EPI 3)
                            const u32x4_t rv = __builtin_amdgcn_raw_buffer_load_b128(rdesc, (row * N + nb2) * 2, 0,
                                                                                     (POL % 10 == 1 || EPI == 3) ? 2 : 0);
                            dst[pass] = uint4{rv[0], rv[1], rv[2], rv[3]};
                        }
                    };
                    uint4 (&res)[4] = resbuf[it & 1];
                    if ((EPI == 1 || EPI == 3) && it == 0) load_res(0, 0, resbuf[0]);
#ifndef RASS_GEMM_EXP_NO_TRANSPOSE
#pragma unroll
                    for (int jj = 0; jj < kPStageTokens / 16; ++jj)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii)
                            *reinterpret_cast<f32x4*>(stg + (jj * 16 + (lane & 15)) * kPitchF + ii * 16 + (lane >> 4) * 4) =
                                acc[4 * ic + ii][(kPStageTokens / 16) * jc + jj];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    if ((EPI == 1 || EPI == 3) && it < 3) load_res((it + 1) >> 1, (it + 1) & 1, resbuf[(it + 1) & 1]);
                    if (jc == 0 && ic == 0) {
                        // Explicit: this wave's prefetch DMAs (and the bias / first residual reads issued
                        // after them) are complete before anything below consumes them and before the
                        // publishing barrier after the epilogue.  No store is outstanding yet.
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if constexpr (EPI >= 3) {   // park the tile's LN-fold scalars in LDS (wave-private: no barrier)
                            *reinterpret_cast<float2*>(aux + 2 * lane) = aux_c0;
                            if constexpr (EPI == 3) *reinterpret_cast<float2*>(aux + 128 + 2 * lane) = aux_c1;
                            *reinterpret_cast<float2*>(aux + 256 + 2 * lane) = aux_mr;
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                    }
                    if constexpr (EPI >= 3) {
#pragma unroll
                        for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                            const float2 v = *reinterpret_cast<const float2*>(aux + 256 + 2 * (jc * kPStageTokens + pass * 8 + tl));
                            mu[pass] = v.x;
                            rs[pass] = v.y;
                        }
                        if constexpr (EPI == 3) {
                            gv[0] = *reinterpret_cast<const f32x4*>(aux + ic * 64 + nq * 8);
                            gv[1] = *reinterpret_cast<const f32x4*>(aux + ic * 64 + nq * 8 + 4);
                            ev[0] = *reinterpret_cast<const f32x4*>(aux + 128 + ic * 64 + nq * 8);
                            ev[1] = *reinterpret_cast<const f32x4*>(aux + 128 + ic * 64 + nq * 8 + 4);
                        } else {
                            cv[0] = *reinterpret_cast<const f32x4*>(aux + ic * 64 + nq * 8);
                            cv[1] = *reinterpret_cast<const f32x4*>(aux + ic * 64 + nq * 8 + 4);
                        }
                    }
#pragma unroll
                    for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                        const int tok = pass * 8 + tl;
#ifdef RASS_GEMM_EXP_NO_TRANSPOSE   // timing experiment: the epilogue without its LDS round trip (values from the wrong lanes)
                        f32x4 v0 = acc[4 * ic + (pass & 3)][2 * jc];
                        f32x4 v1 = acc[4 * ic + (pass & 3)][2 * jc + 1];
#else
                        f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8);
                        f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8 + 4);
#endif
                        if constexpr (EPI >= 4) {   // LN folded into this GEMM: rstd * (x W'^T - mu * colsum(W')) + bias'
                            // two fused ops per element: (b * colsum + bias') first, then acc * rstd + that
                            const float a = rs[pass], b = -mu[pass] * rs[pass];
                            v0.x = fmaf(v0.x, a, fmaf(b, cv[0].x, bv[ic][0].x)); v0.y = fmaf(v0.y, a, fmaf(b, cv[0].y, bv[ic][0].y));
                            v0.z = fmaf(v0.z, a, fmaf(b, cv[0].z, bv[ic][0].z)); v0.w = fmaf(v0.w, a, fmaf(b, cv[0].w, bv[ic][0].w));
                            v1.x = fmaf(v1.x, a, fmaf(b, cv[1].x, bv[ic][1].x)); v1.y = fmaf(v1.y, a, fmaf(b, cv[1].y, bv[ic][1].y));
                            v1.z = fmaf(v1.z, a, fmaf(b, cv[1].z, bv[ic][1].z)); v1.w = fmaf(v1.w, a, fmaf(b, cv[1].w, bv[ic][1].w));
                        } else {
                            v0 += bv[ic][0];
                            v1 += bv[ic][1];
                        }
                        if constexpr (EPI == 3) {   // residual = LayerNorm_prev(raw row), rebuilt from (raw, mu, rstd, gamma, beta)
                            const uint4 r = res[pass];
                            const float a = rs[pass], b = -mu[pass] * rs[pass];
                            v0.x += fmaf(fmaf(bf16_to_f32((u16)(r.x & 0xffff)), a, b), gv[0].x, ev[0].x);
                            v0.y += fmaf(fmaf(bf16_to_f32((u16)(r.x >> 16)), a, b), gv[0].y, ev[0].y);
                            v0.z += fmaf(fmaf(bf16_to_f32((u16)(r.y & 0xffff)), a, b), gv[0].z, ev[0].z);
                            v0.w += fmaf(fmaf(bf16_to_f32((u16)(r.y >> 16)), a, b), gv[0].w, ev[0].w);
                            v1.x += fmaf(fmaf(bf16_to_f32((u16)(r.z & 0xffff)), a, b), gv[1].x, ev[1].x);
                            v1.y += fmaf(fmaf(bf16_to_f32((u16)(r.z >> 16)), a, b), gv[1].y, ev[1].y);
                            v1.z += fmaf(fmaf(bf16_to_f32((u16)(r.w & 0xffff)), a, b), gv[1].z, ev[1].z);
                            v1.w += fmaf(fmaf(bf16_to_f32((u16)(r.w >> 16)), a, b), gv[1].w, ev[1].w);
                        }
                        if (EPI == 1) {
                            const uint4 r = res[pass];
                            v0.x += bf16_to_f32((u16)(r.x & 0xffff));
                            v0.y += bf16_to_f32((u16)(r.x >> 16));
                            v0.z += bf16_to_f32((u16)(r.y & 0xffff));
                            v0.w += bf16_to_f32((u16)(r.y >> 16));
                            v1.x += bf16_to_f32((u16)(r.z & 0xffff));
                            v1.y += bf16_to_f32((u16)(r.z >> 16));
                            v1.z += bf16_to_f32((u16)(r.w & 0xffff));
                            v1.w += bf16_to_f32((u16)(r.w >> 16));
                        }
                        if (EPI == 2 || EPI == 5) {
                            v0.x = gelu_erf(v0.x); v0.y = gelu_erf(v0.y); v0.z = gelu_erf(v0.z); v0.w = gelu_erf(v0.w);
                            v1.x = gelu_erf(v1.x); v1.y = gelu_erf(v1.y); v1.z = gelu_erf(v1.z); v1.w = gelu_erf(v1.w);
                        }
                        if constexpr (EPI == 3) {   // the row statistics of what is STORED (the bf16 values the consumers read)
                            const float q0 = bf16_to_f32(f32_to_bf16(v0.x)), q1 = bf16_to_f32(f32_to_bf16(v0.y)),
                                        q2 = bf16_to_f32(f32_to_bf16(v0.z)), q3 = bf16_to_f32(f32_to_bf16(v0.w)),
                                        q4 = bf16_to_f32(f32_to_bf16(v1.x)), q5 = bf16_to_f32(f32_to_bf16(v1.y)),
                                        q6 = bf16_to_f32(f32_to_bf16(v1.z)), q7 = bf16_to_f32(f32_to_bf16(v1.w));
                            const float s = ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7));
                            float q = q0 * q0;
                            q = fmaf(q1, q1, q); q = fmaf(q2, q2, q); q = fmaf(q3, q3, q);
                            q = fmaf(q4, q4, q); q = fmaf(q5, q5, q); q = fmaf(q6, q6, q); q = fmaf(q7, q7, q);
                            st_s[pass] += sum8_dpp(s);
                            st_q[pass] += sum8_dpp(q);
                        }
#ifdef RASS_GEMM_EXP_NO_STORE   // timing experiment: everything but the output stores (one store per 2^20 keeps the math alive)
                        if (v0.x == 12345.678f)
#endif
                        {
                            uint4 o;
                            o.x = (unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16);
                            o.y = (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16);
                            o.z = (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16);
                            o.w = (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16);
                            const int voff = ((wm * 64 + jc * kPStageTokens + tok) * N + nbase) * 2;   // bytes from the tile's first row
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{o.x, o.y, o.z, o.w}, ydesc, voff, 0, POL % 10 == 1 ? 2 : 0);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if constexpr (EPI == 3) {   // one (sum, sum of squares) pair per token and 128-column chunk of this wave
                    if (nq == 0) {
#pragma unroll
                        for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                            const int m = m0 + wm * 64 + jc * kPStageTokens + pass * 8 + tl;
                            if (m < M)
                                *reinterpret_cast<float2*>(fold.stats + ((int64_t)m * (N / 128) + (n0 / 128 + wn)) * 2) =
                                    float2{st_s[pass], st_q[pass]};
                        }
                    }
                }
            }
        }
        P5_STAMP(2);
#ifdef RASS_GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic: when have this wave's stores left?
        P5_STAMP(3);
#endif
        if (!has_next) break;
        // every wave is done with its staging area: the next tile's third half-load may overwrite it
        __builtin_amdgcn_s_barrier();
        stage_half(mod5(q0 + 2), H0{});
        tile = next;
        ++tile_no;
    }
}

template <int EPI, int POL>
static hipError_t launch_p5_pol(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M, int N,
                                int K, int tiles_total, int grid, hipStream_t stream, const LnFold& fold = LnFold{}) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_p5_kernel<EPI, POL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kP5LdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_p5_kernel<EPI, POL>), dim3(grid), dim3(kRingThreads), kP5LdsBytes, stream, X, W, bias,
                       residual, Y, M, N, K, tiles_total, fold);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// "p4" (round 4; the persistent GEMM of big shapes since then, RASS_GEMM_VARIANT=p5 = the A/B): the same 256 x 256 tile and
// five-half-slot LDS-DMA ring as p5 with
// FOUR waves, one per SIMD, each owning 128 (output columns) x 128 (tokens) = 8 x 8 MFMA tiles in 256 AGPRs:
//   * 16 fragment reads per 64 MFMAs (0.25 per MFMA; p5's 2 x 4 layout reads 0.375), placed BETWEEN the MFMAs of the running
//     sub-step by hand (inline asm: the instruction order below IS the issue order), no phase barriers: one s_barrier per
//     64-deep step, in the middle of it (the next step's fragments are read under the second sub-step's MFMAs);
//   * the epilogue does not store: it leaves the tile's 32 x 16 B per lane in registers ("pending") and the NEXT tile's
//     K loop issues them four per step — what p5's waves spend 3-10 us per tile waiting for (the CU takes a tile's 128 KiB of
//     stores at ~40 GB/s; profiles/r04_gemm_epilogue_experiments.txt: NO_STORE -12 .. -21 % per GEMM) runs under MFMAs.
// Operand delivery by buffer_load ... lds on whole-matrix descriptors (rows past M read as zeros; the wave-uniform part of
// an address is an SGPR offset, one VGPR holds the lane part for the whole kernel).
// Stream protocol (half-load q = 2T + h of step T lives in half-slot (q0 + q) % 5):  iteration t multiplies step t in two
// 32-deep sub-steps; sub-step 0 reads (t, 1)'s fragments and issues half-load (t+2, 0) into the slot of (t-1, 1); the mid-step
// barrier B(t+1) [this wave's pieces of step t+1 landed: vmcnt(8) lets (t+2, 0) fly; every wave's reads of step t done]
// publishes step t+1 and frees step t's slots; sub-step 1 reads (t+1, 0)'s fragments and issues (t+2, 1) into (t, 0)'s slot.
// Every half-load is issued 1-1.5 steps before the barrier that needs it, as in p5.
constexpr int kP4Threads = 256;
#ifndef RASS_P4_DEFER_MAX
#define RASS_P4_DEFER_MAX 24
#endif
#define P4_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define P4_MFMA0(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b))
#define P4_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

// One 32-deep sub-step: 64 MFMAs (acc[i][j] += a[i] * bc[j]) in issue order, and after EVERY one of them `gap(i, j)`: a single
// wave issues in order, so whatever else the sub-step has to issue — the NEXT sub-step's 16 fragment reads, its share of the
// operand DMA, a few of the previous tile's stores — goes one instruction at a time into the ~12 issue cycles each MFMA leaves
// free behind it (eight MFMAs followed by a dozen other instructions, as the first version had it, idle the matrix pipe while
// those issue: 1.9 us per 64-deep step against p5's 1.55).
template <bool ZERO, typename Gap>
__device__ __forceinline__ void p4_substep(f32x4 (&acc)[8][8], const bf16x8 (&a)[8], const bf16x8 (&bc)[8], Gap&& gap) {
#define P4_ROW(i)                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {         \
        if (ZERO) P4_MFMA0(acc[i][j], a[i], bc[j]);        \
        else P4_MFMA(acc[i][j], a[i], bc[j]);              \
        gap(i, j);                                         \
    }
    P4_ROW(0) P4_ROW(1) P4_ROW(2) P4_ROW(3) P4_ROW(4) P4_ROW(5) P4_ROW(6) P4_ROW(7)
#undef P4_ROW
}

typedef int p4_i32x4 __attribute__((ext_vector_type(4)));

template <int EPI, int POL = 1>
__global__ __launch_bounds__(kP4Threads, 1) void gemm_bf16_p4_kernel(const u16* __restrict__ X, const u16* __restrict__ W,
                                                                     const float* __restrict__ bias,
                                                                     const u16* __restrict__ residual, u16* __restrict__ Y,
                                                                     int M, int N, int K, int tiles_total, LnFold fold) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // output pieces (of 32 per lane and tile) stored by the NEXT tile's K loop, eight per step; the others in the epilogue.  24 where
    // the epilogue leaves the registers (bias only, with or without the LN fold), 16 where it also holds a residual tile.
    constexpr int kDefer = (EPI == 0 || EPI == 4) ? RASS_P4_DEFER_MAX : 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;          // which half of the tile's tokens (X rows) / output columns (W rows)
    const int G = gridDim.x, orig = blockIdx.x;
    const int pos = (G % 8 == 0) ? (orig % 8) * (G / 8) + orig / 8 : orig;
    int tile = pos;
    if (tile >= tiles_total) return;
    const int tiles_n = N / RBN;
    const int nk = K / 64;   // >= 8 (launcher)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

    // fragment offsets inside a half-slot (rows of 128 B; 16-B chunk c of row r at c ^ ((r>>1)&7); sub-step s = chunks 4s..4s+3)
    unsigned off_sub[2];
    {
        const int m = lane & 15, sw = (m >> 1) & 7;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) off_sub[sub] = m * 128 + ((sub * 4 + (lane >> 4)) ^ sw) * 16;
    }
    // operand delivery: a half-load = 16 W pieces + 16 X pieces of 8 rows x 128 B; this wave moves pieces wave + 4p, p = 0..3.
    // lane -> (row lane>>3 of the piece, 16-B chunk (lane&7) ^ ((r>>1)&7)), r = piece * 8 + (lane>>3): (r>>1)&7 =
    // (4 * (wave & 1) + (lane >> 4)) & 7 for every p (16 p = 0 mod 8).  Descriptors over the whole matrices, built by hand
    // (base, 48-bit; stride 0; bytes; raw dword format) so that they can be inline-asm operands.
    auto make_desc = [](const void* base, unsigned bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        p4_i32x4 d;
        d[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        d[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
        d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        d[3] = 0x00020000;
        return d;
    };
    const p4_i32x4 wdesc = make_desc(W, (unsigned)N * (unsigned)K * 2u);
    const p4_i32x4 xdesc = make_desc(X, (unsigned)M * (unsigned)K * 2u);
    const int dma_voff = ((lane >> 3) * K + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8)) * 2;
    const unsigned piece_step = (unsigned)K * 64u;           // bytes between pieces p and p + 1 of a wave: 32 rows of K bf16
    unsigned soW[2], soX[2];                                  // byte offset of this wave's piece 0 of half h at the stream's k
    auto point_half = [&](int t, int h) {
        const int tn0 = (t % tiles_n) * RBN, tm0 = (t / tiles_n) * RBM;
        soW[h] = __builtin_amdgcn_readfirstlane(((unsigned)(tn0 + 128 * h + wave * 8) * (unsigned)K) * 2u);
        soX[h] = __builtin_amdgcn_readfirstlane(((unsigned)(tm0 + 128 * h + wave * 8) * (unsigned)K) * 2u);
    };
    // one piece: which = 0..3 X pieces, 4..7 W pieces of half h into half-slot hs (m0 = the piece's LDS address).  In the K loop
    // the three instructions of a piece sit in three different gaps (dma_soff, dma_m0, dma_load): together behind one MFMA they
    // took ~20 issue cycles where the MFMA leaves ~12.
    const unsigned mbase = lds_base + wave * 1024;
    unsigned dma_so = 0;
    auto dma_soff = [&](int h, int which) {
        const unsigned ps = (which & 3) * piece_step;
        if (which < 4) asm volatile("s_add_u32 %0, %1, %2" : "=s"(dma_so) : "s"(soX[h]), "s"(ps) : "scc");
        else asm volatile("s_add_u32 %0, %1, %2" : "=s"(dma_so) : "s"(soW[h]), "s"(ps) : "scc");
    };
    // (m0 is written here and read by the load two gaps later; nothing the compiler emits in between touches it — this
    // kernel's LDS accesses are asm ds_read_b128, which take no m0 on gfx9+, buffer stores and scalar arithmetic)
    auto dma_m0 = [&](int hs, int which) {
        const unsigned v = mbase + hs * kP5HalfBytes + (which < 4 ? 16384 : 0) + (which & 3) * 4096;
        asm volatile("s_mov_b32 m0, %0" ::"s"(v));
    };
    auto dma_load = [&](int which) {
        if (which < 4) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dma_voff), "s"(xdesc), "s"(dma_so) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dma_voff), "s"(wdesc), "s"(dma_so) : "memory");
    };
    auto dma_piece = [&](int hs, int h, int which) {
        dma_soff(h, which);
        dma_m0(hs, which);
        asm volatile("s_nop 0");
        dma_load(which);
    };
    auto dma_half = [&](int hs, int h) {
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) dma_piece(hs, h, w8);
        soW[h] += 128;
        soX[h] += 128;
    };
    auto mod5 = [](int v) { return v >= 5 ? v - 5 : v; };

    // the stream's first four half-loads: steps 0 and 1 of the first tile
    int q0 = 0;   // half-slot of the current tile's (0, 0)
    point_half(tile, 0);
    point_half(tile, 1);
    dma_half(0, 0);
    dma_half(1, 1);
    dma_half(2, 0);
    dma_half(3, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // the previous tile's last kDefer output pieces (16 B per lane each), stored by the next tile's first two steps
    u32x4_t pending[kDefer];
#pragma unroll
    for (int i = 0; i < kDefer; ++i) pending[i] = u32x4_t{0u, 0u, 0u, 0u};
    __amdgpu_buffer_rsrc_t pdesc = __builtin_amdgcn_make_buffer_rsrc(Y, 0, 0, 0x00020000);   // zero records: the first tile's are dropped
    const int st_voff = ((lane >> 3) * N + (lane & 7) * 8) * 2;    // lane part of a store: token row lane>>3, 16 B (lane&7)
    // piece idx of a tile: 16-token chunk jc = idx >> 2, 64-column chunk ic = (idx >> 1) & 1, pass = idx & 1 (8 tokens each)
#define P4_STORE_V(val, desc, idx)                                                                                                 \
    __builtin_amdgcn_raw_buffer_store_b128(val, desc, st_voff,                                                                     \
                                           ((wm * 128 + ((idx) >> 2) * 16 + ((idx) & 1) * 8) * N + wn * 128 + (((idx) >> 1) & 1) * 64) * 2, \
                                           POL % 10 == 1 ? 2 : 0)
#define P4_STORE(v) P4_STORE_V(pending[v], pdesc, 32 - kDefer + (v))

    f32x4 acc[8][8];
    bf16x8 a[8], a6n, a7n, b0[8], b1[8];
    for (;;) {
        const int n0 = (tile % tiles_n) * RBN, m0 = (tile / tiles_n) * RBM;
        const int next = tile + G;
        const bool has_next = next < tiles_total;
        // fragments of (0, 0): the step was published by the previous tile's last mid-step barrier (or the prologue)
        {
            const unsigned ra = lds_base + (wn ? mod5(q0 + 1) : q0) * kP5HalfBytes + off_sub[0];
            const unsigned rb = lds_base + (wm ? mod5(q0 + 1) : q0) * kP5HalfBytes + off_sub[0];
            P4_READ(a[0], ra, 0); P4_READ(a[1], ra, 2048); P4_READ(a[2], ra, 4096); P4_READ(a[3], ra, 6144);
            P4_READ(a[4], ra, 8192); P4_READ(a[5], ra, 10240); P4_READ(a[6], ra, 12288); P4_READ(a[7], ra, 14336);
            P4_READ(b0[0], rb, 16384); P4_READ(b0[1], rb, 18432); P4_READ(b0[2], rb, 20480); P4_READ(b0[3], rb, 22528);
            P4_READ(b0[4], rb, 24576); P4_READ(b0[5], rb, 26624); P4_READ(b0[6], rb, 28672); P4_READ(b0[7], rb, 30720);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // The epilogue's per-column / per-token scalars (bias; LN fold: (mean, rstd) of this wave's 128 tokens, gamma / beta or
        // colsum(W') of its 128 columns) are requested NOW and ride through the K loop in 22 registers: with one wave per SIMD a
        // load at the top of the epilogue is a memory round trip nothing hides.
        f32x4 bv[2][2];
        float2 aux_c0 = {0.f, 0.f}, aux_c1 = {0.f, 0.f}, aux_mr[2] = {{0.f, 1.f}, {0.f, 1.f}};
        {
            const int nq = lane & 7;
#pragma unroll
            for (int ic = 0; ic < 2; ++ic) {
                const float* bp = bias + n0 + wn * 128 + ic * 64 + nq * 8;
                bv[ic][0] = *reinterpret_cast<const f32x4*>(bp);
                bv[ic][1] = *reinterpret_cast<const f32x4*>(bp + 4);
            }
            if constexpr (EPI >= 3) {
                aux_c0 = *reinterpret_cast<const float2*>((EPI == 3 ? fold.gamma : fold.colsum) + n0 + wn * 128 + 2 * lane);
                if constexpr (EPI == 3) aux_c1 = *reinterpret_cast<const float2*>(fold.beta + n0 + wn * 128 + 2 * lane);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int mt = m0 + wm * 128 + hh * 64 + lane;
                    aux_mr[hh] = *reinterpret_cast<const float2*>(fold.mr + 2 * (int64_t)(mt < M ? mt : 0));
                }
            }
        }
        int hs0 = q0;
        // One 64-deep step.  STORES: the first pending piece this step stores (eight of them), or -1.
        auto step = [&](int t, auto zero_c, auto stores_c) {
            constexpr bool ZERO = decltype(zero_c)::value;
            constexpr int STORES = decltype(stores_c)::value;
            const int hs1 = mod5(hs0 + 1), hs2 = mod5(hs0 + 2), hs3 = mod5(hs0 + 3), hs4 = mod5(hs0 + 4);
            // the half-loads this iteration issues are (t+2, 0) and (t+2, 1); beyond the tile they are the next tile's
            if (t + 2 == nk) {   // (no next tile: the stream re-reads this tile's first steps — in bounds, never consumed)
                point_half(has_next ? next : tile, 0);
                point_half(has_next ? next : tile, 1);
            }
            // ---- sub-step 0: multiplies (t, 0) out of a / b0, reads (t, 1) into a / b1, issues (t+2, 0) into hs4
            {
                const unsigned ra = lds_base + (wn ? hs1 : hs0) * kP5HalfBytes + off_sub[1];
                const unsigned rb = lds_base + (wm ? hs1 : hs0) * kP5HalfBytes + off_sub[1];
                p4_substep<ZERO>(acc, a, b0, [&](int i, int j) {
#define P4_GAPS(BN, HS, H)                                                                                              \
    if (j == 7) {                                                                                                       \
        if (i == 0) P4_READ(a[0], ra, 0); else if (i == 1) P4_READ(a[1], ra, 2048); else if (i == 2) P4_READ(a[2], ra, 4096); \
        else if (i == 3) P4_READ(a[3], ra, 6144); else if (i == 4) P4_READ(a[4], ra, 8192); else if (i == 5) P4_READ(a[5], ra, 10240); \
    } else if (j == 1) {                                                                                                \
        if (i == 0) P4_READ(BN[0], rb, 16384); else if (i == 1) P4_READ(BN[2], rb, 20480); else if (i == 2) P4_READ(BN[4], rb, 24576); \
        else if (i == 3) P4_READ(BN[6], rb, 28672); else if (i == 4) P4_READ(a6n, ra, 12288); else if (i == 5) P4_READ(a7n, ra, 14336); \
    } else if (j == 3) {                                                                                                \
        if (i == 0) P4_READ(BN[1], rb, 18432); else if (i == 1) P4_READ(BN[3], rb, 22528); else if (i == 2) P4_READ(BN[5], rb, 26624); \
        else if (i == 3) P4_READ(BN[7], rb, 30720);                                                                     \
    } else if (j == 0) {                                                                                                \
        dma_soff(H, i);                                                                                                 \
    } else if (j == 4) {                                                                                                \
        dma_m0(HS, i);                                                                                                  \
    } else if (j == 5) {                                                                                                \
        dma_load(i);                                                                                                    \
    }
                    P4_GAPS(b1, hs4, 0)
                    if constexpr (STORES >= 0)
                        if (j == 2 && i >= 4) {
                            if (i == 4) P4_STORE(STORES); else if (i == 5) P4_STORE(STORES + 1);
                            else if (i == 6) P4_STORE(STORES + 2); else P4_STORE(STORES + 3);
                        }
                });
                a[6] = a6n;
                a[7] = a7n;
                soW[0] += 128;
                soX[0] += 128;
            }
            // ---- the mid-step barrier B(t+1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- sub-step 1: multiplies (t, 1) out of a / b1, reads (t+1, 0) into a / b0, issues (t+2, 1) into hs0
            {
                const unsigned ra = lds_base + (wn ? hs3 : hs2) * kP5HalfBytes + off_sub[0];
                const unsigned rb = lds_base + (wm ? hs3 : hs2) * kP5HalfBytes + off_sub[0];
                p4_substep<false>(acc, a, b1, [&](int i, int j) {
                    P4_GAPS(b0, hs0, 1)
                    if constexpr (STORES >= 0)
                        if (j == 2 && i >= 4) {
                            if (i == 4) P4_STORE(STORES + 4); else if (i == 5) P4_STORE(STORES + 5);
                            else if (i == 6) P4_STORE(STORES + 6); else P4_STORE(STORES + 7);
                        }
                });
#undef P4_GAPS
                a[6] = a6n;
                a[7] = a7n;
                soW[1] += 128;
                soX[1] += 128;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            hs0 = hs2;
        };
        using std::integral_constant;
        static_assert(kDefer == 16 || kDefer == 24, "the first two or three steps store eight pending pieces each");
        step(0, integral_constant<bool, true>{}, integral_constant<int, 0>{});
        step(1, integral_constant<bool, false>{}, integral_constant<int, 8>{});
        if constexpr (kDefer == 24) step(2, integral_constant<bool, false>{}, integral_constant<int, 16>{});
        for (int t = kDefer / 8; t < nk; ++t) step(t, integral_constant<bool, false>{}, integral_constant<int, -1>{});
        // ---- epilogue: bias (+ residual / GELU), bf16, LDS transpose per wave in the ring's one free half-slot; the first
        // 32 - kDefer pieces are stored here, the rest stay in `pending` for the next tile's K loop
        q0 = hs0;                                   // the next tile's (0, 0)
        {
            constexpr int kPitchF = 68;
            float* const stg = reinterpret_cast<float*>(lds + mod5(q0 + 4) * kP5HalfBytes + wave * 8192);   // 4 352 B staging + 2 KiB aux per wave
            const int tl = lane >> 3, nq = lane & 7;
            const int rows_here = M - m0 < RBM ? (M - m0 > 0 ? M - m0 : 0) : RBM;
            // the descriptors start at the tile's first element; their size covers its rows (the columns right of the tile in
            // its last row would be in range too: they are never addressed)
            const unsigned tile_bytes = __builtin_amdgcn_readfirstlane(rows_here > 0 ? (unsigned)(rows_here - 1) * (unsigned)N * 2u + RBN * 2u : 0u);
            auto tile_desc = [&](const u16* base) {
                const uint64_t bu = reinterpret_cast<uint64_t>(base + (int64_t)m0 * N + n0);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bu), hi = __builtin_amdgcn_readfirstlane((uint32_t)(bu >> 32));
                return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u16*>(((uint64_t)hi << 32) | lo), 0, (int)tile_bytes, 0x00020000);
            };
            const __amdgpu_buffer_rsrc_t rdesc = tile_desc((EPI == 1 || EPI == 3) ? residual : Y);
            const __amdgpu_buffer_rsrc_t ydesc_now = tile_desc(Y);
            // (no explicit waits around the staging area below: a wave's LDS instructions execute in order, so its own reads see
            // its own earlier writes and the next chunk's writes cannot pass this chunk's reads; the compiler's counted waits
            // then let chunk c + 1's transposes run under chunk c's arithmetic)
            // the residual's 32 pieces of this wave's part of the tile, all requested before the first chunk is transposed (one
            // wave per SIMD: a load inside a chunk is a memory round trip nothing hides)
            u32x4_t res[(EPI == 1 || EPI == 3) ? 32 : 1];
            if (EPI == 1 || EPI == 3) {
#pragma unroll
                for (int idx = 0; idx < 32; ++idx) {
                    const int voff = ((wm * 128 + (idx >> 2) * 16 + (idx & 1) * 8 + tl) * N + wn * 128 + ((idx >> 1) & 1) * 64 + nq * 8) * 2;
                    res[idx] = __builtin_amdgcn_raw_buffer_load_b128(rdesc, voff, 0, (POL % 10 == 1 || EPI == 3) ? 2 : 0);
                }
            }
            // LN fold (EPI 3 / 4 / 5, see LnFold): this wave's 128 (mean, rstd) pairs and its 128 columns' vectors (EPI 3: gamma / beta of
            // the residual's LayerNorm, EPI 4 / 5: colsum(W')) are fetched once per tile and parked behind the staging area
            float* const aux = stg + 16 * kPitchF;      // [0, 128) gamma | colsum, [128, 256) beta, [256, 512) (mean, rstd) x 128 tokens
            if constexpr (EPI >= 3) {
                *reinterpret_cast<float2*>(aux + 2 * lane) = aux_c0;
                if constexpr (EPI == 3) *reinterpret_cast<float2*>(aux + 128 + 2 * lane) = aux_c1;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<float2*>(aux + 256 + 2 * (hh * 64 + lane)) = aux_mr[hh];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};   // EPI 3: per pass, this wave's 128-column partial sums of a token
            (void)st_s; (void)st_q;
#define P4_EPI_CHUNK(jc, ic)                                                                                                  \
    {                                                                                                                          \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                                       \
            *reinterpret_cast<f32x4*>(stg + (lane & 15) * kPitchF + ii * 16 + (lane >> 4) * 4) = acc[4 * (ic) + ii][jc];       \
        _Pragma("unroll") for (int pass = 0; pass < 2; ++pass) {                                                               \
            const int tok = pass * 8 + tl;                                                                                     \
            f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8);                                          \
            f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8 + 4);                                      \
            float mu_ = 0.f, rs_ = 1.f;                                                                                        \
            f32x4 c0_ = {0.f, 0.f, 0.f, 0.f}, c1_ = c0_, e0_ = c0_, e1_ = c0_;                                                 \
            if constexpr (EPI >= 3) {                                                                                          \
                const float2 mrv = *reinterpret_cast<const float2*>(aux + 256 + 2 * ((jc) * 16 + tok));                        \
                mu_ = mrv.x; rs_ = mrv.y;                                                                                      \
                c0_ = *reinterpret_cast<const f32x4*>(aux + (ic) * 64 + nq * 8);                                               \
                c1_ = *reinterpret_cast<const f32x4*>(aux + (ic) * 64 + nq * 8 + 4);                                           \
                if constexpr (EPI == 3) {                                                                                      \
                    e0_ = *reinterpret_cast<const f32x4*>(aux + 128 + (ic) * 64 + nq * 8);                                     \
                    e1_ = *reinterpret_cast<const f32x4*>(aux + 128 + (ic) * 64 + nq * 8 + 4);                                 \
                }                                                                                                              \
            }                                                                                                                  \
            if constexpr (EPI >= 4) {   /* rstd * (x W'^T - mu * colsum(W')) + bias' */                                        \
                const float a_ = rs_, b_ = -mu_ * rs_;                                                                         \
                v0.x = fmaf(v0.x, a_, fmaf(b_, c0_.x, bv[ic][0].x)); v0.y = fmaf(v0.y, a_, fmaf(b_, c0_.y, bv[ic][0].y));      \
                v0.z = fmaf(v0.z, a_, fmaf(b_, c0_.z, bv[ic][0].z)); v0.w = fmaf(v0.w, a_, fmaf(b_, c0_.w, bv[ic][0].w));      \
                v1.x = fmaf(v1.x, a_, fmaf(b_, c1_.x, bv[ic][1].x)); v1.y = fmaf(v1.y, a_, fmaf(b_, c1_.y, bv[ic][1].y));      \
                v1.z = fmaf(v1.z, a_, fmaf(b_, c1_.z, bv[ic][1].z)); v1.w = fmaf(v1.w, a_, fmaf(b_, c1_.w, bv[ic][1].w));      \
            } else {                                                                                                           \
                v0 += bv[ic][0];                                                                                               \
                v1 += bv[ic][1];                                                                                               \
            }                                                                                                                  \
            if constexpr (EPI == 3) {   /* residual = LayerNorm_prev(raw row), rebuilt from (raw, mu, rstd, gamma, beta) */    \
                const u32x4_t r = res[EPI == 3 ? (jc) * 4 + (ic) * 2 + pass : 0];                                              \
                const float a_ = rs_, b_ = -mu_ * rs_;                                                                         \
                v0.x += fmaf(fmaf(bf16_to_f32((u16)(r[0] & 0xffff)), a_, b_), c0_.x, e0_.x);                                   \
                v0.y += fmaf(fmaf(bf16_to_f32((u16)(r[0] >> 16)), a_, b_), c0_.y, e0_.y);                                      \
                v0.z += fmaf(fmaf(bf16_to_f32((u16)(r[1] & 0xffff)), a_, b_), c0_.z, e0_.z);                                   \
                v0.w += fmaf(fmaf(bf16_to_f32((u16)(r[1] >> 16)), a_, b_), c0_.w, e0_.w);                                      \
                v1.x += fmaf(fmaf(bf16_to_f32((u16)(r[2] & 0xffff)), a_, b_), c1_.x, e1_.x);                                   \
                v1.y += fmaf(fmaf(bf16_to_f32((u16)(r[2] >> 16)), a_, b_), c1_.y, e1_.y);                                      \
                v1.z += fmaf(fmaf(bf16_to_f32((u16)(r[3] & 0xffff)), a_, b_), c1_.z, e1_.z);                                   \
                v1.w += fmaf(fmaf(bf16_to_f32((u16)(r[3] >> 16)), a_, b_), c1_.w, e1_.w);                                      \
            }                                                                                                                  \
            if (EPI == 1) {                                                                                                    \
                const u32x4_t r = res[EPI == 1 ? (jc) * 4 + (ic) * 2 + pass : 0];                                              \
                v0.x += bf16_to_f32((u16)(r[0] & 0xffff)); v0.y += bf16_to_f32((u16)(r[0] >> 16));                             \
                v0.z += bf16_to_f32((u16)(r[1] & 0xffff)); v0.w += bf16_to_f32((u16)(r[1] >> 16));                             \
                v1.x += bf16_to_f32((u16)(r[2] & 0xffff)); v1.y += bf16_to_f32((u16)(r[2] >> 16));                             \
                v1.z += bf16_to_f32((u16)(r[3] & 0xffff)); v1.w += bf16_to_f32((u16)(r[3] >> 16));                             \
            }                                                                                                                  \
            if (EPI == 2 || EPI == 5) {                                                                                        \
                v0.x = gelu_erf(v0.x); v0.y = gelu_erf(v0.y); v0.z = gelu_erf(v0.z); v0.w = gelu_erf(v0.w);                    \
                v1.x = gelu_erf(v1.x); v1.y = gelu_erf(v1.y); v1.z = gelu_erf(v1.z); v1.w = gelu_erf(v1.w);                    \
            }                                                                                                                  \
            if constexpr (EPI == 3) {   /* the row statistics of what is STORED (the bf16 values the consumers read) */        \
                const float q0 = bf16_to_f32(f32_to_bf16(v0.x)), q1 = bf16_to_f32(f32_to_bf16(v0.y)),                          \
                            q2 = bf16_to_f32(f32_to_bf16(v0.z)), q3 = bf16_to_f32(f32_to_bf16(v0.w)),                          \
                            q4 = bf16_to_f32(f32_to_bf16(v1.x)), q5 = bf16_to_f32(f32_to_bf16(v1.y)),                          \
                            q6 = bf16_to_f32(f32_to_bf16(v1.z)), q7 = bf16_to_f32(f32_to_bf16(v1.w));                          \
                const float ss = ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7));                                            \
                float qq = q0 * q0;                                                                                            \
                qq = fmaf(q1, q1, qq); qq = fmaf(q2, q2, qq); qq = fmaf(q3, q3, qq);                                           \
                qq = fmaf(q4, q4, qq); qq = fmaf(q5, q5, qq); qq = fmaf(q6, q6, qq); qq = fmaf(q7, q7, qq);                    \
                if ((ic) == 0) { st_s[pass] = sum8_dpp(ss); st_q[pass] = sum8_dpp(qq); }                                       \
                else { st_s[pass] += sum8_dpp(ss); st_q[pass] += sum8_dpp(qq); }                                               \
                if ((ic) == 1 && nq == 0) {                                                                                    \
                    const int m_ = m0 + wm * 128 + (jc) * 16 + tok;                                                            \
                    if (m_ < M) *reinterpret_cast<float2*>(fold.stats + ((int64_t)m_ * (N / 128) + (n0 / 128 + wn)) * 2) = float2{st_s[pass], st_q[pass]}; \
                }                                                                                                              \
            }                                                                                                                  \
            const u32x4_t o = u32x4_t{(unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16),                       \
                                      (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16),                       \
                                      (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16),                       \
                                      (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16)};                      \
            if ((jc) * 4 + (ic) * 2 + pass < 32 - kDefer) P4_STORE_V(o, ydesc_now, (jc) * 4 + (ic) * 2 + pass);              \
            else pending[(jc) * 4 + (ic) * 2 + pass - (32 - kDefer)] = o;                                                    \
        }                                                                                                                      \
    }
            P4_EPI_CHUNK(0, 0) P4_EPI_CHUNK(0, 1) P4_EPI_CHUNK(1, 0) P4_EPI_CHUNK(1, 1) P4_EPI_CHUNK(2, 0) P4_EPI_CHUNK(2, 1)
            P4_EPI_CHUNK(3, 0) P4_EPI_CHUNK(3, 1) P4_EPI_CHUNK(4, 0) P4_EPI_CHUNK(4, 1) P4_EPI_CHUNK(5, 0) P4_EPI_CHUNK(5, 1)
            P4_EPI_CHUNK(6, 0) P4_EPI_CHUNK(6, 1) P4_EPI_CHUNK(7, 0) P4_EPI_CHUNK(7, 1)
#undef P4_EPI_CHUNK
            pdesc = ydesc_now;
        }
        if (!has_next) break;
        // every wave is done with its staging area: the next tile's (2, 0) may overwrite it
        __builtin_amdgcn_s_barrier();
        tile = next;
    }
    // the last tile's stores; the stream's run-on half-loads must have landed before the workgroup's LDS is released
#pragma unroll
    for (int i = 0; i < kDefer; ++i) P4_STORE(i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef P4_STORE
#undef P4_STORE_V
}

template <int EPI, int POL>
static hipError_t launch_p4_pol(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M, int N,
                                int K, int tiles_total, int grid, hipStream_t stream, const LnFold& fold) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_p4_kernel<EPI, POL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kP5LdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_p4_kernel<EPI, POL>), dim3(grid), dim3(kP4Threads), kP5LdsBytes, stream, X, W, bias, residual, Y, M,
                       N, K, tiles_total, fold);
    return hipGetLastError();
}

// The persistent GEMM of big shapes is p4 since round 4; RASS_GEMM_VARIANT=p5 brings back the 8-wave kernel (the A/B; same bits).
static bool p4_enabled(int epi) {
    const char* v = rass_env("RASS_GEMM_VARIANT");   // read per launch: the A/B scripts flip it inside one process
    if (v != nullptr && strcmp(v, "p5") == 0) return false;
    if (v != nullptr && strcmp(v, "p4") == 0) return true;
    return epi != 5;
}

template <int EPI>
static hipError_t launch_p5(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M,
                            int M_pad, int N, int K, hipStream_t stream, const LnFold& fold = LnFold{}) {
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0)
            n_cus = 256;
    }
    const int tiles_total = (N / RBN) * (M_pad / RBM);
    int grid = tiles_total < n_cus ? tiles_total : n_cus;
    if (const char* v = rass_env("RASS_GEMM_GRID")) {   // experiment: fewer persistent workgroups than CUs (per-CU vs chip-wide limits)
        const int g = atoi(v);
        if (g >= 1 && g < grid) grid = g;
    }
    // (the folded GELU epilogue, EPI 5, is the one p4 loses: 1 165 vs 1 129 us per FFN-up — a single wave per SIMD has nothing to
    // overlap that epilogue's dependency stalls with; it stays on p5 unless RASS_GEMM_VARIANT=p4 asks for p4 everywhere)
    if (p4_enabled(EPI) && K >= 512 && (uint64_t)M_pad * K * 2 < (1ull << 32) - (1ull << 24) && (uint64_t)N * K * 2 < (1ull << 32) - (1ull << 24))
        return launch_p4_pol<EPI, 1>(X, W, bias, residual, Y, M, N, K, tiles_total, grid, stream, fold);
    if (const char* v = rass_env("RASS_P5_POLICY"))   // A/B: 0 = plain output stores (read per launch)
        if (atoi(v) == 0) return launch_p5_pol<EPI, 0>(X, W, bias, residual, Y, M, N, K, tiles_total, grid, stream, fold);
    return launch_p5_pol<EPI, 1>(X, W, bias, residual, Y, M, N, K, tiles_total, grid, stream, fold);
}

// ---- LN fold: the big-batch forward without the stand-alone LayerNorm passes (see LnFold above) -----------------------
static bool p5_eligible(int M, int M_pad, int N, int K) {
    return N % RBN == 0 && M_pad % RBM == 0 && K % 64 == 0 && K >= 128 && M >= 1024 && (int64_t)(N / RBN) * (M_pad / RBM) >= 192;
}

bool gemm_bf16_fold_ok(int M, int M_pad, int hidden, int intermediate) {
    if (const char* v = rass_env("RASS_ENCODER_LN_FOLD"))
        if (atoi(v) == 0) return false;
    return hidden % 256 == 0 && p5_eligible(M, M_pad, hidden, hidden) && p5_eligible(M, M_pad, 3 * hidden, hidden) &&
           p5_eligible(M, M_pad, intermediate, hidden) && p5_eligible(M, M_pad, hidden, intermediate);
}

hipError_t launch_gemm_bf16_fold(const void* X, const void* W, const float* bias, const void* residual_raw, void* Y, int M,
                                 int M_pad, int N, int K, int epilogue, const float* mr, const float* gamma, const float* beta,
                                 float* stats, const float* colsum, hipStream_t stream) {
    if (!p5_eligible(M, M_pad, N, K) || !mr) return hipErrorInvalidValue;
    LnFold f;
    f.mr = mr;
    f.gamma = gamma;
    f.beta = beta;
    f.stats = stats;
    f.colsum = colsum;
    const u16* x = static_cast<const u16*>(X);
    const u16* w = static_cast<const u16*>(W);
    const u16* r = static_cast<const u16*>(residual_raw);
    u16* y = static_cast<u16*>(Y);
    switch (epilogue) {
        case 3: return (r && gamma && beta && stats) ? launch_p5<3>(x, w, bias, r, y, M, M_pad, N, K, stream, f) : hipErrorInvalidValue;
        case 4: return colsum ? launch_p5<4>(x, w, bias, nullptr, y, M, M_pad, N, K, stream, f) : hipErrorInvalidValue;
        case 5: return colsum ? launch_p5<5>(x, w, bias, nullptr, y, M, M_pad, N, K, stream, f) : hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// (mean, rstd) per row from the residual GEMM's per-chunk partial sums, in fixed order
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ stats, int chunks, int n, float eps,
                                                                float* __restrict__ mr, int rows) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= rows) return;
    const float2* p = reinterpret_cast<const float2*>(stats) + (int64_t)m * chunks;
    float s = 0.f, q = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float2 v = p[c];
        s += v.x;
        q += v.y;
    }
    const float mean = s / (float)n;
    const float var = fmaxf(q / (float)n - mean * mean, 0.f);
    *reinterpret_cast<float2*>(mr + 2 * (int64_t)m) = float2{mean, rsqrtf(var + eps)};
}

hipError_t launch_ln_stats_finalize(const float* stats, int rows, int n, float eps, float* mr, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (n % 128 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, stats, n / 128, n, eps, mr, rows);
    return hipGetLastError();
}

// W'[n][k] = bf16(W[n][k] * gamma[k]);  colsum[n] = sum_k W'[n][k];  bias2[n] = bias[n] + sum_k beta[k] * W[n][k].  One wave per row.
__global__ __launch_bounds__(256) void fold_gamma_kernel(const u16* __restrict__ W, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ bias,
                                                         int N, int K, u16* __restrict__ W2, float* __restrict__ colsum,
                                                         float* __restrict__ bias2) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float cs = 0.f, bs = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = bf16_to_f32(W[(int64_t)n * K + k]);
        const u16 w2 = f32_to_bf16(w * gamma[k]);
        W2[(int64_t)n * K + k] = w2;
        cs += bf16_to_f32(w2);
        bs = fmaf(beta[k], w, bs);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cs += __shfl_xor(cs, off);
        bs += __shfl_xor(bs, off);
    }
    if (lane == 0) {
        colsum[n] = cs;
        bias2[n] = bias[n] + bs;
    }
}

hipError_t launch_fold_gamma(const void* W, const float* gamma, const float* beta, const float* bias, int N, int K, void* W2,
                             float* colsum, float* bias2, hipStream_t stream) {
    hipLaunchKernelGGL(fold_gamma_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, static_cast<const u16*>(W), gamma, beta, bias,
                       N, K, static_cast<u16*>(W2), colsum, bias2);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_epi(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M,
                             int M_pad, int N, int K, hipStream_t stream) {
    // big shapes: the persistent 256^2 kernel (p5); everything else: the 128^2 kernel.  "Big" = enough 256^2 tiles to
    // keep most of the chip's CUs busy (a persistent kernel runs one tile per CU at a time): a 2 048-token upload has
    // 32 tiles at N = 1024 and ran on 32 of 256 CUs; as 128^2 tiles (split over K where those are few) it fills the
    // chip.  RASS_GEMM_VARIANT=p5 keeps the persistent kernel for every shape it accepts (A/B runs, tests).
    static const bool forced = rass_env("RASS_GEMM_VARIANT") != nullptr &&
                               (strcmp(rass_env("RASS_GEMM_VARIANT"), "p5") == 0 || strcmp(rass_env("RASS_GEMM_VARIANT"), "p4") == 0);
    const bool enough_tiles = forced || (int64_t)(N / RBN) * (M_pad / RBM) >= 192;
    if (N % RBN == 0 && M_pad % RBM == 0 && K % 64 == 0 && K >= 128 && M >= 1024 && enough_tiles)
        return launch_p5<EPI>(X, W, bias, residual, Y, M, M_pad, N, K, stream);
    const int grid = (N / GBN) * (M_pad / GBM);
    if (mid_enabled(M) && K >= 4 * GBK && (uint64_t)M_pad * K * 2 < (1ull << 32) - (1ull << 24) && (uint64_t)N * K * 2 < (1ull << 32) - (1ull << 24)) {
        // the four-stage form of the tile, over the row tiles that hold real rows; 64-row tiles while 128-row ones would leave
        // CUs idle (RASS_GEMM_MID_BM=128 / 64: the A/B)
        static bool mid_attr_set = false;
        if (!mid_attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_mid_kernel<EPI, 128>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kMidLdsBytes);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_mid_kernel<EPI, 64>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kMidLdsBytes);
            if (e != hipSuccess) return e;
            mid_attr_set = true;
        }
        int bm = (N / GBN) * ((M + 63) / 64) <= 256 ? 64 : 128;   // 64-row tiles while they still fit one per CU
        if (const char* v = rass_env("RASS_GEMM_MID_BM")) bm = atoi(v) == 128 ? 128 : 64;
        if (bm == 128 || M_pad % 64 != 0) {
            hipLaunchKernelGGL((gemm_bf16_mid_kernel<EPI, 128>), dim3((N / GBN) * ((M + 127) / 128)), dim3(kGemmThreads), kMidLdsBytes,
                               stream, X, W, bias, residual, Y, M, N, K);
        } else {
            hipLaunchKernelGGL((gemm_bf16_mid_kernel<EPI, 64>), dim3((N / GBN) * ((M + 63) / 64)), dim3(kGemmThreads), kMidLdsBytes,
                               stream, X, W, bias, residual, Y, M, N, K);
        }
        return hipGetLastError();
    }
    constexpr int lds_bytes = 4 * kTileBytes;  // 64 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(grid), dim3(kGemmThreads), lds_bytes, stream, X, W, bias,
                       residual, Y, M, N, K);
    return hipGetLastError();
}

hipError_t launch_gemm_bf16(const void* X, const void* W, const float* bias, const void* residual, void* Y, int M,
                            int M_pad, int N, int K, int epilogue, hipStream_t stream, float* splitk_ws,
                            size_t splitk_ws_bytes) {
    if (M < 0 || M_pad < M || N <= 0 || K <= 0) return hipErrorInvalidValue;
    // every kernel here works on whole 128-row / 128-column tiles and 64-deep K steps (include/rass_engine.h)
    if (M_pad % GBM != 0 || N % GBN != 0 || K % GBK != 0) return hipErrorInvalidValue;
    if (M == 0) return hipSuccess;
    const u16* x = static_cast<const u16*>(X);
    const u16* w = static_cast<const u16*>(W);
    const u16* r = static_cast<const u16*>(residual);
    u16* y = static_cast<u16*>(Y);
    if (epilogue < 0 || epilogue > 2 || (epilogue == 1 && !r)) return hipErrorInvalidValue;
    // a few rows against a wide matrix: one launch, epilogue included (query-time embedding; chosen with the scratch
    // lent, i.e. on the same calls that would otherwise be split over K)
    if (const int fw = splitk_ws != nullptr && M_pad >= 64 && fewrows_enabled() ? fewrows_waves(M, N, K) : 0) {
        switch (epilogue) {
            case 0: return launch_fewrows<0>(x, w, bias, r, y, M, N, K, fw, stream);
            case 1: return launch_fewrows<1>(x, w, bias, r, y, M, N, K, fw, stream);
            default: return launch_fewrows<2>(x, w, bias, r, y, M, N, K, fw, stream);
        }
    }
    // few rows: split K over more workgroups (the caller lends the fp32 scratch); with the four-stage kernel a short K
    // (<= 16 steps) is not split any more: one launch with the epilogue fused beats the pair (see gemm_bf16_mid_kernel)
    if (splitk_ws != nullptr && M_pad % GBM == 0 && N % GBN == 0 && K % GBK == 0 && !(mid_enabled(M) && K <= 1024 && K >= 4 * GBK)) {
        const int mp = (M + GBM - 1) / GBM * GBM;   // whole 128-row tiles that hold real rows (<= M_pad)
        const int S = splitk_slices(mp, N, K, splitk_ws_bytes);
        if (S > 0) {
            const int M_pad = mp;
            constexpr int lds_bytes = 4 * kTileBytes;
            static bool attr_set = false;
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_splitk_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                if (e != hipSuccess) return e;
                attr_set = true;
            }
            hipLaunchKernelGGL(gemm_bf16_splitk_kernel, dim3((N / GBN) * (M_pad / GBM), S), dim3(kGemmThreads), lds_bytes,
                               stream, x, w, splitk_ws, M, M_pad, N, K, K / S);
            const int64_t work = (int64_t)M * (N / 4);
            const unsigned blocks = (unsigned)((work + 255) / 256);
            if (epilogue == 0)
                hipLaunchKernelGGL(splitk_epilogue_kernel<0>, dim3(blocks), dim3(256), 0, stream, splitk_ws, S, M, M_pad, N, bias, r, y);
            else if (epilogue == 1)
                hipLaunchKernelGGL(splitk_epilogue_kernel<1>, dim3(blocks), dim3(256), 0, stream, splitk_ws, S, M, M_pad, N, bias, r, y);
            else
                hipLaunchKernelGGL(splitk_epilogue_kernel<2>, dim3(blocks), dim3(256), 0, stream, splitk_ws, S, M, M_pad, N, bias, r, y);
            return hipGetLastError();
        }
    }
    switch (epilogue) {
        case 0: return launch_epi<0>(x, w, bias, r, y, M, M_pad, N, K, stream);
        case 1: return r ? launch_epi<1>(x, w, bias, r, y, M, M_pad, N, K, stream) : hipErrorInvalidValue;
        case 2: return launch_epi<2>(x, w, bias, r, y, M, M_pad, N, K, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gemm_bf16_residual_layernorm(const void* X, const void* W, const float* bias, const void* residual,
                                               void* y, const float* gamma, const float* beta, float eps, void* out,
                                               int M, int M_pad, int N, int K, hipStream_t stream, float* splitk_ws,
                                               size_t splitk_ws_bytes) {
    if (M < 0 || M_pad < M || N <= 0 || K <= 0 || !residual) return hipErrorInvalidValue;
    if (M == 0) return hipSuccess;
    // a query's few rows: the one-launch GEMM (bias + residual in its epilogue) and the row-wise LayerNorm — two launches
    // like the split-K pair below, but 5 + 5 us where that pair takes 6 + 7.4 (16 slices read back by 16 waves)
    // (a K of whole 4096s runs as four K slices of 4-wave workgroups: what must fit is a slice)
    if (splitk_ws != nullptr && M_pad >= 64 && fewrows_enabled() && M <= fewrows_residual_max_rows() &&
        fewrows_waves(M, N, K % 4096 == 0 ? K / 4 : K) != 0) {
        // K = 4096 (FFN-down): 64 workgroups of 16 waves took 9.4 us; 4 x 64 workgroups of 4 waves write partial tiles and
        // the fused reduce + residual + LayerNorm kernel (4 slices) follows
        const int rows_pad = M <= 64 ? 64 : 128;
        if (K % 4096 == 0 && K / 4 <= 3072 && N % 8 == 0 && N <= 2048 && (size_t)4 * rows_pad * N * sizeof(float) <= splitk_ws_bytes) {
            hipError_t e = launch_fewrows_w<-1, 4>(static_cast<const u16*>(X), static_cast<const u16*>(W), nullptr, nullptr,
                                                   nullptr, M, N, K, stream, splitk_ws, rows_pad, 4);
            if (e != hipSuccess) return e;
            return launch_splitk_residual_layernorm(splitk_ws, 4, M, rows_pad, N, bias, residual, gamma, beta, eps, out, stream);
        }
        hipError_t e = launch_gemm_bf16(X, W, bias, residual, y, M, M_pad, N, K, 1, stream, splitk_ws, splitk_ws_bytes);
        if (e != hipSuccess) return e;
        return launch_layernorm(y, gamma, beta, eps, M, N, out, stream);
    }
    if (splitk_ws != nullptr && M_pad % GBM == 0 && N % GBN == 0 && K % GBK == 0 && N % 8 == 0 && N <= 2048 &&
        !(mid_enabled(M) && K <= 1024 && K >= 4 * GBK)) {
        const int mp = (M + GBM - 1) / GBM * GBM;
        const int S = splitk_slices(mp, N, K, splitk_ws_bytes);
        if (S > 0) {
            constexpr int lds_bytes = 4 * kTileBytes;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_splitk_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(gemm_bf16_splitk_kernel, dim3((N / GBN) * (mp / GBM), S), dim3(kGemmThreads), lds_bytes,
                               stream, static_cast<const u16*>(X), static_cast<const u16*>(W), splitk_ws, M, mp, N, K, K / S);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            return launch_splitk_residual_layernorm(splitk_ws, S, M, mp, N, bias, residual, gamma, beta, eps, out, stream);
        }
    }
    hipError_t e = launch_gemm_bf16(X, W, bias, residual, y, M, M_pad, N, K, 1, stream, splitk_ws, splitk_ws_bytes);
    if (e != hipSuccess) return e;
    return launch_layernorm(y, gamma, beta, eps, M, N, out, stream);
}

bool gemm_bf16_ln_input_ok(int M, int N, int K) {
    return M >= 1 && M <= 32 && K == 1024 && N % 16 == 0 && N >= 1024 && fewrows_enabled();
}

template <int EPI, int ROWS, int WAVES>
static hipError_t launch_lnin(const u16* yin, const float* gamma, const float* beta, float eps, u16* x_out, const u16* w,
                              const float* bias, u16* y, int M, int N, hipStream_t stream) {
    constexpr int lds_bytes = 16 * ROWS * (1024 + 8) * 2 + WAVES * ROWS * 64 * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_lnin_kernel<EPI, ROWS, WAVES>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_lnin_kernel<EPI, ROWS, WAVES>), dim3(N / 16), dim3(64 * WAVES), lds_bytes, stream, yin, gamma,
                       beta, eps, x_out, w, bias, y, M, N);
    return hipGetLastError();
}

template <int EPI, int WAVES>
static hipError_t launch_lnin_rows(const u16* yin, const float* gamma, const float* beta, float eps, u16* xo, const u16* w,
                                   const float* bias, u16* y, int M, int N, hipStream_t stream) {
    return M <= 16 ? launch_lnin<EPI, 1, WAVES>(yin, gamma, beta, eps, xo, w, bias, y, M, N, stream)
                   : launch_lnin<EPI, 2, WAVES>(yin, gamma, beta, eps, xo, w, bias, y, M, N, stream);
}

hipError_t launch_gemm_bf16_ln_input(const void* Yin, const float* gamma, const float* beta, float eps, void* x_out,
                                     const void* W, const float* bias, void* Y, int M, int N, int K, int epilogue,
                                     hipStream_t stream) {
    if (!gemm_bf16_ln_input_ok(M, N, K) || (epilogue != 0 && epilogue != 2)) return hipErrorInvalidValue;
    const u16* yin = static_cast<const u16*>(Yin);
    const u16* w = static_cast<const u16*>(W);
    u16* xo = static_cast<u16*>(x_out);
    u16* y = static_cast<u16*>(Y);
    const char* v = rass_env("RASS_GEMM_LNIN_WAVES");   // 4: the 4-wave workgroups of rounds 2-3 (A/B; read per launch)
    if (v && atoi(v) == 4)
        return epilogue == 0 ? launch_lnin_rows<0, 4>(yin, gamma, beta, eps, xo, w, bias, y, M, N, stream)
                             : launch_lnin_rows<2, 4>(yin, gamma, beta, eps, xo, w, bias, y, M, N, stream);
    return epilogue == 0 ? launch_lnin_rows<0, 16>(yin, gamma, beta, eps, xo, w, bias, y, M, N, stream)
                         : launch_lnin_rows<2, 16>(yin, gamma, beta, eps, xo, w, bias, y, M, N, stream);
}

}  // namespace rass
