// encoder_attn.hip — K6 of SURVEY §8a: varlen multi-head self-attention for the sentence
// encoder (BERT: bidirectional, heads of 64, S <= 512, softmax scale 1/8, key padding never
// materialised because sequences are packed).  Replaces llama.cpp's attention behind Ollama.
//
// One workgroup (16 waves) per (sequence, head); the head's K [S][64] and V [S][64] live in LDS
// for the whole workgroup (S <= 512: 64 KiB + 80 KiB), so K/V are fetched from HBM/L2 once per
// (sequence, head).  Each wave owns 16 queries per pass and streams 64-key blocks with an online
// softmax.
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16), chosen so NOTHING is transposed between the
// two products (guide §3 "an accumulator tile as the next MFMA's operand"):
//   S^T[key][q] = K[key][:] . Q[q][:]     A = K rows from LDS, B = Q (registers)
//       -> lane (q = lane&15, g = lane>>4) holds scores of keys 16*kt + 4g + {0..3}
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]  B = the lane's own exponentiated scores packed
//       to bf16 (k index 8g+j <-> key 32*kk + (j<4 ? 4g+j : 16+4g+j-4)), A = V^T taken from the
//       ROW-major V image with two ds_read_b64_tr_b16 (gfx950 transposing LDS read: a 16-lane
//       group reads 4 keys x 16 d and lane i receives column d = i of the 4 keys) — V is staged
//       with plain 16-byte writes like K, no transposed copy is built (the 2-byte scattered
//       writes of a V^T image were 8-way bank conflicted: ~20 % of the kernel).
// The softmax statistics of query q live in the 4 lanes that share lane&15.

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#include "encoder_kernels.h"

namespace rass {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

// 16 waves share one K / V^T image: the 130 KiB of LDS allow only one workgroup per CU, so
// the workgroup itself must bring 4 waves per SIMD to hide LDS / MFMA / shuffle latency
// (4 waves measured 1.31 ms per layer at batch 256 x 512: one wave per SIMD, latency-bound).
constexpr int kAttnThreads = 1024;
constexpr int kAttnWaves = kAttnThreads / 64;
constexpr int kHeadDim = 64;
constexpr int kVPitch = 160;  // bytes per V row in LDS (128 B of data)

__device__ __forceinline__ u16 f2bf_a(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<u16*>(&h);
}

__global__ __launch_bounds__(kAttnThreads) void attention_kernel(const u16* __restrict__ qkv,
                                                                 const int32_t* __restrict__ cu, int hidden, int heads,
                                                                 int s_pad, u16* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Kl = lds;                              // [s_pad][128 B], chunk-swizzled
    unsigned char* Vl = lds + (size_t)s_pad * 128;        // [s_pad][kVPitch B]: row stride 40 dwords, so the 8
                                                          // rows x 32 B one half-wave's transposed read touches
                                                          // fall on 64 distinct banks
    const int seq = blockIdx.x / heads, head = blockIdx.x % heads;
    const int t0 = cu[seq];
    int S = cu[seq + 1] - t0;
    if (S > s_pad) S = s_pad;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ld = 3 * hidden;
    const u16* qbase = qkv + (int64_t)t0 * ld + head * kHeadDim;
    const u16* kbase = qbase + hidden;
    const u16* vbase = qbase + 2 * hidden;

    // ---- stage K (swizzled rows) and V (padded rows); keys >= S are zero (masked later, must be finite)
    for (int e = threadIdx.x; e < s_pad * 8; e += kAttnThreads) {
        const int key = e >> 3, c = e & 7;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < S) {
            kv = *reinterpret_cast<const uint4*>(kbase + (int64_t)key * ld + c * 8);
            vv = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * ld + c * 8);
        }
        *reinterpret_cast<uint4*>(Kl + key * 128 + ((c ^ ((key >> 1) & 7)) * 16)) = kv;
        *reinterpret_cast<uint4*>(Vl + key * kVPitch + c * 16) = vv;
    }
    __syncthreads();

    const int g = lane >> 4, qi = lane & 15;
    const int n_kb = (S + 63) / 64;
    for (int q0 = wave * 16; q0 < S; q0 += kAttnWaves * 16) {
        const int q = q0 + qi;
        // Q fragments (B operand): Q[q][8g + 32ks .. +7]
        bf16x8 qf[2];
        {
            const int qc = q < S ? q : S - 1;
            qf[0] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 8 * g);
            qf[1] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 32 + 8 * g);
        }
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 O[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) O[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kb = 0; kb < n_kb; ++kb) {
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int krow = kb * 64 + kt * 16 + qi;  // A operand row = key (lane&15)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int c = (ks * 4 + g) ^ ((krow >> 1) & 7);
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kl + krow * 128 + c * 16);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], s[kt], 0, 0, 0);
                }
            }
            // s[kt][r] is the raw score of key kb*64 + kt*16 + 4g + r.  The softmax runs in the exp2
            // domain with the scale (1/8 * log2 e) folded into the exponent's fma, and is written to
            // keep the VALU count per score low (this kernel is VALU-bound: 16 x 512 scores per wave
            // and pass against 2 x 32 MFMAs): no per-score mask outside the last key block, max3
            // chains, packed adds, and the O / l rescale only in blocks where some query's max grew.
            constexpr float kScale = 0.18033688011112042f;
            if (kb * 64 + 64 > S) {  // wave-uniform: only the last block of a ragged sequence
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb * 64 + kt * 16 + 4 * g + r >= S) s[kt][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) mx = fmaxf(fmaxf(mx, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));  // (gfx950's v_permlane16/32_swap measured the same)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * kScale);  // finite: key 0 of block 0 is always valid
            if (__any(m_new > m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 1 where the max did not move
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) O[dt] *= alpha;
                m_run = m_new;
            }
            const f32x4 negm = f32x4{-m_run, -m_run, -m_run, -m_run};
            f32x4 rs4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 e = s[kt] * kScale + negm;  // v_pk_fma_f32 x2
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(e[r]);
                rs4 += s[kt];
            }
            float rs = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l_run += rs;
            // P^T fragments (B operand) for the two 32-key steps of this block
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (short)f2bf_a(s[2 * kk][r]);
                    pf[4 + r] = (short)f2bf_a(s[2 * kk + 1][r]);
                }
                // transposed read: lane 4q+p of a 16-lane group addresses key row q, columns 4p..4p+3
                const unsigned char* vblk = Vl + (kb * 64 + kk * 32 + 4 * g + (qi >> 2)) * kVPitch + (qi & 3) * 8;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(vblk + dt * 32));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(vblk + 16 * kVPitch + dt * 32));
                    bf16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    O[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, O[dt], 0, 0, 0);
                }
            }
        }
        // O^T[d = dt*16 + 4g + r][q]: 4 consecutive d of query q -> one 8-byte store
        if (q < S) {
            const float inv_l = 1.f / l_run;
            u16* dst = ctx + (int64_t)(t0 + q) * hidden + head * kHeadDim;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 o;
                o.x = (unsigned)f2bf_a(O[dt][0] * inv_l) | ((unsigned)f2bf_a(O[dt][1] * inv_l) << 16);
                o.y = (unsigned)f2bf_a(O[dt][2] * inv_l) | ((unsigned)f2bf_a(O[dt][3] * inv_l) << 16);
                *reinterpret_cast<uint2*>(dst + dt * 16 + 4 * g) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A query's few rows (all sequences of the batch together <= 32 tokens: embed_query / ollama_embed_text,
// app/main.py:225-237, 266-274): the attention RECOMPUTED inside the attention-output GEMM (round 4).  Every kernel of a
// one-query forward costs ~4-5 us whatever it does, and this attention is 16 heads x a 16 x 16 score tile:
// each of the N / 16 workgroups of the few-rows GEMM (16 output features, encoder_gemm.hip gemm_bf16_fewrows_kernel) runs
// it again instead of reading `ctx` from a launch of its own.  A workgroup has 16 waves, wave = head = a 64-deep slice of
// the GEMM's K: the wave
//   1. has its two weight fragments W[n0 + i][64 head + 32 u + 8 g ..] in flight first,
//   2. per sequence: stages the head's V rows in its OWN piece of LDS, takes K and Q fragments straight from global
//      memory / L2 and does attention_kernel's arithmetic on the one 64-key block there is (same MFMA orientation, the
//      same order of every fp32 operation: the context rows have attention_kernel's bits),
//   3. writes the 16 x 64 context block (bf16) into its private operand tile and reads it back as the GEMM's B fragments
//      (LDS is in order per wave: no barrier), two MFMAs per 16-token block,
//   4. leaves its partial tile in LDS; wave 0 adds the 16 partial tiles in wave order, bias and residual.
// QKV is re-read 64 times from L2 (96 KiB per workgroup at 16 tokens).  The partial tiles are summed in a different
// order than the 4-wave GEMM's, so Y may differ from the unfused pair in the last bit of a bf16 here and there
// (test_gpu_encoder.py bounds it); RASS_ATTN_FUSE=0 keeps the pair.
constexpr int kFusedXPitch = 72;  // u16 per row of a wave's context tile: 64 + 8 (rows 144 B apart: 16-byte aligned, conflict-free)
constexpr int kFusedVRows = 32;   // V rows staged per sequence (rows past the sequence are zeros: P is 0 there, V must be finite)

template <int ROWS>
__global__ __launch_bounds__(1024) void attn_out_fewrows_kernel(const u16* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                                int nseq, int M, const u16* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                const u16* __restrict__ residual, u16* __restrict__ Y, int N) {
    constexpr int hidden = 16 * kHeadDim, K = hidden, ld = 3 * hidden;
    constexpr int kWaveBytes = kFusedVRows * kVPitch + 16 * ROWS * kFusedXPitch * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;   // = head
    unsigned char* Vl = lds + wave * kWaveBytes;
    u16* xs = reinterpret_cast<u16*>(Vl + kFusedVRows * kVPitch);
    const int n0 = blockIdx.x * 16;
    const int g = lane >> 4, qi = lane & 15;
    const u16* wrow = W + (int64_t)(n0 + qi) * K + wave * kHeadDim + 8 * g;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wrow);
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wrow + 32);
    // wave 0's epilogue operands leave now
    const int n = n0 + 4 * g;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 res[ROWS];
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) res[rb] = make_uint2(0, 0);
    if (wave == 0) {
        bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int rb = 0; rb < ROWS; ++rb) {
            const int m = 16 * rb + qi;
            res[rb] = *reinterpret_cast<const uint2*>(residual + (int64_t)(m < M ? m : 0) * N + n);
        }
    }
    // the operand tile starts as zeros (rows past M stay so)
    for (int e = lane; e < 16 * ROWS * 9; e += 64)
        *reinterpret_cast<uint4*>(xs + (e / 9) * kFusedXPitch + (e % 9) * 8) = make_uint4(0, 0, 0, 0);

    constexpr float kScale = 0.18033688011112042f;
#ifdef RASS_ELIM_ATT   // elimination build (timing only, wrong results): no attention, the operand tile stays zero
    nseq = 0;
#endif
    for (int s = 0; s < nseq; ++s) {
        // one sequence: it is the whole batch (the host knows M; no dependent load in front of the Q / K / V loads)
        const int t0 = nseq == 1 ? 0 : cu[s];
        const int S = nseq == 1 ? M : cu[s + 1] - t0;
        if (S <= 0 || S > 16 * ROWS || t0 < 0 || t0 + S > M) continue;   // (the host checked; never index past the tiles)
        const u16* qbase = qkv + (int64_t)t0 * ld + wave * kHeadDim;
        const u16* kbase = qbase + hidden;
        const u16* vbase = qbase + 2 * hidden;
        // V rows of this sequence and head: 32 rows x 8 pieces of 16 B, 4 per lane
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = (lane >> 3) + 8 * j, c = lane & 7;
            uint4 vv = make_uint4(0, 0, 0, 0);
            if (key < S) vv = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * ld + c * 8);
            *reinterpret_cast<uint4*>(Vl + key * kVPitch + c * 16) = vv;
        }
        asm volatile("" ::: "memory");   // the transposing reads below come after these writes (LDS is in order per wave)
        // K fragments (A operand): key 16 kt + qi; keys past S are masked below (any finite row will do)
        bf16x8 kf[ROWS][2];
#pragma unroll
        for (int kt = 0; kt < ROWS; ++kt) {
            const int key = 16 * kt + qi;
            const u16* kp = kbase + (int64_t)(key < S ? key : S - 1) * ld + 8 * g;
            kf[kt][0] = *reinterpret_cast<const bf16x8*>(kp);
            kf[kt][1] = *reinterpret_cast<const bf16x8*>(kp + 32);
        }
#pragma unroll
        for (int qb = 0; qb < ROWS; ++qb) {
            if (16 * qb >= S) break;
            const int q = 16 * qb + qi;
            const u16* qp = qbase + (int64_t)(q < S ? q : S - 1) * ld + 8 * g;
            const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp);
            const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(qp + 32);
            f32x4 sc[2];
            sc[1] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int kt = 0; kt < ROWS; ++kt) {
                sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf0, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], qf1, sc[kt], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * kt + 4 * g + r >= S) sc[kt][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
            mx = fmaxf(fmaxf(mx, fmaxf(sc[1][0], sc[1][1])), fmaxf(sc[1][2], sc[1][3]));
            mx = fmaxf(mx, wave_xor_partner_dpp(mx, lane, 16));   // = __shfl_xor without the LDS crossbar (encoder_kernels.h)
            mx = fmaxf(mx, wave_xor_partner_dpp(mx, lane, 32));
            const float m_run = mx * kScale;   // finite: key 0 is valid
            const f32x4 negm = f32x4{-m_run, -m_run, -m_run, -m_run};
            f32x4 rs4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const f32x4 e = sc[kt] * kScale + negm;
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[kt][r] = __builtin_amdgcn_exp2f(e[r]);
                rs4 += sc[kt];
            }
            float rs = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
            rs += wave_xor_partner_dpp(rs, lane, 16);
            rs += wave_xor_partner_dpp(rs, lane, 32);
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (short)f2bf_a(sc[0][r]);
                pf[4 + r] = (short)f2bf_a(sc[1][r]);
            }
            const unsigned char* vblk = Vl + (4 * g + (qi >> 2)) * kVPitch + (qi & 3) * 8;
            const float inv_l = 1.f / rs;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(vblk + dt * 32));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(vblk + 16 * kVPitch + dt * 32));
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                f32x4 O = f32x4{0.f, 0.f, 0.f, 0.f};
                O = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, O, 0, 0, 0);
                if (q < S) {
                    uint2 o;
                    o.x = (unsigned)f2bf_a(O[0] * inv_l) | ((unsigned)f2bf_a(O[1] * inv_l) << 16);
                    o.y = (unsigned)f2bf_a(O[2] * inv_l) | ((unsigned)f2bf_a(O[3] * inv_l) << 16);
                    *reinterpret_cast<uint2*>(xs + (t0 + q) * kFusedXPitch + dt * 16 + 4 * g) = o;
                }
            }
        }
        asm volatile("" ::: "memory");   // the next sequence's V rows go where these were read
    }
    asm volatile("" ::: "memory");
    // the GEMM's share of this wave: K slice [64 head, 64 head + 64)
    f32x4* part = reinterpret_cast<f32x4*>(Vl);   // [ROWS][64], over the wave's own V rows (its reads are done: in order)
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) {
        const u16* xr = xs + (16 * rb + qi) * kFusedXPitch + 8 * g;
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(xr);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(xr + 32);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
        part[rb * 64 + lane] = acc;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int rb = 0; rb < ROWS; ++rb) {
        const int m = 16 * rb + qi;
        if (m >= M) continue;
        f32x4 v = part[rb * 64 + lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) v += reinterpret_cast<const f32x4*>(lds + w * kWaveBytes)[rb * 64 + lane];
        v += bv;
        v.x += __uint_as_float(res[rb].x << 16);
        v.y += __uint_as_float(res[rb].x & 0xffff0000u);
        v.z += __uint_as_float(res[rb].y << 16);
        v.w += __uint_as_float(res[rb].y & 0xffff0000u);
        uint2 o;
        o.x = (unsigned)f2bf_a(v.x) | ((unsigned)f2bf_a(v.y) << 16);
        o.y = (unsigned)f2bf_a(v.z) | ((unsigned)f2bf_a(v.w) << 16);
        *reinterpret_cast<uint2*>(Y + (int64_t)m * N + n) = o;
    }
}

bool attn_out_fused_ok(int M, int nseq, int hidden, int heads, int N) {
    // RASS_ATTN_FUSE=0 / RASS_GEMM_FEWROWS=0 (read per launch: the A/B) keep the attention launch + the few-rows GEMM
    const char* v = rass_env("RASS_ATTN_FUSE");
    if (v && v[0] == '0') return false;
    const char* f = rass_env("RASS_GEMM_FEWROWS");
    if (f && f[0] == '0') return false;
    return M >= 1 && M <= 32 && nseq >= 1 && nseq <= M && hidden == 16 * kHeadDim && heads == 16 && N % 16 == 0 && N >= 16;
}

bool attn_out_fused_pays(int M, int nseq) {
    // a wave walks the sequences one after the other: measured against the pair (scripts/probe_encoder_attn_fuse.py), one
    // sequence of <= 16 tokens gains 1.5-2.5 us per layer, of 23 tokens 0.2, of 32 it loses 0.5; two sequences in 16 tokens gain
    // 0.4, three or more lose
    const char* v = rass_env("RASS_ATTN_FUSE");
    if (v && v[0] == '2') return true;   // wherever it is valid (tests, A/B)
    return (nseq == 1 && M <= 24) || (nseq == 2 && M <= 16);
}

hipError_t launch_attn_out_fused(const void* qkv, const int32_t* cu_seqlens, int nseq, int M, int hidden, int heads, const void* W,
                                 const float* bias, const void* residual, void* Y, int N, hipStream_t stream) {
    if (!attn_out_fused_ok(M, nseq, hidden, heads, N) || !residual) return hipErrorInvalidValue;
    const int rows = M <= 16 ? 1 : 2;
    const int lds_bytes = 16 * (kFusedVRows * kVPitch + 16 * rows * kFusedXPitch * 2);
    static bool attr_set[2] = {false, false};
    if (!attr_set[rows - 1]) {
        hipError_t e = rows == 1 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_out_fewrows_kernel<1>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)
                                 : hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_out_fewrows_kernel<2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set[rows - 1] = true;
    }
    if (rows == 1)
        hipLaunchKernelGGL(attn_out_fewrows_kernel<1>, dim3(N / 16), dim3(1024), lds_bytes, stream, static_cast<const u16*>(qkv),
                           cu_seqlens, nseq, M, static_cast<const u16*>(W), bias, static_cast<const u16*>(residual),
                           static_cast<u16*>(Y), N);
    else
        hipLaunchKernelGGL(attn_out_fewrows_kernel<2>, dim3(N / 16), dim3(1024), lds_bytes, stream, static_cast<const u16*>(qkv),
                           cu_seqlens, nseq, M, static_cast<const u16*>(W), bias, static_cast<const u16*>(residual),
                           static_cast<u16*>(Y), N);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// attention64_kernel: the kernel for batches of LONG sequences (the ingest shape: 256 chunks x 512 tokens), on 32x32
// tiles (v_mfma_f32_32x32x16_bf16).  How it got here, with every number: profiles/r02_attention_experiments.txt.
//
// Counted on the 16x16 kernel's ISA (prices: MI355X_MICROARCH "vector-instruction ISSUE cost"): per 64 keys a wave
// spends ~80 plain VALU issues + 16 v_exp on 16 scores per lane — 33 of them the maximum (a canonicalising v_max before
// every fmaxf on an MFMA output, two ds_bpermute rounds with lgkmcnt(0) each), 6 the row sum's shuffles — and reads
// 16 KiB of K / V fragments from LDS for 16 queries; and every workgroup (one per CU: the image takes the LDS) first
// waits ~10 k cycles for its 128 KiB (138 of 469 us per launch measured with the loads removed).  Here:
//   * S^T tile = 32 keys x 32 queries: lane (q = lane & 31, h = lane >> 5) holds keys (r&3) + 8(r>>2) + 4h of the tile,
//     so a query's scores live in TWO lanes: the tile maximum is 8 v_max3_f32 in the lane + one v_permlane32_swap (no
//     LDS round trip; this file is built with -fno-honor-nans, so no canonicalisation either), and the row sum stays a
//     per-lane partial until the epilogue — both lanes of a query scale it by the same factor;
//   * the exponentiated scores are still the next MFMA's B operand as they lie (k-slot j of lane half h <-> key
//     (j&3) + 8(j>>2) + 4h of a 16-key step; the V^T operand is fetched in that key order: two ds_read_b64_tr_b16 at
//     rows +0 and +8); V rows have the K pitch (128 B) with the 64-byte halves exchanged on rows with bit 1 set: the
//     4 rows x 64 B a half-wave's transposed read touches fall on 64 distinct banks; K + V = 128 KiB at S = 512;
//   * O is exchanged between the two lanes of a query (v_permlane32_swap) so that each stores 16 contiguous bytes;
//   * eight waves, each with TWO 32-query tiles A and B (a sequence of up to 512 tokens is one pass), software-
//     pipelined half a step apart in ONE instruction stream (step = one 32-key tile):
//
//       block 1(kt):  MFMA { S_A = K Q_A^T (kt) ; O_A += V^T P_A(kt-1) }  beside  VALU { P_B(kt-1) = exp2(..) } ; max_A(kt)
//       block 2(kt):  MFMA { S_B = K Q_B^T (kt) ; O_B += V^T P_B(kt-1) }  beside  VALU { P_A(kt)   = exp2(..) } ; max_B(kt)
//
//     every block holds 8 MFMAs and an INDEPENDENT ~300 issue cycles of VALU; K and V^T fragments are read from LDS
//     once per key tile for both query tiles, a block ahead of their MFMAs; only the maximum's swap and the (out of
//     line) rescale branch sit at a block's end.  The same stream on one wave per SIMD ran 5 % slower: one in-order
//     wave issues a vector instruction every ~9 cycles where independent ones issue in 4;
//   * the reference maximum only moves when a score exceeds it by more than 2^8 (probabilities <= 256: bf16 and the
//     fp32 sums have the exponent range for it and rounding is relative), so after the first tile the rescale of O is
//     the rare branch — with an exact running maximum, 32 queries make it the usual one;
//   * persistent (workgroup b takes items b, b + grid, ...), and the K / V image is replaced IN PLACE, 128 rows (a
//     chunk = 4 key tiles) at a time, by LDS-DMA (global_load_lds, 1 KiB per wave instruction: 8 rows x 128 B, the
//     row's 16-B chunks permuted on the global side so that the lane-linear LDS image is the swizzled one).
//     Boundary b (between key tiles 4b-1 and 4b): vmcnt(0) + barrier — every wave is done reading chunk b-1 of this
//     item and every DMA issued a quarter of an item ago has landed — then the NEXT item's chunk b-1 is sent into the
//     dead rows: no staging phase, no registers held for it.  Rows past the sequence's end are copies of its last row
//     (finite; their scores are masked and their probabilities are exactly 0).
// Vector issue is what bounds it now (two waves per SIMD issue 74 % of the cycles: ~97 vector instructions per block
// beside 8 MFMAs; matrix pipe busy 30 %).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// fmaxf on MFMA outputs: with -fno-honor-nans chains fold into v_max3_f32 (scores are finite or the -inf of a masked
// key, never NaN)
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float max2f(float a, float b) { return __builtin_fmaxf(a, b); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// one LDS-DMA wave instruction: lane l's 16 bytes at gptr land at lds_piece + 16 l (lds_piece wave-uniform)
__device__ __forceinline__ void dma16(const void* gptr, const unsigned char* lds_piece) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)lds_piece);
    // M0 = the piece's LDS address (nothing else in these kernels uses M0)
    // nt: every K / V row is read once per (sequence, head); kept out of L2's way the 16-B pieces of the output stores combine a
    // little better (r03: 409.8 -> 405.1 us per launch at 256 x 512, three A/B pairs; -DRASS_ATTN_DMA_PLAIN = the A/B build)
#ifdef RASS_ATTN_DMA_PLAIN
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(gptr) : "memory");
#else
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(dst), "v"(gptr) : "memory");
#endif
}

constexpr int kA64Threads = 512;
#ifndef RASS_ATTN_CHUNK
#define RASS_ATTN_CHUNK 128   // rows of the K / V image replaced per boundary (experiment: 256 = two barriers per item instead of four)
#endif
constexpr int kChunk = RASS_ATTN_CHUNK;
constexpr int kChunkTiles = kChunk / 32;
static_assert(kChunk == 128 || kChunk == 256, "chunk = 16 or 32 DMA pieces of 8 rows, two or four per wave");
#ifdef RASS_ATTN_STAMPS  // scripts/microbench/attn_stamps.hip only: where an item's cycles go, per wave (s_memtime ticks)
// [block][wave][0 item total, 1 prologue (sync0 .. first step), 2 steps, 3 of which boundaries (wait + barrier + DMA issue),
//  4 epilogue (last products, end-of-item wait + barrier, normalise + store), 5 items]
//  then: 6 epilogue's last products, 7 its vmcnt(0) wait (next item's Q fragments), 8 its barrier
__device__ unsigned long long g_attn_stamps[1024 * 8 * 9];
#define RASS_STAMP(var) const unsigned long long var = clock64()
#else
#define RASS_STAMP(var)
#endif

// FOLD (round 3; built on VERDICT r2 #3's request, measured, NOT adopted — RASS_ATTN_VARIANT=w8f selects it): the softmax's
// scale-and-subtract moves INTO the QK^T MFMA chain.  The plain kernel spends
// one v_fma per score on exp2(s * scale - m): 16 of the ~74 vector instructions of a block, in a kernel whose time is its
// vector issue.  Here (i) Q is scaled by 1/8 * log2(e) in registers when an item's fragments arrive (once per item: a
// second bf16 rounding of Q, |rel err| <= 2^-9 per element — measured cost in profiles/r03_attention_experiments.txt), so
// the MFMA yields scores in the exp2 domain, and (ii) the chain STARTS with a fifth k-step whose A fragment is 1 in one
// k-slot and whose B fragment holds -m_ref of the lane's query there (m_ref is kept bf16-representable, so nothing is
// rounded): S' = K Q'^T - m_ref leaves the chain and the exponentials take it as it is.  The reference maximum used is
// the one known BEFORE the tile's QK^T; the tile's own maximum, now relative to it, moves it in the same rare branch as
// before (by more than 2^8), which then also shifts the tile's scores by the difference.  One MFMA more per tile on a
// matrix pipe that is 30 % busy, 16 fewer vector issues: 68 / 82 instructions per block against 87 / 98 in the ISA.
// Measured (profiles/r03_attention_experiments.txt): the SAME time (377-386 against 381-396 us per launch at 256 x 512,
// interleaved), because the launch is not paced by the instruction count of its steps alone; and 3x the error of the plain
// form on peaked scores (max |err| 8.9e-2 against 3.1e-2 at |q.k|/8 up to 40: the second rounding of Q), outside this
// file's test tolerance.  Same speed, less accuracy: the plain form stays the default.
template <bool FOLD>
__global__ __launch_bounds__(kA64Threads) void attention64_kernel(const u16* __restrict__ qkv,
                                                                  const int32_t* __restrict__ cu, int hidden, int heads,
                                                                  int s_pad, int n_items, u16* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Kl = lds;                            // [s_pad][128 B], 16-B chunk c at c ^ ((key >> 1) & 7)
    unsigned char* Vl = lds + (size_t)s_pad * 128;      // [s_pad][128 B], 16-B chunk c at c ^ (((key >> 1) & 1) << 2)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ld = 3 * hidden;
    const int h = lane >> 5, qi = lane & 31;
    const int vi = lane & 15;
    const int v_sw = (vi >> 3) & 1;
    const int v_lane_off = (4 * h + (vi >> 2)) * 128 + (2 * ((lane >> 4) & 1) + ((vi & 3) >> 1)) * 16 + (vi & 1) * 8;
    constexpr float kScale = 0.18033688011112042f;  // 1/8 * log2(e)

    const int total = cu[n_items / heads];
    if (total <= 0) return;
    auto item_rows = [&](int it, int& t0, int& S) {
        const int seq = it / heads;
        t0 = cu[seq];
        S = cu[seq + 1] - t0;
        if (S > s_pad) S = s_pad;
    };
    auto clamp_tok = [&](int tok) { return tok < 0 ? 0 : (tok < total ? tok : total - 1); };
    // chunk j (rows 128 j .. +127) of item `it`: 16 K pieces + 16 V pieces of 8 rows; wave w sends pieces w and w + 8
    auto dma_chunk = [&](int it, int j) {
        int t0, S;
        item_rows(it, t0, S);
        const u16* base = qkv + (it % heads) * kHeadDim + hidden;
#pragma unroll
        for (int pp = 0; pp < kChunk / 64; ++pp) {
            const int row0 = kChunk * j + 8 * (wave + 8 * pp);
            const int row = row0 + (lane >> 3);
            const int tok = clamp_tok(t0 + (row < S ? row : S - 1));
            const int ck = (lane & 7) ^ ((row >> 1) & 7);
            const int cv = (lane & 7) ^ (((row >> 1) & 1) << 2);
            const u16* src = base + (int64_t)tok * ld;
            // opaque to the compiler on purpose: with the builtin it puts an s_waitcnt vmcnt(0) in front of the LDS reads
            // of every step (any LDS read may alias an LDS-DMA write), i.e. each boundary's DMA is waited for at once;
            // the waits that order these writes against the reads are the boundaries' own vmcnt(0) + barrier
            dma16(src + ck * 8, Kl + row0 * 128);
            dma16(src + hidden + cv * 8, Vl + row0 * 128);
        }
    };
    // Q fragments of the wave's two tiles: B operand of S^T = K Q^T, Q[q][16 ks + 8 h .. +7]
    auto load_q = [&](int t0, int S, int head, bf16x8 (&qf)[2][4]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = wave * 64 + 32 * t + qi;
            const int tok = clamp_tok(t0 + (q < S ? q : S - 1));
            const u16* qrow = qkv + (int64_t)tok * ld + head * kHeadDim + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)  // opaque like the DMA (the compiler's counted waits would not see those)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[t][ks]) : "v"(qrow + 16 * ks) : "memory");
        }
    };

    int item = blockIdx.x;
    bool sync0 = true;  // does this item's first chunk still need a wait + barrier?
    bf16x8 qq[2][4];
    bool q_fresh = true;  // FOLD: the fragments in qq are as loaded (not yet scaled)
    // FOLD: Q' = bf16(Q * scale), in place, once per item (after the wait that retires the fragments' loads)
    auto scale_q = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 w = __builtin_bit_cast(u32x4, qq[t][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    w[i] = pack_bf16(__uint_as_float(w[i] << 16) * kScale, __uint_as_float(w[i] & 0xffff0000u) * kScale);
                qq[t][ks] = __builtin_bit_cast(bf16x8, w);
            }
    };
    // everything this wave has in flight (DMA pieces, Q fragments, stores) is done; the operands tie the uses of qq behind it
    auto wait_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(qq[0][0]), "+v"(qq[0][1]), "+v"(qq[0][2]), "+v"(qq[0][3]), "+v"(qq[1][0]), "+v"(qq[1][1]),
                       "+v"(qq[1][2]), "+v"(qq[1][3])
                     :
                     : "memory");
    };
    {
        int t0, S;
        item_rows(item, t0, S);
        load_q(t0, S, item % heads, qq);
        for (int j = 0; j < (S + kChunk - 1) / kChunk; ++j) dma_chunk(item, j);
    }
#ifdef RASS_ATTN_STAMPS
    unsigned long long st_item = 0, st_pro = 0, st_steps = 0, st_bound = 0, st_n = 0, st_e_tail = 0, st_e_wait = 0, st_e_bar = 0;
#endif
    while (true) {
        RASS_STAMP(c_item0);
        int t0, S;
        item_rows(item, t0, S);
        const int head = item % heads;
        const int n_kt = (S + 31) / 32, nc = (S + kChunk - 1) / kChunk;
        const int next = item + gridDim.x;
        const bool has_next = next < n_items;
        int nc_next = 0;
        if (has_next) {
            int nt0, nS;
            item_rows(next, nt0, nS);
            nc_next = (nS + kChunk - 1) / kChunk;
        }
        if (sync0) {
            wait_all();
            __syncthreads();
        }
        if (FOLD && q_fresh) {   // every path to here has waited for the fragments (sync0, or the previous item's end)
            scale_q();
            q_fresh = false;
        }
        // the next item's Q fragments: issued as soon as this item's last QK^T has read the old ones, a block before
        // the end-of-item wait
        auto load_q_next = [&]() {
            if (has_next) {
                int nt0, nS;
                item_rows(next, nt0, nS);
                load_q(nt0, nS, next % heads, qq);
            }
        };
        for (int j = nc; j < nc_next; ++j) dma_chunk(next, j);  // rows this item never touches
        // boundary b >= 1, between key tiles 4b-1 and 4b
        auto boundary = [&](int b) {
            RASS_STAMP(c_b0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (b - 1 < nc_next) dma_chunk(next, b - 1);
#ifdef RASS_ATTN_STAMPS
            st_bound += clock64() - c_b0;
#endif
        };
        const int q0 = wave * 64;
        if (q0 < S) {
            // one step = one 32-key tile of one query tile: 4 + 4 MFMAs beside 16 exponentials
            float m_ref[2] = {-INFINITY, -INFINITY}, l_part[2] = {0.f, 0.f};
            f32x16 O[2][2], s[2];
            bf16x8 pf[2][2];
            // FOLD: the fifth k-step's operands.  A = 1.0 in k-slot 0 of lane half 0 (every key row), B = -m_ref of the
            // lane's query in that slot: the chain starts at -m_ref.  Before the first tile m_ref stands at 0.
            bf16x8 one_a = {0, 0, 0, 0, 0, 0, 0, 0}, mneg[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
            if (FOLD) {
                one_a[0] = h == 0 ? (short)0x3F80 : (short)0;
                m_ref[0] = m_ref[1] = 0.f;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { O[t][0][r] = 0.f; O[t][1][r] = 0.f; }

            // K fragments of key tile kt (A operand rows = keys) and V^T fragments (two transposed reads per 16-key
            // step and 32-d block): the same for both query tiles, so each is read from LDS once per key tile, one
            // block ahead of its first MFMA
            bf16x8 kf[4], vf[2][2];
            auto load_k = [&](int kt) {
                const int krow = kt * 32 + qi;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int c = (2 * ks + h) ^ ((krow >> 1) & 7);
                    kf[ks] = *reinterpret_cast<const bf16x8*>(Kl + krow * 128 + c * 16);
                }
            };
            auto load_v = [&](int kt) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned char* vblk = Vl + (kt * 32 + 16 * j) * 128 + v_lane_off;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const unsigned char* va = vblk + ((dt ^ v_sw) * 64);
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) bf16x4*)(va));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) bf16x4*)(va + 8 * 128));
                        vf[j][dt][0] = lo[0]; vf[j][dt][1] = lo[1]; vf[j][dt][2] = lo[2]; vf[j][dt][3] = lo[3];
                        vf[j][dt][4] = hi[0]; vf[j][dt][5] = hi[1]; vf[j][dt][6] = hi[2]; vf[j][dt][7] = hi[3];
                    }
                }
            };
            auto qk = [&](int t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#ifndef RASS_ATTN_EXP_NO_MFMA
                if (FOLD) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one_a, mneg[t], s[t], 0, 0, 0);
#endif
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#ifdef RASS_ATTN_EXP_NO_MFMA
                    s[t][ks] += __builtin_bit_cast(float, (int)kf[ks][0] + (int)qq[t][ks][0]);
#else
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qq[t][ks], s[t], 0, 0, 0);
#endif
            };
            auto pv = [&](int t) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#ifdef RASS_ATTN_EXP_NO_MFMA
                        O[t][dt][j] += __builtin_bit_cast(float, (int)vf[j][dt][0] + (int)pf[t][j][0]);
#else
                        O[t][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[j][dt], pf[t][j], O[t][dt], 0, 0, 0);
#endif
            };
            // Tile kt's scores of query tile t just arrived: mask the ragged tail, tile maximum.  The reference maximum
            // m_ref only moves when a score exceeds it by more than 2^8 (the probabilities are then <= 256: bf16 and
            // the fp32 sums have the exponent range for it and rounding is relative), so after the first tile the
            // rescale of O is the rare branch — with an exact running maximum 32 queries make it the usual one.
            auto maxres = [&](int t, int kt, auto last_tile) {
#ifdef RASS_ATTN_EXP_NO_MAX
                if (kt == 0) m_ref[t] = 0.f;
                return;
#endif
                // only a sequence's LAST key tile can hold keys >= S; its steps are a copy of the loop body with the
                // mask in it (a wave-uniform branch here would cut the block between the MFMAs and this maximum, and the
                // maximum's dependent chain would run with nothing beside it)
                if constexpr (decltype(last_tile)::value) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= S) s[t][r] = -INFINITY;
                }
                const float a0 = max3f(s[t][0], s[t][1], s[t][2]), a1 = max3f(s[t][3], s[t][4], s[t][5]);
                const float a2 = max3f(s[t][6], s[t][7], s[t][8]), a3 = max3f(s[t][9], s[t][10], s[t][11]);
                const float a4 = max3f(s[t][12], s[t][13], s[t][14]);
                float mx = max3f(max3f(a0, a1, a2), max3f(a3, a4, s[t][15]), a0);
                {
                    const auto sw =
                        __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = max2f(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                if constexpr (FOLD) {
                    // mx is the tile maximum RELATIVE to m_ref (the chain started at -m_ref, scores are in the exp2 domain)
                    if (__builtin_expect(__any(mx > 8.f), 0)) {
                        // the new reference: the running maximum, rounded UP to a bf16 value (so that it fits the B
                        // fragment exactly; scores then stay <= 0 up to that rounding); delta is exact in fp32
                        const float cand = m_ref[t] + max2f(mx, 0.f);
                        unsigned u = __float_as_uint(cand);
                        u = (cand > 0.f ? u + 0xffffu : u) & 0xffff0000u;
                        const float m_new = __uint_as_float(u);
                        const float delta = m_new - m_ref[t];
                        const float alpha = __builtin_amdgcn_exp2f(-delta);  // 1 where the reference did not move
                        l_part[t] *= alpha;
                        O[t][0] *= alpha;
                        O[t][1] *= alpha;
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[t][r] -= delta;   // this tile's scores were taken against the old one
                        m_ref[t] = m_new;
                        mneg[t][0] = h == 0 ? (short)((__float_as_uint(-m_new)) >> 16) : (short)0;
                    }
                    return;
                }
                mx *= kScale;  // finite in tile 0 (key 0 is always valid); -inf for a fully masked query never happens
                if (__builtin_expect(__any(mx > m_ref[t] + 8.f), 0)) {  // out of line: the usual path falls through
                    const float m_new = max2f(m_ref[t], mx);
                    const float alpha = __builtin_amdgcn_exp2f(m_ref[t] - m_new);  // 1 where m_ref did not move
                    l_part[t] *= alpha;
                    O[t][0] *= alpha;
                    O[t][1] *= alpha;
                    m_ref[t] = m_new;
                }
            };
            // P = exp2(S * scale - m_ref) of query tile t -> bf16 operand fragments, row-sum partials
            auto exps = [&](int t) {
                f32x16 pe;
#pragma unroll
#ifdef RASS_ATTN_EXP_NO_EXP
                for (int r = 0; r < 16; ++r) pe[r] = __builtin_fmaf(s[t][r], kScale, -m_ref[t]);
#else
                for (int r = 0; r < 16; ++r)
                    pe[r] = FOLD ? __builtin_amdgcn_exp2f(s[t][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], kScale, -m_ref[t]));
#endif
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 pw;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pw[i] = pack_bf16(pe[8 * j + 2 * i], pe[8 * j + 2 * i + 1]);
                    pf[t][j] = __builtin_bit_cast(bf16x8, pw);
                }
                l_part[t] += (((pe[0] + pe[1]) + (pe[2] + pe[3])) + ((pe[4] + pe[5]) + (pe[6] + pe[7]))) +
                             (((pe[8] + pe[9]) + (pe[10] + pe[11])) + ((pe[12] + pe[13]) + (pe[14] + pe[15])));
                // pin the exponentials HERE, in the block of the other tile's MFMAs: left alone, the optimiser sinks them
                // past the branches of maxres() to their first use, in front of the MFMAs that wait for them
                asm volatile("" : "+v"(pf[t][0]), "+v"(pf[t][1]), "+v"(l_part[t]));
            };

            // tile A leads, tile B follows half a step behind; QK^T goes first in a block (its scores feed the block's
            // closing maximum), the fragments of the next block's MFMAs are fetched before that maximum's branches
            constexpr std::true_type kMasked{};
            constexpr std::false_type kFull{};
            load_k(0);
            qk(0);
            maxres(0, 0, kMasked);
            qk(1);
            if (n_kt == 1) load_q_next();
            load_k(n_kt > 1 ? 1 : 0);
            load_v(0);
            exps(0);
            maxres(1, 0, kMasked);
            auto step = [&](int kt, auto last_tile) {
#ifdef RASS_ATTN_EXP_PRIO
                // experiment: the two waves of a SIMD take turns at the higher issue priority, one step each
                if (((kt ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
                qk(0);            // S_A(kt)
                pv(0);            // O_A += V^T P_A(kt-1)
                exps(1);          // P_B(kt-1)
                maxres(0, kt, last_tile);
                qk(1);            // S_B(kt)
                if constexpr (decltype(last_tile)::value) load_q_next();
                pv(1);            // O_B += V^T P_B(kt-1)
                load_k(kt + 1 < n_kt ? kt + 1 : kt);
                load_v(kt);
                exps(0);          // P_A(kt)
                maxres(1, kt, last_tile);
            };
            RASS_STAMP(c_steps0);
            for (int kt = 1; kt < n_kt - 1; ++kt) {
                if ((kt % kChunkTiles) == 0) boundary(kt / kChunkTiles);
                step(kt, kFull);
            }
            if (n_kt > 1) {
                if (((n_kt - 1) % kChunkTiles) == 0) boundary((n_kt - 1) / kChunkTiles);
                step(n_kt - 1, kMasked);
            }
            RASS_STAMP(c_steps1);
#ifdef RASS_ATTN_STAMPS
            st_pro += c_steps0 - c_item0;
            st_steps += c_steps1 - c_steps0;
#endif
            pv(0);
            exps(1);
            pv(1);
            RASS_STAMP(c_e1);
            // end of the item: everyone is done with the last chunk (and the next item's Q fragments have arrived)
            wait_all();
            RASS_STAMP(c_e2);
            __syncthreads();
            RASS_STAMP(c_e3);
#ifdef RASS_ATTN_STAMPS
            st_e_tail += c_e1 - c_steps1;
            st_e_wait += c_e2 - c_e1;
            st_e_bar += c_e3 - c_e2;
#endif
#ifndef RASS_ATTN_EXP_NO_ENDDMA
            if (nc - 1 < nc_next) dma_chunk(next, nc - 1);
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q = q0 + 32 * t + qi;
                // the query's sum = this lane's partial + its partner's; O^T[d = 32 dt + (r&3) + 8(r>>2) + 4h][q]
                float l_run;
                {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_part[t]),
                                                                     __float_as_uint(l_part[t]), false, false);
                    l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                }
                const float inv_l = 1.f / l_run;
                u16* dst = ctx + (int64_t)(t0 + (q < S ? q : 0)) * hidden + head * kHeadDim + 8 * h;
                // (Round 3 also staged a tile through 4 KiB of LDS per wave and stored whole 128-B rows, 8 lines per wave
                // instruction instead of 64 partial ones: same time, 404 against 398-401 us — the stores cost ~25 us of a launch
                // whatever their shape, profiles/r03_attention_experiments.txt — so the registers are stored as they lie.
                // Nontemporal stores, which win 3 % in the GEMM epilogue's whole-line stores, LOSE here: 430-512 us per launch
                // against 396-405 — these are 16-B pieces of a line, and without L2 to combine them every piece is a partial write.)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    unsigned w[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        w[i] = pack_bf16(O[t][dt][2 * i] * inv_l, O[t][dt][2 * i + 1] * inv_l);
                    // (w0,w1) = d 4h+0..3, (w2,w3) = d 8+4h.., (w4,w5) = d 16+4h.., (w6,w7) = d 24+4h..  ->  the low
                    // lane keeps d 0-7 and 16-23 of the 32-block, the high lane d 8-15 and 24-31
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(w[4 * p + 0], w[4 * p + 2], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(w[4 * p + 1], w[4 * p + 3], false, false);
#ifdef RASS_ATTN_EXP_NO_STORE   // timing experiment: everything but the global stores (one lane keeps the data alive)
                        if (q < S && s0[0] == 0x12345678u && lane == 63)
#else
                        if (q < S)
#endif
                            *reinterpret_cast<u32x4*>(dst + 32 * dt + 16 * p) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    }
                }
            }
        } else {  // a wave without queries keeps the rhythm and moves its share of the rows
            for (int b = 1; b < nc; ++b) boundary(b);
            load_q_next();
            wait_all();
            __syncthreads();
            if (nc - 1 < nc_next) dma_chunk(next, nc - 1);
        }
#ifdef RASS_ATTN_STAMPS
        {
            const unsigned long long c_end = clock64();
            st_item += c_end - c_item0;
            st_n += 1;
        }
#endif
        if (!has_next) break;
        sync0 = nc < 3;  // chunk 1 (nc == 2) or chunk 0 (nc == 1) of the next item was only just sent
        q_fresh = true;  // load_q_next() replaced the fragments (both branches above end in wait_all())
        item = next;
    }
#ifdef RASS_ATTN_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = g_attn_stamps + ((size_t)blockIdx.x * 8 + wave) * 9;
        o[0] = st_item; o[1] = st_pro; o[2] = st_steps; o[3] = st_bound; o[4] = st_item - st_pro - st_steps; o[5] = st_n;
        o[6] = st_e_tail; o[7] = st_e_wait; o[8] = st_e_bar;
    }
#endif
}

static int attn_cus() {  // one persistent workgroup per CU (the K / V image takes most of a CU's LDS)
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
               prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount
                  : 256;
    }
    return cus;
}

static const char* attn_variant() {
    const char* v = rass_env("RASS_ATTN_VARIANT");
    return v ? v : "";
}

hipError_t launch_attention(const void* qkv, const int32_t* cu_seqlens, int nseq, int total_tokens, int max_seqlen,
                            int hidden, int heads, void* ctx, hipStream_t stream) {
    if (hidden != heads * kHeadDim || max_seqlen < 1 || max_seqlen > 512 || total_tokens < 0) return hipErrorInvalidValue;
    if (nseq <= 0 || total_tokens == 0) return hipSuccess;
    const int s_pad = (max_seqlen + 63) / 64 * 64;
    // Mostly-long sequences (mean >= 320 tokens): the 32x32-tile persistent kernel (256 x 512: 396-405 us against 480;
    // lengths ~ N(420, 60): 341 against 387; N(340, 80): 286 against 298).  Short or very ragged batches keep the 16x16
    // kernel: 16 queries per wave fill a CU from 256 tokens on, and its one workgroup per item is balanced by the
    // dispatcher (equal at a mean of 256-300 tokens, 9 % ahead at 192, 50 % at 128, 3 % on one 16-token query).
    // RASS_ATTN_VARIANT=w8 / w16 forces one (tests, A/B).
    const char* variant = attn_variant();
    const bool long_rows = (long long)total_tokens >= 320LL * nseq;
    const bool w8_plain = strcmp(variant, "w8") == 0;     // scale-and-subtract as one v_fma per score (the default form)
#ifdef RASS_ATTN_EXPERIMENTS
    const bool w8_fold = strcmp(variant, "w8f") == 0;     // both folded into the QK^T chain: measured, not adopted (see the kernel)
#else
    const bool w8_fold = false;   // the FOLD experiment (3x the error on peaked scores, no faster) is not instantiated in the
                                  // product library: build with -DRASS_ATTN_EXPERIMENTS (make EXTRA=-DRASS_ATTN_EXPERIMENTS)
#endif
    if (w8_plain || w8_fold || (long_rows && strcmp(variant, "w16") != 0)) {
        const int s_pad128 = (max_seqlen + kChunk - 1) / kChunk * kChunk;  // whole chunks are sent
        const size_t lds_bytes = (size_t)s_pad128 * 256;      // 128 KiB at S = 512
        static size_t attr64_bytes = 0;
        if (lds_bytes > attr64_bytes) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention64_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
#ifdef RASS_ATTN_EXPERIMENTS
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention64_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
#endif
            if (e != hipSuccess) return e;
            attr64_bytes = lds_bytes;
        }
        const int n_items = nseq * heads;
        const dim3 grid(n_items < attn_cus() ? n_items : attn_cus());
        if (!w8_fold)
            hipLaunchKernelGGL(attention64_kernel<false>, grid, dim3(kA64Threads), lds_bytes, stream,
                               static_cast<const u16*>(qkv), cu_seqlens, hidden, heads, s_pad128, n_items,
                               static_cast<u16*>(ctx));
#ifdef RASS_ATTN_EXPERIMENTS
        else
            hipLaunchKernelGGL(attention64_kernel<true>, grid, dim3(kA64Threads), lds_bytes, stream,
                               static_cast<const u16*>(qkv), cu_seqlens, hidden, heads, s_pad128, n_items,
                               static_cast<u16*>(ctx));
#endif
        return hipGetLastError();
    }
    const size_t lds_bytes = (size_t)s_pad * 128 + (size_t)s_pad * kVPitch;  // 144 KiB at S = 512
    static size_t attr_bytes = 0;
    if (lds_bytes > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL(attention_kernel, dim3(nseq * heads), dim3(kAttnThreads), lds_bytes, stream,
                       static_cast<const u16*>(qkv), cu_seqlens, hidden, heads, s_pad, static_cast<u16*>(ctx));
    return hipGetLastError();
}

}  // namespace rass
