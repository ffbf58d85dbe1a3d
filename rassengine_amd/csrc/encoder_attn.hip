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

hipError_t launch_attention(const void* qkv, const int32_t* cu_seqlens, int nseq, int max_seqlen, int hidden,
                            int heads, void* ctx, hipStream_t stream) {
    if (hidden != heads * kHeadDim || max_seqlen < 1 || max_seqlen > 512) return hipErrorInvalidValue;
    if (nseq <= 0) return hipSuccess;
    const int s_pad = (max_seqlen + 63) / 64 * 64;
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
