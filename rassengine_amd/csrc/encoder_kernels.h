// encoder_kernels.h — internal launcher interface of the sentence-encoder kernels (K4-K8 of
// SURVEY §8a).  Activations are bf16 [tokens][features] row-major, tokens of all sequences
// packed back to back (varlen, cu_seqlens), padded to a multiple of 128 rows.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rass {

// The A/B switches of the encoder kernels (RASS_GEMM_*, RASS_ATTN_*, RASS_LN_*) are environment variables read "per launch";
// a one-query forward is 120 launches of ~5 us each with up to a dozen such reads per launch, and getenv walks the whole
// environment: the host could not enqueue faster than the GPU ran (round 4: 4.5 us per launch on the host).  rass_env(name)
// = getenv(name), remembered per thread until rass_env_new_scope() — called at every C-ABI entry of the encoder, so a switch is
// still read once per call.  `name` must be a string literal (the table is keyed by its address).
const char* rass_env(const char* name);
void rass_env_new_scope();

// K5: Y = epilogue(X[M,K] * W[N,K]^T + bias); epilogue 0 bias, 1 bias+residual, 2 bias+GELU(erf).
// M_pad (multiple of 128) rows of X / Y / residual must be allocated; N % 128 == 0, K % 64 == 0.
// With a scratch (`splitk_ws`, fp32) a GEMM over few rows (M_pad <= 256) is split over K so that enough workgroups
// stream the weights (query-time embedding); the result is deterministic (slices summed in fixed order).
hipError_t launch_gemm_bf16(const void* X, const void* W, const float* bias, const void* residual, void* Y, int M,
                            int M_pad, int N, int K, int epilogue, hipStream_t stream, float* splitk_ws = nullptr,
                            size_t splitk_ws_bytes = 0);

// out = LayerNorm(X W^T + bias + residual): launch_gemm_bf16(epilogue 1) into `y` followed by launch_layernorm, or — for
// few rows, with a split-K scratch — the split-K GEMM and ONE kernel that reduces the slices, adds bias and residual,
// rounds to bf16 where `y` would have been stored, and normalises (same bits, one launch less; `y` is then not written).
hipError_t launch_gemm_bf16_residual_layernorm(const void* X, const void* W, const float* bias, const void* residual,
                                               void* y, const float* gamma, const float* beta, float eps, void* out,
                                               int M, int M_pad, int N, int K, hipStream_t stream, float* splitk_ws,
                                               size_t splitk_ws_bytes);
// A query's few rows (M <= 16, K = 1024): Y = epilogue(LayerNorm(Yin) W^T + bias), epilogue 0 bias / 2 bias + GELU, with
// the LayerNorm recomputed by every workgroup into its operand tile (one launch less than layernorm + GEMM; a launch
// costs ~4 us whatever it does) and stored once to x_out (the bits launch_layernorm would write).
bool gemm_bf16_ln_input_ok(int M, int N, int K);
hipError_t launch_gemm_bf16_ln_input(const void* Yin, const float* gamma, const float* beta, float eps, void* x_out,
                                     const void* W, const float* bias, void* Y, int M, int N, int K, int epilogue,
                                     hipStream_t stream);
hipError_t launch_splitk_residual_layernorm(const float* partial, int S, int rows, int rows_pad, int hidden,
                                            const float* bias, const void* residual, const float* gamma,
                                            const float* beta, float eps, void* out, hipStream_t stream);

// ---- LayerNorm folded into the GEMMs around it (big batches on the persistent kernel; encoder_gemm.hip, LnFold) ----
// gemm_bf16_fold_ok: all four GEMMs of a layer run on the persistent 256^2 kernel at this batch (and RASS_ENCODER_LN_FOLD != 0).
// launch_gemm_bf16_fold, epilogue 3: Y = X W^T + bias + LN(residual_raw) with LN rebuilt from (mr, gamma, beta) per element;
//   Y is the RAW sum (bf16); `stats` [M][N/128][2] receives the partial (sum, sum of squares) of the stored values.
// epilogue 4 / 5: Y = rstd * (X W'^T - mean * colsum) + bias'  [ + GELU ], X raw rows with (mean, rstd) in `mr`, W' / colsum /
//   bias' from launch_fold_gamma.  launch_ln_stats_finalize: stats -> mr [M][2] (mean, rstd), fixed summation order.
bool gemm_bf16_fold_ok(int M, int M_pad, int hidden, int intermediate);
hipError_t launch_gemm_bf16_fold(const void* X, const void* W, const float* bias, const void* residual_raw, void* Y, int M,
                                 int M_pad, int N, int K, int epilogue, const float* mr, const float* gamma, const float* beta,
                                 float* stats, const float* colsum, hipStream_t stream);
hipError_t launch_ln_stats_finalize(const float* stats, int rows, int n, float eps, float* mr, hipStream_t stream);
hipError_t launch_fold_gamma(const void* W, const float* gamma, const float* beta, const float* bias, int N, int K, void* W2,
                             float* colsum, float* bias2, hipStream_t stream);

// K4: x[t] = LayerNorm(word[ids[t]] + pos[position of t in its sequence] + type[0]) -> bf16
hipError_t launch_embed_layernorm(const int32_t* ids, const int32_t* cu_seqlens, int nseq, int total_tokens,
                                  const void* word_emb, const void* pos_emb, const void* type_emb,
                                  const float* gamma, const float* beta, float eps, int hidden, int vocab,
                                  int max_pos, void* out, hipStream_t stream);

// K7: out[t] = LayerNorm(in[t]) (fp32 statistics), bf16 in/out, rows of `hidden`
hipError_t launch_layernorm(const void* in, const float* gamma, const float* beta, float eps, int rows, int hidden,
                            void* out, hipStream_t stream);

// K6: varlen self-attention, heads of 64, S <= 512 per sequence, softmax scale 1/8.
// qkv: [total_tokens][3*hidden] (q | k | v), ctx: [total_tokens][hidden]; total_tokens (= cu_seqlens[nseq], known to the
// host) also picks the kernel: mostly-long sequences take the 32x32-tile persistent kernel
hipError_t launch_attention(const void* qkv, const int32_t* cu_seqlens, int nseq, int total_tokens, int max_seqlen,
                            int hidden, int heads, void* ctx, hipStream_t stream);

// K6 + K5 for a query's few rows (all sequences together <= 32 tokens, hidden 1024, 16 heads): Y = attention(qkv) W^T + bias +
// residual in ONE launch — every workgroup of the few-rows GEMM recomputes the attention (context rows with attention_kernel's
// bits) instead of reading it from a launch of its own.  RASS_ATTN_FUSE=0: never ok.
bool attn_out_fused_ok(int M, int nseq, int hidden, int heads, int N);
bool attn_out_fused_pays(int M, int nseq);   // the encoder's rule: where the one launch is faster than the pair (RASS_ATTN_FUSE=2: wherever valid)
hipError_t launch_attn_out_fused(const void* qkv, const int32_t* cu_seqlens, int nseq, int M, int hidden, int heads, const void* W,
                                 const float* bias, const void* residual, void* Y, int N, hipStream_t stream);

// K8: pooled[s] = cls (first token) or mean over the sequence's tokens of x, fp32 [nseq][hidden];
// optional L2 normalise with the reference's formula (app/main.py:1249-1251)
hipError_t launch_pool(const void* x, const int32_t* cu_seqlens, int nseq, int hidden, int mode_mean, int normalize,
                       float* out, hipStream_t stream);

// The wave-wide sum of the row kernels: v + (lane ^ 32) + (lane ^ 16) + ... + (lane ^ 1), the butterfly `v += __shfl_xor(v, off)`
// for off = 32 .. 1 with the SAME partners in the same order (an IEEE add is commutative: the same bits in every lane), but
// without the LDS crossbar: __shfl_xor is a ds_bpermute (~100 cycles of latency each, twelve of them in a row per LayerNorm:
// most of what a few-row LayerNorm lasts); gfx950's v_permlane32_swap / v_permlane16_swap and DPP row operations take 4-8.
#ifdef __HIPCC__
__device__ __forceinline__ float wave_xor_partner_dpp(float x, int lane, int off) {
    const int v = __float_as_int(x);
    int o;
    if (off == 32) {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        o = (int)((lane & 32) ? sw[0] : sw[1]);
    } else if (off == 16) {
        const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        o = (int)((lane & 16) ? sw[0] : sw[1]);
    } else if (off == 8) {   // i ^ 8 = row_mirror(half_mirror(i))
        o = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true), 0x140, 0xf, 0xf, true);
    } else if (off == 4) {   // i ^ 4 = half_mirror(quad_reverse(i))
        o = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x1B, 0xf, 0xf, true), 0x141, 0xf, 0xf, true);
    } else if (off == 2) {
        o = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    } else {
        o = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    }
    return __int_as_float(o);
}
__device__ __forceinline__ float wave_sum_butterfly(float v) {
    const int lane = (int)(threadIdx.x & 63);
    v += wave_xor_partner_dpp(v, lane, 32);
    v += wave_xor_partner_dpp(v, lane, 16);
    v += wave_xor_partner_dpp(v, lane, 8);
    v += wave_xor_partner_dpp(v, lane, 4);
    v += wave_xor_partner_dpp(v, lane, 2);
    v += wave_xor_partner_dpp(v, lane, 1);
    return v;
}
#endif

// fp32 -> bf16 (weights upload)
hipError_t launch_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t stream);

}  // namespace rass
