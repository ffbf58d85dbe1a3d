// encoder_kernels.h — internal launcher interface of the sentence-encoder kernels (K4-K8 of
// SURVEY §8a).  Activations are bf16 [tokens][features] row-major, tokens of all sequences
// packed back to back (varlen, cu_seqlens), padded to a multiple of 128 rows.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rass {

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

// K8: pooled[s] = cls (first token) or mean over the sequence's tokens of x, fp32 [nseq][hidden];
// optional L2 normalise with the reference's formula (app/main.py:1249-1251)
hipError_t launch_pool(const void* x, const int32_t* cu_seqlens, int nseq, int hidden, int mode_mean, int normalize,
                       float* out, hipStream_t stream);

// fp32 -> bf16 (weights upload)
hipError_t launch_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t stream);

}  // namespace rass
