// encoder_misc.hip — the HBM-bound row kernels of the sentence encoder:
//   K4 embedding gather (word + position + token-type) + LayerNorm
//   K7 LayerNorm (post-LN BERT: after each residual sum; eps 1e-12)
//   K8 pooling (cls | mean over the sequence's tokens) + the reference's L2 normalise
// One wave per row, 16 B (8 bf16) per lane per step, fp32 statistics, two-pass variance on
// register-resident values, wave butterfly reductions, no LDS.  hidden % 8 == 0, <= 2048.

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>

#include "encoder_kernels.h"

namespace rass {

typedef unsigned short u16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowThreadsE = 256;  // 4 rows per block
constexpr int kMaxSteps = 4;       // hidden <= 4 * 512

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<u16*>(&h);
}
__device__ __forceinline__ float wave_sum_e(float v) {
    // = `for (off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64)`, bit for bit (encoder_kernels.h)
    return wave_sum_butterfly(v);
}

struct Vals8 {
    float v[8];
};

__device__ __forceinline__ Vals8 load8_bf16(const u16* p) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    Vals8 o;
    o.v[0] = bf2f((u16)(r.x & 0xffff));
    o.v[1] = bf2f((u16)(r.x >> 16));
    o.v[2] = bf2f((u16)(r.y & 0xffff));
    o.v[3] = bf2f((u16)(r.y >> 16));
    o.v[4] = bf2f((u16)(r.z & 0xffff));
    o.v[5] = bf2f((u16)(r.z >> 16));
    o.v[6] = bf2f((u16)(r.w & 0xffff));
    o.v[7] = bf2f((u16)(r.w >> 16));
    return o;
}

__device__ __forceinline__ void store8_bf16(u16* p, const Vals8& x) {
    uint4 r;
    r.x = (unsigned)f2bf(x.v[0]) | ((unsigned)f2bf(x.v[1]) << 16);
    r.y = (unsigned)f2bf(x.v[2]) | ((unsigned)f2bf(x.v[3]) << 16);
    r.z = (unsigned)f2bf(x.v[4]) | ((unsigned)f2bf(x.v[5]) << 16);
    r.w = (unsigned)f2bf(x.v[6]) | ((unsigned)f2bf(x.v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = r;
}

// LayerNorm of a register-resident row (x[s] holds columns lane*8 + 512*s .. +7), write bf16.
__device__ __forceinline__ void layernorm_store(Vals8 (&x)[kMaxSteps], int hidden, int lane,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                float eps, u16* __restrict__ dst) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < kMaxSteps; ++s)
        if (lane * 8 + 512 * s < hidden)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += x[s].v[e];
    const float mean = wave_sum_e(sum) / (float)hidden;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < kMaxSteps; ++s)
        if (lane * 8 + 512 * s < hidden)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = x[s].v[e] - mean;
                sq = fmaf(d, d, sq);
            }
    const float rstd = rsqrtf(wave_sum_e(sq) / (float)hidden + eps);
#pragma unroll
    for (int s = 0; s < kMaxSteps; ++s) {
        const int c = lane * 8 + 512 * s;
        if (c < hidden) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
            Vals8 o;
            o.v[0] = (x[s].v[0] - mean) * rstd * g0.x + b0.x;
            o.v[1] = (x[s].v[1] - mean) * rstd * g0.y + b0.y;
            o.v[2] = (x[s].v[2] - mean) * rstd * g0.z + b0.z;
            o.v[3] = (x[s].v[3] - mean) * rstd * g0.w + b0.w;
            o.v[4] = (x[s].v[4] - mean) * rstd * g1.x + b1.x;
            o.v[5] = (x[s].v[5] - mean) * rstd * g1.y + b1.y;
            o.v[6] = (x[s].v[6] - mean) * rstd * g1.z + b1.z;
            o.v[7] = (x[s].v[7] - mean) * rstd * g1.w + b1.w;
            store8_bf16(dst + c, o);
        }
    }
}

__global__ __launch_bounds__(kRowThreadsE) void embed_layernorm_kernel(
    const int32_t* __restrict__ ids, const int32_t* __restrict__ cu, int nseq, int total, const u16* __restrict__ word,
    const u16* __restrict__ pos, const u16* __restrict__ type0, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, int hidden, int vocab, int max_pos, u16* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int t = blockIdx.x * 4 + wave; t < total; t += gridDim.x * 4) {
        // sequence of token t: largest s with cu[s] <= t (wave-uniform binary search)
        int lo = 0, hi = nseq;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cu[mid] <= t) lo = mid; else hi = mid;
        }
        int p = t - cu[lo];
        p = p < max_pos ? p : max_pos - 1;
        int id = ids[t];
        id = (id < 0 || id >= vocab) ? 0 : id;
        Vals8 x[kMaxSteps];
#pragma unroll
        for (int s = 0; s < kMaxSteps; ++s) {
            const int c = lane * 8 + 512 * s;
            if (c < hidden) {
                const Vals8 w = load8_bf16(word + (int64_t)id * hidden + c);
                const Vals8 pe = load8_bf16(pos + (int64_t)p * hidden + c);
                const Vals8 te = load8_bf16(type0 + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[s].v[e] = (w.v[e] + te.v[e]) + pe.v[e];
            }
        }
        layernorm_store(x, hidden, lane, gamma, beta, eps, out + (int64_t)t * hidden);
    }
}

hipError_t launch_embed_layernorm(const int32_t* ids, const int32_t* cu_seqlens, int nseq, int total_tokens,
                                  const void* word_emb, const void* pos_emb, const void* type_emb,
                                  const float* gamma, const float* beta, float eps, int hidden, int vocab,
                                  int max_pos, void* out, hipStream_t stream) {
    if (hidden % 8 != 0 || hidden > 512 * kMaxSteps || nseq < 1) return hipErrorInvalidValue;
    if (total_tokens <= 0) return hipSuccess;
    int blocks = (total_tokens + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(embed_layernorm_kernel, dim3(blocks), dim3(kRowThreadsE), 0, stream, ids, cu_seqlens, nseq,
                       total_tokens, static_cast<const u16*>(word_emb), static_cast<const u16*>(pos_emb),
                       static_cast<const u16*>(type_emb), gamma, beta, eps, hidden, vocab, max_pos,
                       static_cast<u16*>(out));
    return hipGetLastError();
}

// The streaming form for whole batches (2 x 268 MB per call at 256 x 512 tokens, twice per layer = 6 % of a forward):
// a wave walks rows r, r + stride, ...; gamma / beta of its 8 * STEPS columns are loaded ONCE (the per-row form fetched
// 128 B of them per lane for every 32 B of data) and the next row's data is on its way while this one is reduced.
// Arithmetic and its order are layernorm_store's, bit for bit.
template <int STEPS>
__global__ __launch_bounds__(kRowThreadsE) void layernorm_kernel(const u16* __restrict__ in,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, int rows,
                                                                 int hidden, u16* __restrict__ out, int nt_mode) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto ld = [&](const u16* p) -> uint4 {
        if (nt_mode & 1) {
            const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
            return uint4{v[0], v[1], v[2], v[3]};
        }
        return *reinterpret_cast<const uint4*>(p);
    };
    f32x4 g[STEPS][2], b[STEPS][2];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = lane * 8 + 512 * s;
        const int cc = c < hidden ? c : 0;
        g[s][0] = *reinterpret_cast<const f32x4*>(gamma + cc);
        g[s][1] = *reinterpret_cast<const f32x4*>(gamma + cc + 4);
        b[s][0] = *reinterpret_cast<const f32x4*>(beta + cc);
        b[s][1] = *reinterpret_cast<const f32x4*>(beta + cc + 4);
    }
    const int stride = gridDim.x * 4;
    int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    uint4 cur[STEPS], nxt[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = lane * 8 + 512 * s;
        cur[s] = ld(in + (int64_t)r * hidden + (c < hidden ? c : 0));
    }
    for (; r < rows; r += stride) {
        const int rn = r + stride < rows ? r + stride : r;  // no branch around the loads: the waits stay counted
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = lane * 8 + 512 * s;
            nxt[s] = ld(in + (int64_t)rn * hidden + (c < hidden ? c : 0));
        }
        Vals8 x[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            x[s].v[0] = bf2f((u16)(cur[s].x & 0xffff)); x[s].v[1] = bf2f((u16)(cur[s].x >> 16));
            x[s].v[2] = bf2f((u16)(cur[s].y & 0xffff)); x[s].v[3] = bf2f((u16)(cur[s].y >> 16));
            x[s].v[4] = bf2f((u16)(cur[s].z & 0xffff)); x[s].v[5] = bf2f((u16)(cur[s].z >> 16));
            x[s].v[6] = bf2f((u16)(cur[s].w & 0xffff)); x[s].v[7] = bf2f((u16)(cur[s].w >> 16));
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            if (lane * 8 + 512 * s < hidden)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += x[s].v[e];
        const float mean = wave_sum_e(sum) / (float)hidden;
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            if (lane * 8 + 512 * s < hidden)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = x[s].v[e] - mean;
                    sq = fmaf(d, d, sq);
                }
        const float rstd = rsqrtf(wave_sum_e(sq) / (float)hidden + eps);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = lane * 8 + 512 * s;
            if (c < hidden) {
                Vals8 o;
                o.v[0] = (x[s].v[0] - mean) * rstd * g[s][0].x + b[s][0].x;
                o.v[1] = (x[s].v[1] - mean) * rstd * g[s][0].y + b[s][0].y;
                o.v[2] = (x[s].v[2] - mean) * rstd * g[s][0].z + b[s][0].z;
                o.v[3] = (x[s].v[3] - mean) * rstd * g[s][0].w + b[s][0].w;
                o.v[4] = (x[s].v[4] - mean) * rstd * g[s][1].x + b[s][1].x;
                o.v[5] = (x[s].v[5] - mean) * rstd * g[s][1].y + b[s][1].y;
                o.v[6] = (x[s].v[6] - mean) * rstd * g[s][1].z + b[s][1].z;
                o.v[7] = (x[s].v[7] - mean) * rstd * g[s][1].w + b[s][1].w;
                if (nt_mode & 2) {
                    u32x4_t w;
                    w[0] = (unsigned)f2bf(o.v[0]) | ((unsigned)f2bf(o.v[1]) << 16);
                    w[1] = (unsigned)f2bf(o.v[2]) | ((unsigned)f2bf(o.v[3]) << 16);
                    w[2] = (unsigned)f2bf(o.v[4]) | ((unsigned)f2bf(o.v[5]) << 16);
                    w[3] = (unsigned)f2bf(o.v[6]) | ((unsigned)f2bf(o.v[7]) << 16);
                    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(out + (int64_t)r * hidden + c));
                } else {
                    store8_bf16(out + (int64_t)r * hidden + c, o);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) cur[s] = nxt[s];
    }
}

// Few rows (query-time embedding): the reduction of a split-K GEMM's fp32 partial tiles, bias, residual AND the
// LayerNorm that follows in one pass — out[r] = LayerNorm(bf16(sum_s partial[s][r] + bias + residual[r])).  The sum is
// rounded to bf16 exactly where the unfused path (splitk_epilogue_kernel<1> then layernorm_kernel) stores it, so both
// give the same bits; fused, the residual GEMMs of a layer cost one launch less each (every kernel of a one-query
// forward sits at the ~5 us dependent-launch floor).
template <int S>
__global__ __launch_bounds__(kRowThreadsE) void splitk_residual_layernorm_kernel(
    const float* __restrict__ partial, int rows, int rows_pad, int hidden, const float* __restrict__ bias,
    const u16* residual, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, u16* out) {
    // `residual` and `out` may be the same buffer (the encoder normalises in place): a wave reads its whole row
    // before it writes it, and rows are independent
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        Vals8 x[kMaxSteps];
#pragma unroll
        for (int s = 0; s < kMaxSteps; ++s) {
            const int c = lane * 8 + 512 * s;
            if (c < hidden) {
                // all S slices' loads go out before the first add (a rolled loop waited for each in turn: with 16
                // waves in the whole launch nothing else hides that latency); summed in ascending order, as
                // splitk_epilogue_kernel does
                f32x4 p0[S], p1[S];
#pragma unroll
                for (int sl = 0; sl < S; ++sl) {
                    const float* q = partial + ((int64_t)sl * rows_pad + r) * hidden + c;
                    p0[sl] = *reinterpret_cast<const f32x4*>(q);
                    p1[sl] = *reinterpret_cast<const f32x4*>(q + 4);
                }
                f32x4 v0 = p0[0], v1 = p1[0];
#pragma unroll
                for (int sl = 1; sl < S; ++sl) {
                    v0 += p0[sl];
                    v1 += p1[sl];
                }
                v0 += *reinterpret_cast<const f32x4*>(bias + c);
                v1 += *reinterpret_cast<const f32x4*>(bias + c + 4);
                const Vals8 res = load8_bf16(residual + (int64_t)r * hidden + c);
                x[s].v[0] = bf2f(f2bf(v0.x + res.v[0]));
                x[s].v[1] = bf2f(f2bf(v0.y + res.v[1]));
                x[s].v[2] = bf2f(f2bf(v0.z + res.v[2]));
                x[s].v[3] = bf2f(f2bf(v0.w + res.v[3]));
                x[s].v[4] = bf2f(f2bf(v1.x + res.v[4]));
                x[s].v[5] = bf2f(f2bf(v1.y + res.v[5]));
                x[s].v[6] = bf2f(f2bf(v1.z + res.v[6]));
                x[s].v[7] = bf2f(f2bf(v1.w + res.v[7]));
            }
        }
        layernorm_store(x, hidden, lane, gamma, beta, eps, out + (int64_t)r * hidden);
    }
}

// The same for hidden = 512 * STEPS exactly (1 024: BERT-large), as the one-query forward runs it 24 times: no lane-dependent
// branch around a step, so EVERY load of the row — the S slices, bias, residual, gamma, beta: 30 x 16 B per lane at S = 4 — is
// in flight before the first add (the general form above walks the steps one branch at a time: two dependent round trips for
// the data and a third for gamma / beta, in a kernel that lasts 4 us).  Same arithmetic in the same order: the same bits.
// 16 bytes, ISSUED here and not waited for: the compiler places its own loads where its register-pressure heuristics like
// (for this kernel: ten in flight, then load / wait / add in turn — several dependent round trips), and neither sched_barrier nor
// source order binds the instruction selector's placement of loads and pure arithmetic.  Inline asm is kept in program order;
// ld16_wait_all() is the one wait, ld16_pin() makes every later use depend on it.
__device__ __forceinline__ f32x4 ld16_issue(const void* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void ld16_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ld16_pin(f32x4& v) { asm volatile("" : "+v"(v)); }

template <int S, int STEPS>
__global__ __launch_bounds__(kRowThreadsE) void splitk_residual_layernorm_exact_kernel(
    const float* __restrict__ partial, int rows, int rows_pad, const float* __restrict__ bias, const u16* residual,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, u16* out) {
    constexpr int hidden = 512 * STEPS;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    f32x4 p0[STEPS][S], p1[STEPS][S], bb[STEPS][2], g[STEPS][2], b[STEPS][2], rrv[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = lane * 8 + 512 * s;
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            const float* q = partial + ((int64_t)sl * rows_pad + r) * hidden + c;
            p0[s][sl] = ld16_issue(q);
            p1[s][sl] = ld16_issue(q + 4);
        }
        rrv[s] = ld16_issue(residual + (int64_t)r * hidden + c);
        bb[s][0] = ld16_issue(bias + c);
        bb[s][1] = ld16_issue(bias + c + 4);
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = lane * 8 + 512 * s;
        g[s][0] = ld16_issue(gamma + c);
        g[s][1] = ld16_issue(gamma + c + 4);
        b[s][0] = ld16_issue(beta + c);
        b[s][1] = ld16_issue(beta + c + 4);
    }
    ld16_wait_all();   // (at most 2 * (2 S + 7) = 46 loads at S = 8: inside the 6-bit counter)
    uint4 rr[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            ld16_pin(p0[s][sl]);
            ld16_pin(p1[s][sl]);
        }
        ld16_pin(rrv[s]);
        ld16_pin(bb[s][0]); ld16_pin(bb[s][1]);
        ld16_pin(g[s][0]); ld16_pin(g[s][1]);
        ld16_pin(b[s][0]); ld16_pin(b[s][1]);
        rr[s] = make_uint4(__float_as_uint(rrv[s].x), __float_as_uint(rrv[s].y), __float_as_uint(rrv[s].z), __float_as_uint(rrv[s].w));
    }
    float x[STEPS][8];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        f32x4 v0 = p0[s][0], v1 = p1[s][0];
#pragma unroll
        for (int sl = 1; sl < S; ++sl) {
            v0 += p0[s][sl];
            v1 += p1[s][sl];
        }
        v0 += bb[s][0];
        v1 += bb[s][1];
        x[s][0] = bf2f(f2bf(v0.x + bf2f((u16)(rr[s].x & 0xffff))));
        x[s][1] = bf2f(f2bf(v0.y + bf2f((u16)(rr[s].x >> 16))));
        x[s][2] = bf2f(f2bf(v0.z + bf2f((u16)(rr[s].y & 0xffff))));
        x[s][3] = bf2f(f2bf(v0.w + bf2f((u16)(rr[s].y >> 16))));
        x[s][4] = bf2f(f2bf(v1.x + bf2f((u16)(rr[s].z & 0xffff))));
        x[s][5] = bf2f(f2bf(v1.y + bf2f((u16)(rr[s].z >> 16))));
        x[s][6] = bf2f(f2bf(v1.z + bf2f((u16)(rr[s].w & 0xffff))));
        x[s][7] = bf2f(f2bf(v1.w + bf2f((u16)(rr[s].w >> 16))));
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += x[s][e];
    const float mean = wave_sum_e(sum) / (float)hidden;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = x[s][e] - mean;
            sq = fmaf(d, d, sq);
        }
    const float rstd = rsqrtf(wave_sum_e(sq) / (float)hidden + eps);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = lane * 8 + 512 * s;
        Vals8 o;
        o.v[0] = (x[s][0] - mean) * rstd * g[s][0].x + b[s][0].x;
        o.v[1] = (x[s][1] - mean) * rstd * g[s][0].y + b[s][0].y;
        o.v[2] = (x[s][2] - mean) * rstd * g[s][0].z + b[s][0].z;
        o.v[3] = (x[s][3] - mean) * rstd * g[s][0].w + b[s][0].w;
        o.v[4] = (x[s][4] - mean) * rstd * g[s][1].x + b[s][1].x;
        o.v[5] = (x[s][5] - mean) * rstd * g[s][1].y + b[s][1].y;
        o.v[6] = (x[s][6] - mean) * rstd * g[s][1].z + b[s][1].z;
        o.v[7] = (x[s][7] - mean) * rstd * g[s][1].w + b[s][1].w;
        store8_bf16(out + (int64_t)r * hidden + c, o);
    }
}

namespace {
struct EnvSlot {
    const char* name;
    const char* value;
    unsigned long scope;
};
thread_local EnvSlot t_env[48];
thread_local int t_env_n = 0;
thread_local unsigned long t_env_scope = 1;
}  // namespace

void rass_env_new_scope() { ++t_env_scope; }

const char* rass_env(const char* name) {
    for (int i = 0; i < t_env_n; ++i) {
        if (t_env[i].name == name) {
            if (t_env[i].scope != t_env_scope) {
                t_env[i].value = getenv(name);
                t_env[i].scope = t_env_scope;
            }
            return t_env[i].value;
        }
    }
    const char* v = getenv(name);
    if (t_env_n < 48) t_env[t_env_n++] = EnvSlot{name, v, t_env_scope};
    return v;
}

hipError_t launch_splitk_residual_layernorm(const float* partial, int S, int rows, int rows_pad, int hidden,
                                            const float* bias, const void* residual, const float* gamma,
                                            const float* beta, float eps, void* out, hipStream_t stream) {
    if (hidden % 8 != 0 || hidden > 512 * kMaxSteps || S < 1) return hipErrorInvalidValue;
    if (rows <= 0) return hipSuccess;
    const int blocks = (rows + 3) / 4;
    const u16* res = static_cast<const u16*>(residual);
    u16* o = static_cast<u16*>(out);
    if (hidden == 1024 && (S == 2 || S == 4 || S == 8)) {   // BERT-large rows: the query-time reduction (FFN-down's four K slices)
        const char* v = rass_env("RASS_LN_EXACT");            // and the mid-size batches' (8); =0: the general kernel (A/B; per launch)
        if (!(v && v[0] == '0')) {
            if (S == 2)
                hipLaunchKernelGGL((splitk_residual_layernorm_exact_kernel<2, 2>), dim3(blocks), dim3(kRowThreadsE), 0, stream,
                                   partial, rows, rows_pad, bias, res, gamma, beta, eps, o);
            else if (S == 4)
                hipLaunchKernelGGL((splitk_residual_layernorm_exact_kernel<4, 2>), dim3(blocks), dim3(kRowThreadsE), 0, stream,
                                   partial, rows, rows_pad, bias, res, gamma, beta, eps, o);
            else
                hipLaunchKernelGGL((splitk_residual_layernorm_exact_kernel<8, 2>), dim3(blocks), dim3(kRowThreadsE), 0, stream,
                                   partial, rows, rows_pad, bias, res, gamma, beta, eps, o);
            return hipGetLastError();
        }
    }
#define RASS_SKLN(SV)                                                                                              \
    case SV:                                                                                                       \
        hipLaunchKernelGGL(splitk_residual_layernorm_kernel<SV>, dim3(blocks), dim3(kRowThreadsE), 0, stream, partial, \
                           rows, rows_pad, hidden, bias, res, gamma, beta, eps, o);                                \
        break;
    switch (S) {
        RASS_SKLN(1) RASS_SKLN(2) RASS_SKLN(4) RASS_SKLN(8) RASS_SKLN(16)
        default: return hipErrorInvalidValue;
    }
#undef RASS_SKLN
    return hipGetLastError();
}

hipError_t launch_layernorm(const void* in, const float* gamma, const float* beta, float eps, int rows, int hidden,
                            void* out, hipStream_t stream) {
    if (hidden % 8 != 0 || hidden > 512 * kMaxSteps) return hipErrorInvalidValue;
    if (rows <= 0) return hipSuccess;
    // one row per wave while that fills the device; beyond 8 workgroups per CU the waves walk rows (the hoisted
    // gamma / beta and the prefetch pay from the second row on: 16 rows per wave at 256 x 512 tokens)
    int blocks = (rows + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const int steps = (hidden + 511) / 512;
    const u16* x = static_cast<const u16*>(in);
    u16* y = static_cast<u16*>(out);
    // Whole batches read their input rows nontemporal (they are read once; a query's rows stay in L2 between its kernels).
    // RASS_LN_NT (read per launch; bit 0 = nontemporal loads, bit 1 = nontemporal stores) is the A/B: 256 x 512-token ingest
    // 3 094 chunks/s plain, 3 115 with the loads, 3 073-3 079 with the stores too (the next GEMM reads these rows 4-16 times)
    int nt_mode = rows >= 4096 ? 1 : 0;
    if (rows >= 4096)
        if (const char* v = rass_env("RASS_LN_NT")) nt_mode = atoi(v) & 3;
    switch (steps) {
        case 1: hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(kRowThreadsE), 0, stream, x, gamma, beta, eps, rows, hidden, y, nt_mode); break;
        case 2: hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(kRowThreadsE), 0, stream, x, gamma, beta, eps, rows, hidden, y, nt_mode); break;
        case 3: hipLaunchKernelGGL(layernorm_kernel<3>, dim3(blocks), dim3(kRowThreadsE), 0, stream, x, gamma, beta, eps, rows, hidden, y, nt_mode); break;
        default: hipLaunchKernelGGL(layernorm_kernel<4>, dim3(blocks), dim3(kRowThreadsE), 0, stream, x, gamma, beta, eps, rows, hidden, y, nt_mode); break;
    }
    return hipGetLastError();
}

// K8: one wave per sequence.  mean: fp32 sum over the sequence's tokens in token order, then
// / n_tokens (what sentence-transformers' mean pooling does with an all-ones mask on real
// tokens); cls: the first token.  Optional reference normalise e / (||e|| + 1e-9).
__global__ __launch_bounds__(kRowThreadsE) void pool_kernel(const u16* __restrict__ x, const int32_t* __restrict__ cu,
                                                            int nseq, int hidden, int mode_mean, int normalize,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int s = blockIdx.x * 4 + wave; s < nseq; s += gridDim.x * 4) {
        const int t0 = cu[s], t1 = cu[s + 1];
        Vals8 acc[kMaxSteps];
#pragma unroll
        for (int st = 0; st < kMaxSteps; ++st)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[st].v[e] = 0.f;
        const int last = mode_mean ? t1 : (t1 > t0 ? t0 + 1 : t0);
        // 8 tokens' loads are in flight before the first is added (one dependent 16-B load per token
        // made this kernel latency-bound: 515 us for 268 MB); the sums still run in token order.
        constexpr int kTok = 8;
        for (int tb = t0; tb < last; tb += kTok) {
            uint4 raw[kTok][kMaxSteps];
#pragma unroll
            for (int u = 0; u < kTok; ++u) {
                const int t = tb + u < last ? tb + u : last - 1;  // clamp: re-read, not added
#pragma unroll
                for (int st = 0; st < kMaxSteps; ++st) {
                    const int c = lane * 8 + 512 * st;
                    raw[u][st] = c < hidden ? *reinterpret_cast<const uint4*>(x + (int64_t)t * hidden + c)
                                            : uint4{0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int u = 0; u < kTok; ++u) {
                if (tb + u >= last) break;
#pragma unroll
                for (int st = 0; st < kMaxSteps; ++st) {
                    const uint4 r = raw[u][st];
                    acc[st].v[0] += bf2f((u16)(r.x & 0xffff));
                    acc[st].v[1] += bf2f((u16)(r.x >> 16));
                    acc[st].v[2] += bf2f((u16)(r.y & 0xffff));
                    acc[st].v[3] += bf2f((u16)(r.y >> 16));
                    acc[st].v[4] += bf2f((u16)(r.z & 0xffff));
                    acc[st].v[5] += bf2f((u16)(r.z >> 16));
                    acc[st].v[6] += bf2f((u16)(r.w & 0xffff));
                    acc[st].v[7] += bf2f((u16)(r.w >> 16));
                }
            }
        }
        const float inv_n = (mode_mean && t1 > t0) ? 1.f / (float)(t1 - t0) : 1.f;
        float ss = 0.f;
#pragma unroll
        for (int st = 0; st < kMaxSteps; ++st)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[st].v[e] *= inv_n;
                if (lane * 8 + 512 * st < hidden) ss = fmaf(acc[st].v[e], acc[st].v[e], ss);
            }
        const float denom = normalize ? sqrtf(wave_sum_e(ss)) + 1e-9f : 1.f;
#pragma unroll
        for (int st = 0; st < kMaxSteps; ++st) {
            const int c = lane * 8 + 512 * st;
            if (c < hidden) {
                float* dst = out + (int64_t)s * hidden + c;
                *reinterpret_cast<f32x4*>(dst) = f32x4{acc[st].v[0] / denom, acc[st].v[1] / denom, acc[st].v[2] / denom,
                                                       acc[st].v[3] / denom};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[st].v[4] / denom, acc[st].v[5] / denom,
                                                           acc[st].v[6] / denom, acc[st].v[7] / denom};
            }
        }
    }
}

hipError_t launch_pool(const void* x, const int32_t* cu_seqlens, int nseq, int hidden, int mode_mean, int normalize,
                       float* out, hipStream_t stream) {
    if (hidden % 8 != 0 || hidden > 512 * kMaxSteps) return hipErrorInvalidValue;
    if (nseq <= 0) return hipSuccess;
    int blocks = (nseq + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pool_kernel, dim3(blocks), dim3(kRowThreadsE), 0, stream, static_cast<const u16*>(x),
                       cu_seqlens, nseq, hidden, mode_mean, normalize, out);
    return hipGetLastError();
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, u16* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f2bf(in[i]);
}

hipError_t launch_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, static_cast<u16*>(out), n);
    return hipGetLastError();
}

}  // namespace rass
