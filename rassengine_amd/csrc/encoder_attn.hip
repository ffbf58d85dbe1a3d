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
// attention32_kernel: the same algorithm on 32x32 tiles (v_mfma_f32_32x32x16_bf16), 8 waves of 32 queries.
//
// Why (counted on the 16x16 kernel's ISA, prices from MI355X_MICROARCH "vector-instruction ISSUE cost"): per 64 keys a
// wave of the 16x16 kernel spends ~80 plain VALU issues + 16 v_exp on 16 scores per lane — 33 of them the maximum
// (canonicalising v_max before every fmaxf on an MFMA output, two ds_bpermute rounds with lgkmcnt(0) each), 6 the row
// sum's shuffles — and reads 16 KiB of K / V fragments from LDS for 16 queries.  Here:
//   * S^T tile = 32 keys x 32 queries: lane (q = lane & 31, h = lane >> 5) holds keys (r&3) + 8(r>>2) + 4h of the tile,
//     so a query's scores live in TWO lanes: the block maximum is 16 v_max3_f32 in the lane + one v_permlane32_swap
//     (no LDS round trip), and the row sum stays a per-lane partial until the epilogue (both lanes of a query scale it
//     by the same alpha), one swap per 512 keys instead of two shuffles per 64;
//   * a K or V fragment (1 KiB of LDS traffic) now feeds 32 queries: half the LDS bytes per score;
//   * the exponentiated scores are still the next MFMA's B operand as they lie (k-slot j of lane half h <-> key
//     (j&3) + 8(j>>2) + 4h of a 16-key step; the V^T operand is fetched in that key order: two ds_read_b64_tr_b16 at
//     rows +0 and +8);
//   * QK^T of block kb+1 is issued BEFORE the softmax of block kb (two score buffers, the loop unrolled by two), so
//     the matrix pipe works under the wave's own VALU phase and not only under the other wave's;
//   * V rows have the K pitch (128 B) with the 64-byte halves exchanged on rows with bit 1 set: the 4 rows x 64 B a
//     half-wave's transposed read touches fall on 64 distinct banks; K + V = 128 KiB at S = 512;
//   * O is exchanged between the two lanes of a query (v_permlane32_swap) so that each stores 16 contiguous bytes.
constexpr int kA32Threads = 512;
constexpr int kA32Waves = kA32Threads / 64;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// fmaxf on MFMA outputs: this file is built with -fno-honor-nans (Makefile), so no canonicalising v_max precedes each
// one and chains fold into v_max3_f32 (scores are finite or the -inf of a masked key, never NaN)
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float max2f(float a, float b) { return __builtin_fmaxf(a, b); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// One sequence-head's K and V rows, held in registers between their loads (issued while the PREVIOUS item is being
// computed) and their LDS writes: chunk e = threadIdx.x + 512 i  ->  key e >> 3, 16-B column chunk e & 7.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct KvRegs {
    u32x4 k[8], v[8];
};

__global__ __launch_bounds__(kA32Threads) void attention32_kernel(const u16* __restrict__ qkv,
                                                                  const int32_t* __restrict__ cu, int hidden, int heads,
                                                                  int s_pad, int n_items, u16* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Kl = lds;                            // [s_pad][128 B], 16-B chunk c at c ^ ((key >> 1) & 7)
    unsigned char* Vl = lds + (size_t)s_pad * 128;      // [s_pad][128 B], 16-B chunk c at c ^ (((key >> 1) & 1) << 2)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ld = 3 * hidden;
    const int h = lane >> 5, qi = lane & 31;
    // transposed V read: lane i of a 16-lane group addresses key row (i >> 2), d columns 4 (i & 3) .. +3 of the group's
    // 16-d block (block 2 dt + ((lane >> 4) & 1)); the row's swizzle bit is (i >> 3) & 1 for every row this lane touches
    const int vi = lane & 15;
    const int v_sw = (vi >> 3) & 1;
    const int v_lane_off = (4 * h + (vi >> 2)) * 128 + (2 * ((lane >> 4) & 1) + ((vi & 3) >> 1)) * 16 + (vi & 1) * 8;
    constexpr float kScale = 0.18033688011112042f;  // 1/8 * log2(e)

    // rows of item `it` (sequence it / heads, head it % heads): token offset and length
    const int total = cu[n_items / heads];
    if (total <= 0) return;
    auto item_rows = [&](int it, int& t0, int& S) {
        const int seq = it / heads;
        t0 = cu[seq];
        S = cu[seq + 1] - t0;
        if (S > s_pad) S = s_pad;
    };
    // Issue the loads of an item's K / V rows.  NO branch around any of them (a row past the sequence re-reads its last
    // row and is zeroed at store time; an item past the last re-reads the current one): the compiler counts the loads
    // in flight per basic block, and one conditional load among them turns the wait for the Q fragments below into
    // "all but 7 done" — the whole prefetch waited for before the first MFMA.
    auto load_kv = [&](int it, KvRegs& r) {
        int t0, S;
        item_rows(it, t0, S);
        const u16* kb_ = qkv + (it % heads) * kHeadDim + hidden + (threadIdx.x & 7) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = (threadIdx.x >> 3) + 64 * i;
            int tok = t0 + (key < S ? key : S - 1);
            tok = tok < 0 ? 0 : (tok < total ? tok : total - 1);
#ifndef RASS_ATTN_EXP_NO_STAGE
            r.k[i] = *reinterpret_cast<const u32x4*>(kb_ + (int64_t)tok * ld);
            r.v[i] = *reinterpret_cast<const u32x4*>(kb_ + hidden + (int64_t)tok * ld);
#else
            r.k[i] = u32x4{(unsigned)tok, 0u, 0u, 0u};
            r.v[i] = u32x4{0u, (unsigned)tok, 0u, 0u};
#endif
        }
    };
    auto store_kv = [&](int S, const KvRegs& r) {  // keys >= S: zeros — masked later, but they must be finite
        const int c = threadIdx.x & 7;
        const int rows = (S + 63) / 64 * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = (threadIdx.x >> 3) + 64 * i;
            if (key < rows) {
                const unsigned keep = key < S ? 0xffffffffu : 0u;  // a value mask: no select between two lvalues
                *reinterpret_cast<u32x4*>(Kl + key * 128 + ((c ^ ((key >> 1) & 7)) * 16)) = r.k[i] & keep;
                *reinterpret_cast<u32x4*>(Vl + key * 128 + ((c ^ (((key >> 1) & 1) << 2)) * 16)) = r.v[i] & keep;
            }
        }
    };

    // Persistent: workgroup b takes items b, b + grid, b + 2 grid, ...  While item i is being computed, the K / V rows
    // of item i + 1 are on their way into 64 registers per thread (all of LDS belongs to item i: the 128 KiB leave no
    // room for a second image), so the ~9-12 k cycles every workgroup of a one-workgroup-per-CU kernel spends waiting
    // for its 128 KiB at ~11 B/clk/CU (MI355X_MICROARCH "prologue HBM burst"; 138 of 469 us per launch measured with
    // the loads removed) run under the previous item's MFMAs.
    KvRegs kv;
    int item = blockIdx.x;
    load_kv(item, kv);
    while (true) {
        int t0, S;
        item_rows(item, t0, S);
        const int head = item % heads;
        const int n_kb = (S + 63) / 64;
        // this item's Q fragments for both passes first (vmcnt retires in order: a Q load issued after the next item's
        // 16 K / V loads would wait for all of them), B operand of S^T = K Q^T: Q[q][16 ks + 8 h .. +7]
        bf16x8 qf[2][4];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int q = ps * kA32Waves * 32 + wave * 32 + qi;
            int tok = t0 + (q < S ? q : S - 1);
            tok = tok < 0 ? 0 : (tok < total ? tok : total - 1);
            const u16* qrow = qkv + (int64_t)tok * ld + head * kHeadDim + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ps][ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks);
        }
        store_kv(S, kv);
        __syncthreads();
        const int next = item + gridDim.x;
        load_kv(next < n_items ? next : item, kv);

        auto run_pass = [&](int q0, const bf16x8 (&qq)[4]) {
            const int q = q0 + qi;
            float m_run = -INFINITY, l_part = 0.f;
            f32x16 O[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { O[0][r] = 0.f; O[1][r] = 0.f; }
            for (int kb = 0; kb < n_kb; ++kb) {
                f32x16 s[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
                    const int krow = kb * 64 + kt * 32 + qi;  // A operand row = key
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int c = (2 * ks + h) ^ ((krow >> 1) & 7);
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kl + krow * 128 + c * 16);
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qq[ks], s[kt], 0, 0, 0);
                    }
                }
                if (kb * 64 + 64 > S) {  // wave-uniform: the ragged last block
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kb * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= S) s[kt][r] = -INFINITY;
                }
#ifndef RASS_ATTN_EXP_NO_SOFTMAX
                // 32 scores -> 1: a tree of v_max3_f32 (depth 4), then the partner lane's
                float t1[11];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    t1[i] = max3f(s[0][3 * i], s[0][3 * i + 1], s[0][3 * i + 2]);
                    t1[5 + i] = max3f(s[1][3 * i], s[1][3 * i + 1], s[1][3 * i + 2]);
                }
                t1[10] = max2f(s[0][15], s[1][15]);
                const float t2a = max3f(t1[0], t1[1], t1[2]), t2b = max3f(t1[3], t1[4], t1[5]);
                const float t2c = max3f(t1[6], t1[7], t1[8]), t2d = max2f(t1[9], t1[10]);
                float mx = max2f(max3f(t2a, t2b, t2c), t2d);
                {
                    const auto sw =
                        __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = max2f(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                const float m_new = max2f(m_run, mx * kScale);  // finite: key 0 of block 0 is always valid
                if (__any(m_new > m_run)) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 1 where the maximum did not move
                    l_part *= alpha;
                    O[0] *= alpha;
                    O[1] *= alpha;
                    m_run = m_new;
                }
                f32x16 negm;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m_run;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const f32x16 ex = s[kt] * kScale + negm;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] = __builtin_amdgcn_exp2f(ex[r]);
                    acc += s[kt];
                }
                l_part += ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])) +
                          ((acc[8] + acc[9]) + (acc[10] + acc[11])) + ((acc[12] + acc[13]) + (acc[14] + acc[15]));
#endif
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                        u32x4v pw;
#pragma unroll
                        for (int i = 0; i < 4; ++i) pw[i] = pack_bf16(s[kt][8 * j + 2 * i], s[kt][8 * j + 2 * i + 1]);
                        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
                        const unsigned char* vblk = Vl + (kb * 64 + kt * 32 + 16 * j) * 128 + v_lane_off;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const unsigned char* va = vblk + ((dt ^ v_sw) * 64);
                            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) bf16x4*)(va));
                            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) bf16x4*)(va + 8 * 128));
                            bf16x8 vf;
                            vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                            vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                            O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, O[dt], 0, 0, 0);
                        }
                    }
                }
            }
            // the query's sum = this lane's partial + its partner's; O^T[d = 32 dt + (r&3) + 8(r>>2) + 4h][q]
            float l_run;
            {
                const auto sw =
                    __builtin_amdgcn_permlane32_swap(__float_as_uint(l_part), __float_as_uint(l_part), false, false);
                l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            const float inv_l = 1.f / l_run;
            u16* dst = ctx + (int64_t)(t0 + (q < S ? q : 0)) * hidden + head * kHeadDim + 8 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                unsigned w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = pack_bf16(O[dt][2 * i] * inv_l, O[dt][2 * i + 1] * inv_l);
                // (w0,w1) = d 4h+0..3, (w2,w3) = d 8+4h.., (w4,w5) = d 16+4h.., (w6,w7) = d 24+4h..  ->  the low lane
                // keeps d 0-7 and 16-23 of the 32-block, the high lane d 8-15 and 24-31
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(w[4 * p + 0], w[4 * p + 2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(w[4 * p + 1], w[4 * p + 3], false, false);
                    if (q < S)
                        *reinterpret_cast<uint4*>(dst + 32 * dt + 16 * p) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        };
        if (wave * 32 < S) run_pass(wave * 32, qf[0]);
        if (kA32Waves * 32 + wave * 32 < S) run_pass(kA32Waves * 32 + wave * 32, qf[1]);

        if (next >= n_items) break;
        item = next;
        __syncthreads();  // every wave is done reading this item's K / V image
    }
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
    const char* v = getenv("RASS_ATTN_VARIANT");
    return v ? v : "";
}

hipError_t launch_attention(const void* qkv, const int32_t* cu_seqlens, int nseq, int max_seqlen, int hidden,
                            int heads, void* ctx, hipStream_t stream) {
    if (hidden != heads * kHeadDim || max_seqlen < 1 || max_seqlen > 512) return hipErrorInvalidValue;
    if (nseq <= 0) return hipSuccess;
    const int s_pad = (max_seqlen + 63) / 64 * 64;
    if (strcmp(attn_variant(), "w16") != 0) {
        const size_t lds_bytes = (size_t)s_pad * 256;  // 128 KiB at S = 512
        static size_t attr32_bytes = 0;
        if (lds_bytes > attr32_bytes) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention32_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            attr32_bytes = lds_bytes;
        }
        const int n_items = nseq * heads;
        hipLaunchKernelGGL(attention32_kernel, dim3(n_items < attn_cus() ? n_items : attn_cus()), dim3(kA32Threads),
                           lds_bytes, stream, static_cast<const u16*>(qkv), cu_seqlens, hidden, heads, s_pad, n_items,
                           static_cast<u16*>(ctx));
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
