// gemm_retired_kernels.hip — ARCHIVE of the encoder GEMM kernels that the persistent "p5" kernel replaced, moved out of
// librass_hip.so in round 3 (dead weight on the hot path's build and review surface).  Each was measured
// slower than p5 on the encoder's shapes (profiles/r01_gemm_phases_microbench.txt, r01_gemm_phase_timers_microbench.txt,
// r02_gemm_w4_experiments.txt); the numbers in DESIGN.md §4 / §10 that name them come from these sources:
//   gemm_bf16_ring_kernel   256^2 x 32 tiles on a 3-slot LDS ring, one tile per block           (round 1)
//   gemm_bf16_pring_kernel  its persistent form, next tile's first steps under the epilogue       (round 1 default)
//   gemm_bf16_p64_kernel    64-deep K steps on two 64-KiB slots (whole cache lines per row)       (round 2)
//   gemm_bf16_w4l_kernel    four waves of 128 x 128, accumulators pinned in AGPRs                 (round 2)
// It still builds and runs (it includes the product source for the shared helpers and times every variant against p5):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_retired.bin gemm_retired_kernels.hip ../../rassengine_amd/csrc/encoder_misc.hip && ./gemm_retired.bin
// Not part of the product; nothing under rassengine_amd/ refers to it.
#ifndef RASS_RETIRED_NO_MAIN
#define RASS_RETIRED_WITH_MAIN 1
#endif
#include "../../rassengine_amd/csrc/encoder_gemm.hip"

namespace rass {


// ------------------------------------------------------------------------------------------
// Large-shape kernel: 256 (N) x 256 (M) x 32 (K) tiles, 512 threads = 2 (N) x 4 (M) waves, each
// wave 128 x 64 = 8 x 4 MFMA tiles (128 accumulator VGPRs).  Versus the 128^2 kernel above
// it halves the staging instructions per MFMA (4 global_load_lds per 32 MFMAs per wave
// instead of 8) and replaces "vmcnt(0) + barrier every K step" by a 3-slot LDS ring with a
// COUNTED wait: while step t is multiplied, the DMAs of steps t+1 and t+2 are in flight;
// `s_waitcnt vmcnt(4)` (= all but the 4 pieces this wave just issued for t+2) retires t+1,
// then ONE raw s_barrier per step publishes it (guide §5 "Pipelining across barriers": never
// __syncthreads() here, its fence drains vmcnt(0)).  Hazards: RAW — a slot is read one
// iteration after the wait+barrier that retired it; WAR — slot (t+2)%3 == (t-1)%3 was last
// read in iteration t-1, and every wave passed that iteration's closing barrier before any
// wave issues the t+2 DMAs.
// LDS image: rows of 64 B (32 bf16), 16 rows per 1 KiB DMA piece, lane-linear; the
// bank-conflict swizzle chunk ^= 3*((row>>3)&1) lives on the SOURCE address and on the read.
constexpr int RBK = 32;   // (RBM, RBN: product file)
constexpr int kRingTileBytes = 256 * RBK * 2;        // 16 KiB per operand per slot
constexpr int kRingSlotBytes = 2 * kRingTileBytes;   // W tile | X tile
constexpr int kRingSlots = 3;             // 96 KiB: steps t+1, t+2 in flight while t is multiplied (a 4th
                                          // slot measured no gain: the K loop is not latency-bound)
constexpr int kRingAhead = kRingSlots - 1;

// Wait until at most `steps` whole K steps of this wave's DMAs (4 pieces each) are outstanding.
__device__ __forceinline__ void ring_wait_steps(int steps) {
    if (steps >= 2) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (steps == 1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int EPI>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_bf16_ring_kernel(const u16* __restrict__ X,
                                                                        const u16* __restrict__ W,
                                                                        const float* __restrict__ bias,
                                                                        const u16* __restrict__ residual,
                                                                        u16* __restrict__ Y, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [3 slots][W tile | X tile]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 2, wm = wave & 3;
    const int nblk = gridDim.x, orig = blockIdx.x;
#ifdef RASS_GEMM_CLOCKS  // scripts/microbench/gemm_phases.hip only
    if (threadIdx.x == 0) g_gemm_clocks[4 * blockIdx.x] = wall_clock64();
#endif
    const int q = nblk / 8, rr = nblk % 8, xcd = orig % 8;
    const int bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    const int tiles_n = N / RBN;
    const int bn = bid % tiles_n, bm = bid / tiles_n;
    const int n0 = bn * RBN, m0 = bm * RBM;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / RBK;
    // Loop-invariant addressing, hoisted: the K loop then issues ~1 VALU per 4 MFMAs instead of
    // ~2 per MFMA (measured: SQ_INSTS_VALU / MFMA = 1.8 with the addresses recomputed per step).
    //   DMA: this wave moves pieces {wave, wave+8} of each operand tile; per lane one source
    //   pointer per piece, advanced by RBK elements per step; LDS destination = slot + piece*1024.
    //   Fragments: byte offset inside a slot of each of the 8 A (W) and 4 B (X) fragments.
    const u16* srcW[2];
    const u16* srcX[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (wave + 8 * p) * 16 + (lane >> 2);
        const int c_src = (lane & 3) ^ (((r >> 3) & 1) * 3);
        srcW[p] = W + (int64_t)(n0 + r) * K + c_src * 8;
        srcX[p] = X + (int64_t)(m0 + r) * K + c_src * 8;
    }
    int offA[8], offB[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wn * 128 + i * 16 + (lane & 15);
        offA[i] = row * 64 + (((lane >> 4) ^ (((row >> 3) & 1) * 3)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wm * 64 + j * 16 + (lane & 15);
        offB[j] = kRingTileBytes + row * 64 + (((lane >> 4) ^ (((row >> 3) & 1) * 3)) * 16);
    }
    auto stage_step = [&](unsigned char* slot_base) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcW[p],
                                             (__attribute__((address_space(3))) void*)(slot_base + (wave + 8 * p) * 1024),
                                             16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)srcX[p],
                (__attribute__((address_space(3))) void*)(slot_base + kRingTileBytes + (wave + 8 * p) * 1024), 16, 0, 0);
            srcW[p] += RBK;
            srcX[p] += RBK;
        }
    };
    {
        const int pre = nk < kRingAhead ? nk : kRingAhead;  // steps 0 .. pre-1 go out; step 0 must land
        for (int s0 = 0; s0 < pre; ++s0) stage_step(lds + s0 * kRingSlotBytes);
        ring_wait_steps(pre - 1);
    }
    __builtin_amdgcn_s_barrier();

#ifdef RASS_GEMM_CLOCKS
    if (threadIdx.x == 0) g_gemm_clocks[4 * blockIdx.x + 1] = wall_clock64();
    const unsigned long long core0 = clock64();
#endif
    // Stagger (MI355X_MICROARCH "two waves per SIMD" item 9): every K step is a LOAD phase (DMA
    // issue for step t+2 + this step's 12 LDS fragment reads) and a COMPUTE phase (32 MFMAs),
    // each closed by a barrier.  Waves 4-7 (the SIMD partners of waves 0-3) run ONE phase behind:
    // while a SIMD's older wave multiplies, its partner issues DMAs and LDS reads, instead of
    // both doing the same thing at the same time.  Step t+1 must be complete before the FIRST
    // reader (group A, interval 2t+2): group A retires its pieces at the end of its compute
    // phase, group B at the end of its load phase — both are interval 2t+1.  WAR on slot
    // (t+2)%3 == (t-1)%3: its last reads (group B's load phase of step t-1, drained by
    // lgkmcnt(0) before the barrier) end in interval 2t-1, the first new DMA is interval 2t.
    const bool grpB = wave >= 4;
    if (grpB) __builtin_amdgcn_s_barrier();
#ifdef RASS_GEMM_PHASE_TIMERS
    unsigned long long ph_load = 0, ph_bar1 = 0, ph_comp = 0, ph_bar2 = 0;
#endif
    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        const bool more = t + kRingAhead < nk;
        // steps still in flight once step t+1 has been retired: t+2 .. min(t+kRingAhead, nk-1)
        const int keep = more ? kRingAhead - 1 : (nk - 2 - t > 0 ? nk - 2 - t : 0);
#ifdef RASS_GEMM_PHASE_TIMERS
        const unsigned long long pt0 = clock64();
#endif
        // ---- load phase
        if (more) {
            int s2 = slot + kRingAhead;
            s2 = s2 >= kRingSlots ? s2 - kRingSlots : s2;
            stage_step(lds + s2 * kRingSlotBytes);
        }
        const unsigned char* buf = lds + slot * kRingSlotBytes;
        bf16x8 a[8], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(buf + offB[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(buf + offA[i]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grpB) ring_wait_steps(keep);
        __builtin_amdgcn_sched_barrier(0);
#ifdef RASS_GEMM_PHASE_TIMERS
        const unsigned long long pt1 = clock64();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef RASS_GEMM_PHASE_TIMERS
        const unsigned long long pt2 = clock64();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- compute phase
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (!grpB) ring_wait_steps(keep);
        __builtin_amdgcn_sched_barrier(0);
#ifdef RASS_GEMM_PHASE_TIMERS
        const unsigned long long pt3 = clock64();
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef RASS_GEMM_PHASE_TIMERS
        {
            const unsigned long long pt4 = clock64();
            ph_load += pt1 - pt0; ph_bar1 += pt2 - pt1; ph_comp += pt3 - pt2; ph_bar2 += pt4 - pt3;
        }
#endif
        slot = slot + 1 >= kRingSlots ? 0 : slot + 1;
    }
    if (!grpB) __builtin_amdgcn_s_barrier();  // both groups execute the same number of barriers
#ifdef RASS_GEMM_CLOCKS
    if (threadIdx.x == 0) g_gemm_clocks[4 * blockIdx.x + 2] = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x < 64) g_gemm_core_cycles[blockIdx.x] = clock64() - core0;
#endif
#ifdef RASS_GEMM_PHASE_TIMERS
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 64) {
        unsigned long long* o = g_gemm_phase_cycles + (blockIdx.x * 2 + (wave >> 2)) * 4;
        o[0] = ph_load; o[1] = ph_bar1; o[2] = ph_comp; o[3] = ph_bar2;
    }
#endif

    // Epilogue through LDS (free after the last barrier).  The accumulator layout gives every
    // lane 4 consecutive features of ONE token per tile, i.e. 8-byte stores scattered over 16
    // token rows per instruction: measured 1.7 TB/s, ~45 % of a K=1024 GEMM's time.  Instead
    // each wave transposes 32-token x 64-feature chunks (fp32, 8.5 KiB of its private 12 KiB
    // LDS region) and writes them back as 16-byte stores, 8 lanes = 128 contiguous bytes per
    // token row; bias / residual / GELU are applied on the coalesced side.
    {
        constexpr int kPitchF = 68;  // floats per token row of the chunk image (64 + pad)
        float* stg = reinterpret_cast<float*>(lds + wave * (kRingSlots * kRingSlotBytes / 8));
        const int tl = lane >> 3, nq = lane & 7;  // read-back: token row (of 8 per pass), 8-feature group
        // Every global read of the epilogue is issued before it is needed (a dependent load costs
        // 0.5-1 us here and there would be 4 bias + 16 residual ones per wave in a row): bias for
        // both feature halves up front, the residual rows of a chunk before its LDS staging.
        f32x4 bv[2][2];
#pragma unroll
        for (int ic = 0; ic < 2; ++ic) {
            const int nb = n0 + wn * 128 + ic * 64 + nq * 8;
            bv[ic][0] = *reinterpret_cast<const f32x4*>(bias + nb);
            bv[ic][1] = *reinterpret_cast<const f32x4*>(bias + nb + 4);
        }
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) {
#pragma unroll
            for (int ic = 0; ic < 2; ++ic) {
                const int nbase = n0 + wn * 128 + ic * 64 + nq * 8;
                uint4 res[4];
                if (EPI == 1) {
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        const int m = m0 + wm * 64 + jc * 32 + pass * 8 + tl;
                        res[pass] = m < M ? *reinterpret_cast<const uint4*>(residual + (int64_t)m * N + nbase)
                                          : uint4{0u, 0u, 0u, 0u};
                    }
                }
                // stage: tiles i = 4ic..4ic+3, j = 2jc..2jc+1
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        *reinterpret_cast<f32x4*>(stg + (jj * 16 + (lane & 15)) * kPitchF + ii * 16 + (lane >> 4) * 4) =
                            acc[4 * ic + ii][2 * jc + jj];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS write -> read
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int tok = pass * 8 + tl;
                    const int m = m0 + wm * 64 + jc * 32 + tok;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8 + 4);
                    v0 += bv[ic][0];
                    v1 += bv[ic][1];
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
                    if (EPI == 2) {
                        v0.x = gelu_erf(v0.x); v0.y = gelu_erf(v0.y); v0.z = gelu_erf(v0.z); v0.w = gelu_erf(v0.w);
                        v1.x = gelu_erf(v1.x); v1.y = gelu_erf(v1.y); v1.z = gelu_erf(v1.z); v1.w = gelu_erf(v1.w);
                    }
                    if (m < M) {
                        uint4 o;
                        o.x = (unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16);
                        o.y = (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16);
                        o.z = (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16);
                        o.w = (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16);
                        *reinterpret_cast<uint4*>(Y + (int64_t)m * N + nbase) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next chunk overwrites
            }
        }
    }
#ifdef RASS_GEMM_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) g_gemm_clocks[4 * blockIdx.x + 3] = wall_clock64();
#endif
}

template <int EPI>
static hipError_t launch_ring(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M,
                              int M_pad, int N, int K, hipStream_t stream) {
    constexpr int lds_bytes = kRingSlots * kRingSlotBytes;  // 96 KiB at 3 slots
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_ring_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (N / RBN) * (M_pad / RBM);
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI>), dim3(grid), dim3(kRingThreads), lds_bytes, stream, X, W, bias,
                       residual, Y, M, N, K);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Persistent form of the ring kernel: one workgroup per CU walks tiles pos, pos+G, pos+2G, ...
// Per tile the non-persistent kernel pays ~2 us of pipeline prologue (first DMAs in flight, MFMA
// idle) and ~2 us of block turnover (wave launch, drain of the last stores before the block
// retires) on top of a ~28 us K loop and a 6-9 us epilogue (scripts/microbench/gemm_phases.hip).
// Here the first kRingAhead K steps of the NEXT tile are put in flight right after the K loop and
// land during the epilogue, and the epilogue's stores drain under the next tile's K loop.
// LDS: the ring (3 x 32 KiB) + epilogue staging that must not overlap slots 0/1 (prefetch target):
// waves 0-2 stage in slot 2, waves 3-7 behind the ring.
// RASS_PRING_SLOTS=4 (experiment): a 4-slot ring, three K steps in flight; the next tile's three first steps then
// occupy slots 0-2 during the epilogue, so the staging shrinks to 16-token chunks (4 352 B per wave): waves 0-6 in
// slot 3, wave 7 behind the ring.
#ifndef RASS_PRING_SLOTS
#define RASS_PRING_SLOTS 3
#endif
constexpr int kPSlots = RASS_PRING_SLOTS;
constexpr int kPAhead = kPSlots - 1;
static_assert(kPSlots == 3, "the product file fixes kPStageTokens = 32 (3 slots)");
constexpr int kPringStageBytes = kPStageTokens * 68 * 4;                               // 8704 B per wave (3 slots)
constexpr int kPringLdsBytes = kPSlots == 3 ? kPSlots * kRingSlotBytes + 5 * kPringStageBytes      // 141 824 B
                                            : kPSlots * kRingSlotBytes + 1 * kPringStageBytes;     // 135 424 B
static_assert(kPSlots == 3 || kPSlots == 4, "ring of 3 or 4 slots");
static_assert(kPSlots != 3 || 3 * kPringStageBytes <= kRingSlotBytes, "3 staging areas share slot 2");
static_assert(kPSlots != 4 || 7 * kPringStageBytes <= kRingSlotBytes, "7 staging areas share slot 3");
// wait until at most `steps` whole K steps of this wave's DMAs (4 pieces each) are outstanding
__device__ __forceinline__ void pring_wait_steps(int steps) {
    if (steps >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (steps == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


template <int EPI>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_bf16_pring_kernel(const u16* __restrict__ X,
                                                                         const u16* __restrict__ W,
                                                                         const float* __restrict__ bias,
                                                                         const u16* __restrict__ residual,
                                                                         u16* __restrict__ Y, int M, int N, int K,
                                                                         int tiles_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 2, wm = wave & 3;
    const int G = gridDim.x, orig = blockIdx.x;
    // position inside a round of G tiles: XCD x (= blockIdx % 8) owns a contiguous eighth, so the
    // N tiles that share an X panel run on one XCD (one L2) at the same time
    const int pos = (G % 8 == 0) ? (orig % 8) * (G / 8) + orig / 8 : orig;
    int tile = pos;
    if (tile >= tiles_total) return;
    const int tiles_n = N / RBN;
    const int nk = K / RBK;
    const int pre = nk < kPAhead ? nk : kPAhead;

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned offA0, offB0;
    {
        const int rowA = wn * 128 + (lane & 15), rowB = wm * 64 + (lane & 15);
        offA0 = rowA * 64 + (((lane >> 4) ^ (((rowA >> 3) & 1) * 3)) * 16);
        offB0 = kRingTileBytes + rowB * 64 + (((lane >> 4) ^ (((rowB >> 3) & 1) * 3)) * 16);
    }
    const u16* srcW[2];
    const u16* srcX[2];
    auto point_at = [&](int t) {
#ifdef RASS_GEMM_EXP_SAME_TILE   // timing experiment: every workgroup streams tile 0's operands (all L2 hits)
        const int tn0 = 0, tm0 = 0;
        (void)t;
#else
        const int tn0 = (t % tiles_n) * RBN, tm0 = (t / tiles_n) * RBM;
#endif
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = (wave + 8 * p) * 16 + (lane >> 2);
            const int c_src = (lane & 3) ^ (((r >> 3) & 1) * 3);
            srcW[p] = W + (int64_t)(tn0 + r) * K + c_src * 8;
            srcX[p] = X + (int64_t)(tm0 + r) * K + c_src * 8;
        }
    };
    auto stage_step = [&](unsigned char* slot_base) {
#ifdef RASS_GEMM_EXP_NO_DMA      // timing experiment: no operand delivery at all (stale LDS)
        (void)slot_base;
        return;
#endif
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcW[p],
                                             (__attribute__((address_space(3))) void*)(slot_base + (wave + 8 * p) * 1024),
                                             16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)srcX[p],
                (__attribute__((address_space(3))) void*)(slot_base + kRingTileBytes + (wave + 8 * p) * 1024), 16, 0, 0);
            srcW[p] += RBK;
            srcX[p] += RBK;
        }
    };
    float* const stg = reinterpret_cast<float*>(
        kPSlots == 3 ? (wave < 3 ? lds + 2 * kRingSlotBytes + wave * kPringStageBytes
                                 : lds + kPSlots * kRingSlotBytes + (wave - 3) * kPringStageBytes)
                     : (wave < 7 ? lds + 3 * kRingSlotBytes + wave * kPringStageBytes : lds + kPSlots * kRingSlotBytes));
    const bool grpB = wave >= 4;

    point_at(tile);
    for (int s0 = 0; s0 < pre; ++s0) stage_step(lds + s0 * kRingSlotBytes);
    pring_wait_steps(pre - 1);
    __builtin_amdgcn_s_barrier();

    for (;;) {
        const int n0 = (tile % tiles_n) * RBN, m0 = (tile / tiles_n) * RBM;
#ifdef RASS_GEMM_CLOCKS
        if (threadIdx.x == 0) g_gemm_clocks[4 * tile] = g_gemm_clocks[4 * tile + 1] = wall_clock64();
#endif
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- K loop: identical to gemm_bf16_ring_kernel (group B one phase behind group A)
        if (grpB) __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int t = 0; t < nk; ++t) {
            const bool more = t + kPAhead < nk;
            // steps still in flight once step t+1 has been retired: t+2 .. min(t+kPAhead, nk-1)
            const int keep = more ? kPAhead - 1 : (nk - 2 - t > 0 ? nk - 2 - t : 0);
            if (more) {
                int s2 = slot + kPAhead;
                s2 = s2 >= kPSlots ? s2 - kPSlots : s2;
                stage_step(lds + s2 * kRingSlotBytes);
            }
            bf16x8 a[8], b[4];
#ifdef RASS_GEMM_EXP_NO_MFMA    // timing experiment: the operand stream alone (DMA + waits + barriers)
            for (int i = 0; i < 8; ++i) a[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 4; ++j) b[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#else
            {
                // fragment i / j of a slot sits i / j KiB after fragment 0 (16 rows x 64 B; the swizzle
                // term only depends on lane bits)
                const unsigned ab = lds_base + slot * kRingSlotBytes + offA0;
                const unsigned bb = lds_base + slot * kRingSlotBytes + offB0;
                RASS_DS_READ_B128(b[0], bb, 0);
                RASS_DS_READ_B128(b[1], bb, 1024);
                RASS_DS_READ_B128(b[2], bb, 2048);
                RASS_DS_READ_B128(b[3], bb, 3072);
                RASS_DS_READ_B128(a[0], ab, 0);
                RASS_DS_READ_B128(a[1], ab, 1024);
                RASS_DS_READ_B128(a[2], ab, 2048);
                RASS_DS_READ_B128(a[3], ab, 3072);
                RASS_DS_READ_B128(a[4], ab, 4096);
                RASS_DS_READ_B128(a[5], ab, 5120);
                RASS_DS_READ_B128(a[6], ab, 6144);
                RASS_DS_READ_B128(a[7], ab, 7168);
            }
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (grpB) pring_wait_steps(keep);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#ifndef RASS_GEMM_EXP_NO_MFMA
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#endif
            __builtin_amdgcn_s_setprio(0);
            if (!grpB) pring_wait_steps(keep);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            slot = slot + 1 >= kPSlots ? 0 : slot + 1;
        }
        if (!grpB) __builtin_amdgcn_s_barrier();  // groups re-aligned: every ring read is done, no DMA in flight
#ifdef RASS_GEMM_CLOCKS
        if (threadIdx.x == 0) g_gemm_clocks[4 * tile + 2] = wall_clock64();
#endif

        // ---- next tile's first K steps go out now and land under the epilogue (slots 0 .. pre-1)
        const int next = tile + G;
        const bool has_next = next < tiles_total;
        if (has_next) {
            point_at(next);
            for (int s0 = 0; s0 < pre; ++s0) stage_step(lds + s0 * kRingSlotBytes);
        }

        // ---- epilogue (see gemm_bf16_ring_kernel): LDS transpose per wave, coalesced 16-B stores
        {
            constexpr int kPitchF = 68;
            const int tl = lane >> 3, nq = lane & 7;
            // Bias through opaque asm loads, retired by the explicit vmcnt(0) below: a load hipcc can
            // see stays "possibly pending" on its destination registers across the tile loop, and
            // when the K loop's fragment reads get the same registers the waitcnt pass protects them
            // with a vmcnt(0) in EVERY K step (seen in two of the three epilogue variants).
            f32x4 bv[2][2];
#pragma unroll
            for (int ic = 0; ic < 2; ++ic) {
                const float* bp = bias + n0 + wn * 128 + ic * 64 + nq * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[ic][0]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv[ic][1]) : "v"(bp));
            }
#pragma unroll
            for (int jc = 0; jc < 64 / kPStageTokens; ++jc) {
#pragma unroll
                for (int ic = 0; ic < 2; ++ic) {
                    const int nbase = n0 + wn * 128 + ic * 64 + nq * 8;
                    uint4 res[4];
                    if (EPI == 1) {
#pragma unroll
                        for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                            const int m = m0 + wm * 64 + jc * kPStageTokens + pass * 8 + tl;
                            res[pass] = m < M ? *reinterpret_cast<const uint4*>(residual + (int64_t)m * N + nbase)
                                              : uint4{0u, 0u, 0u, 0u};
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < kPStageTokens / 16; ++jj)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii)
                            *reinterpret_cast<f32x4*>(stg + (jj * 16 + (lane & 15)) * kPitchF + ii * 16 + (lane >> 4) * 4) =
                                acc[4 * ic + ii][(kPStageTokens / 16) * jc + jj];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (jc == 0 && ic == 0) {
                        // Explicit: this wave's prefetch DMAs (and the bias / first residual reads issued
                        // after them) are complete before anything below consumes them and before the
                        // publishing barrier after the epilogue.  No store is outstanding yet.
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
#pragma unroll
                    for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                        const int tok = pass * 8 + tl;
                        const int m = m0 + wm * 64 + jc * kPStageTokens + tok;
                        f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8);
                        f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8 + 4);
                        v0 += bv[ic][0];
                        v1 += bv[ic][1];
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
                        if (EPI == 2) {
                            v0.x = gelu_erf(v0.x); v0.y = gelu_erf(v0.y); v0.z = gelu_erf(v0.z); v0.w = gelu_erf(v0.w);
                            v1.x = gelu_erf(v1.x); v1.y = gelu_erf(v1.y); v1.z = gelu_erf(v1.z); v1.w = gelu_erf(v1.w);
                        }
                        if (m < M) {
                            uint4 o;
                            o.x = (unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16);
                            o.y = (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16);
                            o.z = (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16);
                            o.w = (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16);
                            *reinterpret_cast<uint4*>(Y + (int64_t)m * N + nbase) = o;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
#ifdef RASS_GEMM_CLOCKS
        if (threadIdx.x == 0) g_gemm_clocks[4 * tile + 3] = wall_clock64();
#endif
        if (!has_next) break;
        // The epilogue's stores stay in flight: they are older than every DMA of the next K loop in the
        // in-order vmcnt queue, so the first counted wait there also retires them (they have had the
        // whole first K step to drain).
        // publish the next tile's first steps: every wave retired its own pieces (vmcnt(0) above)
        // and finished reading its staging area (slot 2 is a DMA target again from step 0 on)
        __builtin_amdgcn_s_barrier();
        tile = next;
    }
}

// ------------------------------------------------------------------------------------------
// 64-deep K steps ("p64"): the persistent ring kernel with whole cache lines per row.  With a 32-deep step every row
// of an operand tile is asked for in 64-B segments — two L2 requests per 128-B line — and the counters showed the L2
// request rate, not latency or misses, to be the operand stream's ceiling (L2 channels busy 78 % of the launch at
// 58 B per request; the same bytes asked for as 128-B segments stream 40 % faster:
// profiles/r02_gemm_w4_experiments.txt).  Here a ring slot holds K = 64: rows of 128 B, a DMA piece = 8 rows x 128 B
// whose 8 lanes per row coalesce into ONE request, the bank-conflict swizzle of the 128 x 128 kernel (16-B chunk c
// of row r at chunk c ^ ((r>>1)&7)).  A slot is 64 KiB, so the ring has TWO slots, one step in flight; a step is two
// 32-deep sub-steps, each a load phase (fragment reads) and a compute phase (32 MFMAs) closed by a barrier, waves 4-7
// one phase behind waves 0-3 as before.  Hazards: step t+1's DMAs go out in the load phase of sub-step 0 of step t
// (group A in phase 4t, group B in 4t+1) into the slot step t-1 was read from, whose last reads (group B, phase
// 4t-1) are behind a barrier; they are retired by a vmcnt(0) in front of the barrier that ends phase 4t+3 (group A
// at the end of its second compute phase, group B of its second load phase), the barrier group A's first reads of
// step t+1 follow.  Epilogue staging: waves 0-6 in slot 1, wave 7 behind the ring (the next tile's step 0 lands in
// slot 0 meanwhile).
constexpr int kP64TileBytes = 256 * 64 * 2;            // 32 KiB per operand per slot
constexpr int kP64SlotBytes = 2 * kP64TileBytes;       // W tile | X tile
constexpr int kP64StageBytes = 32 * 68 * 4;            // 8704 B per wave
constexpr int kP64LdsBytes = 2 * kP64SlotBytes + kP64StageBytes;   // 139 776 B
static_assert(7 * kP64StageBytes <= kP64SlotBytes, "7 staging areas share slot 1");

template <int EPI>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_bf16_p64_kernel(const u16* __restrict__ X,
                                                                       const u16* __restrict__ W,
                                                                       const float* __restrict__ bias,
                                                                       const u16* __restrict__ residual,
                                                                       u16* __restrict__ Y, int M, int N, int K,
                                                                       int tiles_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 2, wm = wave & 3;
    const int G = gridDim.x, orig = blockIdx.x;
    const int pos = (G % 8 == 0) ? (orig % 8) * (G / 8) + orig / 8 : orig;
    int tile = pos;
    if (tile >= tiles_total) return;
    const int tiles_n = N / RBN;
    const int nk = K / 64;

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // fragment i / j of a slot: 16 rows x 128 B = 2 KiB after fragment 0; sub-step s reads chunks 4s + (lane>>4),
    // stored at chunk ^ ((row>>1)&7): the swizzle term only depends on lane bits, sub-step 1 = sub-step 0 ^ 64 B
    unsigned offA[2], offB[2];
    {
        const int rowA = wn * 128 + (lane & 15), rowB = wm * 64 + (lane & 15);
        const int sw = ((lane & 15) >> 1) & 7;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ch = (sub * 4 + (lane >> 4)) ^ sw;
            offA[sub] = rowA * 128 + ch * 16;
            offB[sub] = kP64TileBytes + rowB * 128 + ch * 16;
        }
    }
    // DMA: an operand tile is 32 pieces of 8 rows x 128 B; this wave moves pieces wave, wave+8, wave+16, wave+24
    const u16* srcW[4];
    const u16* srcX[4];
    auto point_at = [&](int t) {
        const int tn0 = (t % tiles_n) * RBN, tm0 = (t / tiles_n) * RBM;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = (wave + 8 * p) * 8 + (lane >> 3);
            const int c_src = (lane & 7) ^ ((r >> 1) & 7);
            srcW[p] = W + (int64_t)(tn0 + r) * K + c_src * 8;
            srcX[p] = X + (int64_t)(tm0 + r) * K + c_src * 8;
        }
    };
    // a step's DMAs in two halves: the X pieces first (their lines are new in every step: the long latencies), the W
    // pieces (mostly L2 hits) half a sub-step later, so that no phase carries all eight issues
    auto stage_x = [&](unsigned char* slot_base) {
#ifdef RASS_GEMM_EXP_NO_DMA      // timing experiment: no operand delivery at all (stale LDS)
        (void)slot_base;
        return;
#endif
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)srcX[p],
                (__attribute__((address_space(3))) void*)(slot_base + kP64TileBytes + (wave + 8 * p) * 1024), 16, 0, 0);
            srcX[p] += 64;
        }
    };
    auto stage_w = [&](unsigned char* slot_base) {
#ifdef RASS_GEMM_EXP_NO_DMA
        (void)slot_base;
        return;
#endif
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcW[p],
                                             (__attribute__((address_space(3))) void*)(slot_base + (wave + 8 * p) * 1024),
                                             16, 0, 0);
            srcW[p] += 64;
        }
    };
    auto stage_step = [&](unsigned char* slot_base) {
        stage_x(slot_base);
        stage_w(slot_base);
    };
    float* const stg = reinterpret_cast<float*>(wave < 7 ? lds + kP64SlotBytes + wave * kP64StageBytes
                                                         : lds + 2 * kP64SlotBytes);
    const bool grpB = wave >= 4;

    point_at(tile);
    stage_step(lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (;;) {
        const int n0 = (tile % tiles_n) * RBN, m0 = (tile / tiles_n) * RBM;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (grpB) __builtin_amdgcn_s_barrier();   // group B runs one phase behind group A
        for (int t = 0; t < nk; ++t) {
            const int slot = t & 1;
            const bool more = t + 1 < nk;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                // ---- load phase
                if (sub == 0 && more) stage_x(lds + (slot ^ 1) * kP64SlotBytes);
                bf16x8 a[8], b[4];
#ifdef RASS_GEMM_EXP_NO_MFMA    // timing experiment: the operand stream alone (DMA + waits + barriers)
                for (int i = 0; i < 8; ++i) a[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                for (int j = 0; j < 4; ++j) b[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#else
                {
                    const unsigned ab = lds_base + slot * kP64SlotBytes + offA[sub];
                    const unsigned bb = lds_base + slot * kP64SlotBytes + offB[sub];
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
                if (sub == 1 && grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of step t+1
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
                if (sub == 0 && more) stage_w(lds + (slot ^ 1) * kP64SlotBytes);
                if (sub == 1 && !grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!grpB) __builtin_amdgcn_s_barrier();  // groups re-aligned: every ring read is done, no DMA in flight

        // ---- next tile's first K step goes out now and lands under the epilogue (slot 0)
        const int next = tile + G;
        const bool has_next = next < tiles_total;
        if (has_next) {
            point_at(next);
            stage_step(lds);
        }

        // ---- epilogue (see gemm_bf16_ring_kernel): LDS transpose per wave, coalesced 16-B stores
        {
            constexpr int kPitchF = 68;
            const int tl = lane >> 3, nq = lane & 7;
            // Bias through opaque asm loads, retired by the explicit vmcnt(0) below: a load hipcc can
            // see stays "possibly pending" on its destination registers across the tile loop, and
            // when the K loop's fragment reads get the same registers the waitcnt pass protects them
            // with a vmcnt(0) in EVERY K step (seen in two of the three epilogue variants).
            f32x4 bv[2][2];
#pragma unroll
            for (int ic = 0; ic < 2; ++ic) {
                const float* bp = bias + n0 + wn * 128 + ic * 64 + nq * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[ic][0]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv[ic][1]) : "v"(bp));
            }
#pragma unroll
            for (int jc = 0; jc < 64 / kPStageTokens; ++jc) {
#pragma unroll
                for (int ic = 0; ic < 2; ++ic) {
                    const int nbase = n0 + wn * 128 + ic * 64 + nq * 8;
                    uint4 res[4];
                    if (EPI == 1) {
#pragma unroll
                        for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                            const int m = m0 + wm * 64 + jc * kPStageTokens + pass * 8 + tl;
                            res[pass] = m < M ? *reinterpret_cast<const uint4*>(residual + (int64_t)m * N + nbase)
                                              : uint4{0u, 0u, 0u, 0u};
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < kPStageTokens / 16; ++jj)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii)
                            *reinterpret_cast<f32x4*>(stg + (jj * 16 + (lane & 15)) * kPitchF + ii * 16 + (lane >> 4) * 4) =
                                acc[4 * ic + ii][(kPStageTokens / 16) * jc + jj];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (jc == 0 && ic == 0) {
                        // Explicit: this wave's prefetch DMAs (and the bias / first residual reads issued
                        // after them) are complete before anything below consumes them and before the
                        // publishing barrier after the epilogue.  No store is outstanding yet.
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
#pragma unroll
                    for (int pass = 0; pass < kPStageTokens / 8; ++pass) {
                        const int tok = pass * 8 + tl;
                        const int m = m0 + wm * 64 + jc * kPStageTokens + tok;
                        f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8);
                        f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + tok * kPitchF + nq * 8 + 4);
                        v0 += bv[ic][0];
                        v1 += bv[ic][1];
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
                        if (EPI == 2) {
                            v0.x = gelu_erf(v0.x); v0.y = gelu_erf(v0.y); v0.z = gelu_erf(v0.z); v0.w = gelu_erf(v0.w);
                            v1.x = gelu_erf(v1.x); v1.y = gelu_erf(v1.y); v1.z = gelu_erf(v1.z); v1.w = gelu_erf(v1.w);
                        }
                        if (m < M) {
                            uint4 o;
                            o.x = (unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16);
                            o.y = (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16);
                            o.z = (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16);
                            o.w = (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16);
                            *reinterpret_cast<uint4*>(Y + (int64_t)m * N + nbase) = o;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
        if (!has_next) break;
        __builtin_amdgcn_s_barrier();
        tile = next;
    }
}


template <int EPI>
static hipError_t launch_p64(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M,
                             int M_pad, int N, int K, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_p64_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kP64LdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0)
            n_cus = 256;
    }
    const int tiles_total = (N / RBN) * (M_pad / RBM);
    const int grid = tiles_total < n_cus ? tiles_total : n_cus;
    hipLaunchKernelGGL((gemm_bf16_p64_kernel<EPI>), dim3(grid), dim3(kRingThreads), kP64LdsBytes, stream, X, W, bias,
                       residual, Y, M, N, K, tiles_total);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_pring(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M,
                               int M_pad, int N, int K, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pring_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kPringLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0)
            n_cus = 256;
    }
    const int tiles_total = (N / RBN) * (M_pad / RBM);
    const int grid = tiles_total < n_cus ? tiles_total : n_cus;
    hipLaunchKernelGGL((gemm_bf16_pring_kernel<EPI>), dim3(grid), dim3(kRingThreads), kPringLdsBytes, stream, X, W, bias,
                       residual, Y, M, N, K, tiles_total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// 4-wave form (RASS_GEMM_VARIANT=w4l, an A/B variant: correct, slower than the 8-wave kernels on this part): one
// workgroup of FOUR waves per CU, every wave a 128 x 128 tile (64 MFMAs per 32-deep sub-step against 16 fragment reads:
// two thirds of the 8-wave kernel's LDS reads per MFMA, one barrier per sub-step, no second wave competing for the
// SIMD's issue slots).  What makes it possible:
//   * the 256 fp32 accumulators live in AGPRs: the MFMA is issued as inline asm with "a" constraints (the builtin
//     form made hipcc spill 115 registers and shuffle accumulators through v_accvgpr_mov, DESIGN round 1);
//   * with one wave per SIMD nothing hides a wave's own latencies, so everything is software-pipelined in registers:
//     fragments double-buffered, operand pieces staged through 32 registers (buffer_load_dwordx4 + ds_write_b128, not
//     LDS-DMA: a global_load_lds piece costs the issuing wave 60-185 cycles, which only a partner wave can hide);
//   * no LDS transpose in the epilogue: the W rows are staged PERMUTED (MFMA row rho of M-tile i holds output feature
//     32*(rho>>2) + 4*i + (rho&3) of the wave's 128), so the 4 accumulator rows a lane owns in the 8 tiles of a token
//     column are 32 CONSECUTIVE features — bias, residual, GELU and the bf16 pack happen in registers and a lane
//     writes its 64 contiguous bytes with four 16-B stores.
// History and measurements of its three 64-B-segment predecessors: profiles/r02_gemm_w4_experiments.txt.
constexpr int kW4Threads = 256;

#define RASS_MFMA_BF16_ACC(acc, a, b) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define RASS_MFMA_BF16_NEW(acc, a, b) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b))

// ------------------------------------------------------------------------------------------
// The kernel: the structure above (AGPR accumulators, register-pipelined fragments and staging, register epilogue
// over permuted W rows) on the LDS layout of gemm_bf16_p64_kernel: rows of
// 128 B (K = 64 per slot, two 64-KiB slots), so a staging load covers 8 rows x 128 B — one L2 request per row where
// w4 asked for two 64-B halves (the L2 request rate was what bounded w4: profiles/r02_gemm_w4_experiments.txt).
// The K loop is one stream of 32-deep SUB-steps u across tiles; HALF-loads q = 2T + h (rows 128h .. 128h+127 of both
// operand tiles of step T) are loaded from global memory during sub-step q-4 into 32 staging registers and written
// to LDS during sub-step q-3; the fragments of sub-step u+1 are read during sub-step u.  Slot T&1 holds step T:
// half-load q goes into the slot whose last fragment reads (sub-step 2(T-1)+1, issued during 2(T-1)) are behind the
// barrier of the sub-step that writes it.  K % 128 == 0, so that a tile's sub-step count is a multiple of 4 and the
// slot / register-set parities carry over from tile to tile.
constexpr int kW4LLdsBytes = 2 * kP64SlotBytes;  // 128 KiB

template <int EPI>
__global__ __launch_bounds__(kW4Threads) void gemm_bf16_w4l_kernel(const u16* __restrict__ X, const u16* __restrict__ W,
                                                                   const float* __restrict__ bias,
                                                                   const u16* __restrict__ residual, u16* __restrict__ Y,
                                                                   int M, int N, int K, int tiles_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int G = gridDim.x, orig = blockIdx.x;
    const int pos = (G % 8 == 0) ? (orig % 8) * (G / 8) + orig / 8 : orig;
    int tile = pos;
    if (tile >= tiles_total) return;
    const int tiles_n = N / RBN;
    const int nu = K / 32;  // sub-steps per tile: a multiple of 4, >= 8 (launcher)

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // fragment i / j: 16 rows x 128 B = 2 KiB after fragment 0; sub-step s reads chunk 4s + (lane>>4), stored at
    // chunk ^ ((row>>1)&7)
    unsigned offA[2], offB[2];
    {
        const int rowA = wn * 128 + (lane & 15), rowB = wm * 128 + (lane & 15);
        const int sw = ((lane & 15) >> 1) & 7;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ch = (sub * 4 + (lane >> 4)) ^ sw;
            offA[sub] = rowA * 128 + ch * 16;
            offB[sub] = kP64TileBytes + rowB * 128 + ch * 16;
        }
    }
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u16*>(W), 0, (int)(unsigned)((uint64_t)N * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u16*>(X), 0, (int)(unsigned)((uint64_t)(tiles_total / tiles_n) * RBM * K * 2), 0x00020000);
    // An operand tile of a step is 32 pieces of 8 rows x 128 B; half-load h of this wave = pieces wave + 4p + 16h.
    // Image row r = piece*8 + (lane>>3).  X: source row = r.  W: image row (half hh, M-tile i, MFMA row rho) <- feature
    // hh*128 + 32*(rho>>2) + 4*i + (rho&3), which for this wave's pieces is base(wave, lane) + 128h + 8p: both
    // operands need ONE per-lane offset each, the (h, p) part goes into the load's scalar offset.
    unsigned voffW = 0, voffX = 0;
    int koff = 0;  // byte offset into K of the step whose half-loads go out next
    const int lr = lane >> 3;
    const int c_src = (lane & 7) ^ ((((wave & 1) * 8 + lr) >> 1) & 7);
    const int featbase = 64 * (wave & 1) + 32 * (lr >> 2) + 4 * (wave >> 1) + (lr & 3);
    auto point_at = [&](int t) {
        const int tn0 = (t % tiles_n) * RBN, tm0 = (t / tiles_n) * RBM;
        voffW = ((unsigned)(tn0 + featbase) * (unsigned)K + c_src * 8) * 2u;
        voffX = ((unsigned)(tm0 + wave * 8 + lr) * (unsigned)K + c_src * 8) * 2u;
        koff = 0;
    };
    bf16x8 g[2][8];  // staging registers
    const int rowbytes = K * 2;
    auto load_half = [&g, &voffW, &voffX, &koff, rsrcW, rsrcX, rowbytes](auto set_c, auto h_c) {
        constexpr int set = decltype(set_c)::value;
        constexpr int h = decltype(h_c)::value;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            g[set][p] = __builtin_bit_cast(
                bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcW, voffW, koff + (128 * h + 8 * p) * rowbytes, 0));
            g[set][4 + p] = __builtin_bit_cast(
                bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voffX, koff + (128 * h + 32 * p) * rowbytes, 0));
        }
        if (h == 1) koff += 128;
    };
    unsigned char* const my_piece = lds + wave * 1024 + lane * 16;
    auto write_piece = [&g, my_piece](auto set_c, int slot, int h, int p) {
        constexpr int set = decltype(set_c)::value;
        unsigned char* dst = my_piece + slot * kP64SlotBytes + (4 * p + 16 * h) * 1024;
        *reinterpret_cast<bf16x8*>(dst) = g[set][p];
        *reinterpret_cast<bf16x8*>(dst + kP64TileBytes) = g[set][4 + p];
    };

    f32x4 acc[8][8];
    bf16x8 fa[2][8], fb[2][8];
    auto read_frags = [&fa, &fb, lds_base, &offA, &offB](auto set_c, int slot, int sub) {
        constexpr int set = decltype(set_c)::value;
        const unsigned ab = lds_base + slot * kP64SlotBytes + offA[sub];
        const unsigned bb = lds_base + slot * kP64SlotBytes + offB[sub];
        RASS_DS_READ_B128(fa[set][0], ab, 0);     RASS_DS_READ_B128(fb[set][0], bb, 0);
        RASS_DS_READ_B128(fa[set][1], ab, 2048);  RASS_DS_READ_B128(fb[set][1], bb, 2048);
        RASS_DS_READ_B128(fa[set][2], ab, 4096);  RASS_DS_READ_B128(fb[set][2], bb, 4096);
        RASS_DS_READ_B128(fa[set][3], ab, 6144);  RASS_DS_READ_B128(fb[set][3], bb, 6144);
        RASS_DS_READ_B128(fa[set][4], ab, 8192);  RASS_DS_READ_B128(fb[set][4], bb, 8192);
        RASS_DS_READ_B128(fa[set][5], ab, 10240); RASS_DS_READ_B128(fb[set][5], bb, 10240);
        RASS_DS_READ_B128(fa[set][6], ab, 12288); RASS_DS_READ_B128(fb[set][6], bb, 12288);
        RASS_DS_READ_B128(fa[set][7], ab, 14336); RASS_DS_READ_B128(fb[set][7], bb, 14336);
    };

    // Sub-step u of a tile (SUB = u & 1 = the fragment register set it multiplies; SLOT = (u >> 1) & 1):
    //   FIRST     the tile's first sub-step writes the accumulators
    //   READS     fragments of sub-step u+1 (slot SLOT sub 1, or slot SLOT^1 sub 0) into the other register set
    //   do_write  half-load q = u+3 (loaded during sub-step u-1 into g[SUB^1]) goes to its slot: half SUB^1 of slot
    //             SLOT^1 (SUB = 0) or of slot SLOT (SUB = 1: this slot's last reads were issued a sub-step ago)
    //   do_load   half-load q = u+4 = half SUB of step T+2 into g[SUB]
    auto kstep = [&](auto sub_c, auto slot_c, auto first_c, auto reads_c, bool do_write, bool do_load) {
        constexpr int sub = decltype(sub_c)::value;
        constexpr int slot = decltype(slot_c)::value;
        constexpr bool first = decltype(first_c)::value;
        constexpr bool reads = decltype(reads_c)::value;
        constexpr int set = sub;
        constexpr int rslot = sub == 0 ? slot : slot ^ 1, rsub = sub ^ 1;
        constexpr int wslot = sub == 0 ? slot ^ 1 : slot, wh = sub ^ 1;
        const unsigned ab = lds_base + rslot * kP64SlotBytes + offA[rsub];
        const unsigned bb = lds_base + rslot * kP64SlotBytes + offB[rsub];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this sub-step's fragments, this wave's last ds_writes
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (first) RASS_MFMA_BF16_NEW(acc[0][j], fa[set][0], fb[set][j]);
            else RASS_MFMA_BF16_ACC(acc[0][j], fa[set][0], fb[set][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (first) RASS_MFMA_BF16_NEW(acc[i][j], fa[set][i], fb[set][j]);
                else RASS_MFMA_BF16_ACC(acc[i][j], fa[set][i], fb[set][j]);
            }
            if (i == 1 && do_load) load_half(std::integral_constant<int, set>{}, std::integral_constant<int, sub>{});
            if (reads) {
                if (i == 1) { RASS_DS_READ_B128(fa[set ^ 1][0], ab, 0);     RASS_DS_READ_B128(fb[set ^ 1][0], bb, 0);
                              RASS_DS_READ_B128(fb[set ^ 1][1], bb, 2048);  RASS_DS_READ_B128(fb[set ^ 1][2], bb, 4096); }
                if (i == 2) { RASS_DS_READ_B128(fb[set ^ 1][3], bb, 6144);  RASS_DS_READ_B128(fb[set ^ 1][4], bb, 8192);
                              RASS_DS_READ_B128(fb[set ^ 1][5], bb, 10240); RASS_DS_READ_B128(fb[set ^ 1][6], bb, 12288); }
                if (i == 3) { RASS_DS_READ_B128(fb[set ^ 1][7], bb, 14336); RASS_DS_READ_B128(fa[set ^ 1][1], ab, 2048);
                              RASS_DS_READ_B128(fa[set ^ 1][2], ab, 4096);  RASS_DS_READ_B128(fa[set ^ 1][3], ab, 6144); }
                if (i == 4) { RASS_DS_READ_B128(fa[set ^ 1][4], ab, 8192);  RASS_DS_READ_B128(fa[set ^ 1][5], ab, 10240);
                              RASS_DS_READ_B128(fa[set ^ 1][6], ab, 12288); RASS_DS_READ_B128(fa[set ^ 1][7], ab, 14336); }
            }
            if (i >= 4 && do_write) write_piece(std::integral_constant<int, set ^ 1>{}, wslot, wh, i - 4);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;

    // pipeline prologue of the first tile: half-loads 0, 1 (step 0 -> slot 0) and 2 (half 0 of step 1 -> slot 1) into
    // LDS, half-load 3 into g[1]
    point_at(tile);
    load_half(C0{}, C0{});
    load_half(C1{}, C1{});
#pragma unroll
    for (int p = 0; p < 4; ++p) write_piece(C0{}, 0, 0, p);
#pragma unroll
    for (int p = 0; p < 4; ++p) write_piece(C1{}, 0, 1, p);
    load_half(C0{}, C0{});
#pragma unroll
    for (int p = 0; p < 4; ++p) write_piece(C0{}, 1, 0, p);
    load_half(C1{}, C1{});

    for (;;) {
        const int n0 = (tile % tiles_n) * RBN, m0 = (tile / tiles_n) * RBM;
        const int next = tile + G;
        const bool has_next = next < tiles_total;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(C0{}, 0, 0);
        //     sub  slot first reads write load
        kstep(C0{}, C0{}, BT{}, BT{}, true, true);
        kstep(C1{}, C0{}, BF{}, BT{}, true, true);
        kstep(C0{}, C1{}, BF{}, BT{}, true, true);
        kstep(C1{}, C1{}, BF{}, BT{}, true, true);
        for (int u = 4; u + 4 < nu; u += 4) {
            kstep(C0{}, C0{}, BF{}, BT{}, true, true);
            kstep(C1{}, C0{}, BF{}, BT{}, true, true);
            kstep(C0{}, C1{}, BF{}, BT{}, true, true);
            kstep(C1{}, C1{}, BF{}, BT{}, true, true);
        }
        // the tile's last four sub-steps: their half-loads (and, from the second on, writes) belong to the next tile
        if (has_next) point_at(next);
        kstep(C0{}, C0{}, BF{}, BT{}, true, has_next);       // u = nu-4: writes this tile's last half-load
        kstep(C1{}, C0{}, BF{}, BT{}, has_next, has_next);   // u = nu-3: next tile's half-load 0 -> slot 0
        kstep(C0{}, C1{}, BF{}, BT{}, has_next, has_next);   // u = nu-2: half-load 1 -> slot 0
        kstep(C1{}, C1{}, BF{}, BF{}, has_next, has_next);   // u = nu-1: half-load 2 -> slot 1; 3 waits in g[1]

        // ---- epilogue in registers: lane (g = lane>>4, c = lane&15) owns features 32g .. 32g+31 of token 16j + c.
        // Order of this wave's memory operations (vmcnt retires in order): the epilogue's READS (bias, the whole
        // residual tile: 128 registers the fragments no longer need), then the next tile's step 2 (8 loads, which
        // land under the epilogue), then the stores.
        {
            const int gq = lane >> 4, c = lane & 15;
            const int fbase = n0 + wn * 128 + 32 * gq;
            // bias / residual through opaque asm loads: a load hipcc can see stays "possibly pending" on its
            // destination registers across the tile loop and costs a vmcnt(0) in front of the K loop's first MFMA
            f32x4 bv[8];
            {
                const float* bp = bias + fbase;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[0]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv[1]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(bv[2]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(bv[3]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(bv[4]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:80" : "=v"(bv[5]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:96" : "=v"(bv[6]) : "v"(bp));
                asm volatile("global_load_dwordx4 %0, %1, off offset:112" : "=v"(bv[7]) : "v"(bp));
            }
            u32x4 res[8][4];
            if (EPI == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int m = m0 + wm * 128 + j * 16 + c;
                    m = m < M ? m : M - 1;  // rows past M are never stored: read a valid row instead of branching
                    const u16* rp = residual + (int64_t)m * N + fbase;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(res[j][0]) : "v"(rp));
                    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(res[j][1]) : "v"(rp));
                    asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(res[j][2]) : "v"(rp));
                    asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(res[j][3]) : "v"(rp));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the reads (and the staged half-load issued before them)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results (inline asm: no hazard tracking)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = m0 + wm * 128 + j * 16 + c;
                u16* yp = Y + (int64_t)m * N + fbase;
#pragma unroll
                for (int ip = 0; ip < 4; ++ip) {
                    __builtin_amdgcn_sched_barrier(0);  // accumulators leave the AGPRs 8 at a time, not all 256 up front
                    f32x4 v0 = acc[2 * ip][j] + bv[2 * ip];
                    f32x4 v1 = acc[2 * ip + 1][j] + bv[2 * ip + 1];
                    if (EPI == 1) {
                        const u32x4 r = res[j][ip];
                        v0.x += bf16_to_f32((u16)(r.x & 0xffff));
                        v0.y += bf16_to_f32((u16)(r.x >> 16));
                        v0.z += bf16_to_f32((u16)(r.y & 0xffff));
                        v0.w += bf16_to_f32((u16)(r.y >> 16));
                        v1.x += bf16_to_f32((u16)(r.z & 0xffff));
                        v1.y += bf16_to_f32((u16)(r.z >> 16));
                        v1.z += bf16_to_f32((u16)(r.w & 0xffff));
                        v1.w += bf16_to_f32((u16)(r.w >> 16));
                    }
                    if (EPI == 2) {
                        v0.x = gelu_erf(v0.x); v0.y = gelu_erf(v0.y); v0.z = gelu_erf(v0.z); v0.w = gelu_erf(v0.w);
                        v1.x = gelu_erf(v1.x); v1.y = gelu_erf(v1.y); v1.z = gelu_erf(v1.z); v1.w = gelu_erf(v1.w);
                    }
                    uint4 o;
                    o.x = (unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16);
                    o.y = (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16);
                    o.z = (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16);
                    o.w = (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16);
                    if (m < M) *reinterpret_cast<uint4*>(yp + 8 * ip) = o;
                }
            }
        }
        if (!has_next) break;
        tile = next;
    }
}

template <int EPI>
static hipError_t launch_w4l(const u16* X, const u16* W, const float* bias, const u16* residual, u16* Y, int M,
                             int M_pad, int N, int K, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_w4l_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kW4LLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0)
            n_cus = 256;
    }
    const int tiles_total = (N / RBN) * (M_pad / RBM);
    const int grid = tiles_total < n_cus ? tiles_total : n_cus;
    hipLaunchKernelGGL((gemm_bf16_w4l_kernel<EPI>), dim3(grid), dim3(kW4Threads), kW4LLdsBytes, stream, X, W, bias,
                       residual, Y, M, N, K, tiles_total);
    return hipGetLastError();
}

// variant: "ring" | "pring" | "p64" | "w4l" | "p5" (the product kernel, for comparison)
template <int EPI>
static hipError_t launch_retired_epi(const char* variant, const u16* X, const u16* W, const float* bias, const u16* residual,
                                     u16* Y, int M, int M_pad, int N, int K, hipStream_t stream) {
    if (strcmp(variant, "ring") == 0) return launch_ring<EPI>(X, W, bias, residual, Y, M, M_pad, N, K, stream);
    if (strcmp(variant, "pring") == 0) return launch_pring<EPI>(X, W, bias, residual, Y, M, M_pad, N, K, stream);
    if (strcmp(variant, "p64") == 0) return launch_p64<EPI>(X, W, bias, residual, Y, M, M_pad, N, K, stream);
    if (strcmp(variant, "w4l") == 0) return launch_w4l<EPI>(X, W, bias, residual, Y, M, M_pad, N, K, stream);
    return launch_p5<EPI>(X, W, bias, residual, Y, M, M_pad, N, K, stream);
}

hipError_t launch_retired(const char* variant, const void* X, const void* W, const float* bias, const void* residual, void* Y,
                          int M, int M_pad, int N, int K, int epilogue, hipStream_t stream) {
    const u16* x = static_cast<const u16*>(X);
    const u16* w = static_cast<const u16*>(W);
    const u16* r = static_cast<const u16*>(residual);
    u16* y = static_cast<u16*>(Y);
    if (epilogue == 0) return launch_retired_epi<0>(variant, x, w, bias, r, y, M, M_pad, N, K, stream);
    if (epilogue == 1) return launch_retired_epi<1>(variant, x, w, bias, r, y, M, M_pad, N, K, stream);
    return launch_retired_epi<2>(variant, x, w, bias, r, y, M, M_pad, N, K, stream);
}

}  // namespace rass

#ifdef RASS_RETIRED_WITH_MAIN
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void retired_fill_bf16(unsigned short* x, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((float)(h & 0xffff) - 32768.f) * (1.f / 32768.f);
        x[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}

int main() {
    const int M = 131072;   // batch 256 x 512 tokens
    const int shapes[4][3] = {{3072, 1024, 0}, {1024, 1024, 1}, {4096, 1024, 2}, {1024, 4096, 1}};
    const char* variants[5] = {"p5", "p64", "pring", "ring", "w4l"};
    for (const int* sh : shapes) {
        const int N = sh[0], K = sh[1], epi = sh[2];
        unsigned short *X, *W, *R, *Y, *Yref; float* b;
        CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&R, (size_t)M * N * 2));
        CK(hipMalloc(&Y, (size_t)M * N * 2)); CK(hipMalloc(&Yref, (size_t)M * N * 2)); CK(hipMalloc(&b, N * 4));
        retired_fill_bf16<<<4096, 256>>>(X, (size_t)M * K, 1); retired_fill_bf16<<<1024, 256>>>(W, (size_t)N * K, 2);
        retired_fill_bf16<<<4096, 256>>>(R, (size_t)M * N, 3); CK(hipMemset(b, 0, N * 4));
        CK(hipDeviceSynchronize());
        std::vector<unsigned short> ref((size_t)M * N), got((size_t)M * N);
        for (const char* v : variants) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms = 0, best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                CK(hipEventRecord(e0, 0));
                CK(rass::launch_retired(v, X, W, b, R, v == variants[0] ? Yref : Y, M, M, N, K, epi, 0));
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            size_t bad = 0;
            if (v == variants[0]) {
                CK(hipMemcpy(ref.data(), Yref, ref.size() * 2, hipMemcpyDeviceToHost));
            } else {
                CK(hipMemcpy(got.data(), Y, got.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < got.size(); ++i) {   // same fp32 sum in another order: a few bf16 ulps at most
                    const int d = (int)got[i] - (int)ref[i];
                    bad += (d > 2 || d < -2) && ((got[i] ^ ref[i]) & 0x8000) == 0;
                }
            }
            printf("N=%d K=%d epi=%d  %-5s %8.1f us  %7.1f TF/s  elements off p5 by > 2 bf16 ulp: %zu\n", N, K, epi, v, best * 1e3,
                   2.0 * M * N * K / (best * 1e-3) / 1e12, bad);
        }
        (void)hipFree(X); (void)hipFree(W); (void)hipFree(R); (void)hipFree(Y); (void)hipFree(Yref); (void)hipFree(b);
    }
    return 0;
}
#endif
