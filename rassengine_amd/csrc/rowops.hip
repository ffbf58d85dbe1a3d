// rowops.hip — row-wise helpers around the scan: the reference's L2 normalise (a4), query
// padding, tag fills and the on-device synthetic corpus generator of SURVEY §8d.
//
// All of them are HBM-bound one-wave-per-row kernels: 16 B per lane coalesced accesses, a
// wave-level butterfly for the row reduction, no LDS.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace rass {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowThreads = 256;  // 4 waves = 4 rows per block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Reference app/main.py:1249-1251 / 1536-1537:
//   norms = np.linalg.norm(e, axis=1, keepdims=True); e = e / (norms + 1e-9)
// fp32 throughout; IEEE sqrt and divide (hipcc default: correctly rounded).  Only the
// order of the sum of squares differs from numpy's pairwise sum (<= a few ulp of the norm).
__global__ __launch_bounds__(kRowThreads) void normalize_rows_kernel(const float* __restrict__ in,
                                                                     int64_t in_stride, float* __restrict__ out,
                                                                     int64_t out_stride, int64_t n, int dim) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool vec = ((dim & 3) == 0) && ((in_stride & 3) == 0) && ((out_stride & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const float* src = in + r * in_stride;
        float* dst = out + r * out_stride;
        float ss = 0.f;
        if (vec) {
            for (int c = lane * 4; c < dim; c += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
                ss = fmaf(v.x, v.x, ss);
                ss = fmaf(v.y, v.y, ss);
                ss = fmaf(v.z, v.z, ss);
                ss = fmaf(v.w, v.w, ss);
            }
        } else {
            for (int c = lane; c < dim; c += 64) ss = fmaf(src[c], src[c], ss);
        }
        ss = wave_sum(ss);
        const float denom = sqrtf(ss) + 1e-9f;
        if (vec) {
            for (int c = lane * 4; c < (int)out_stride; c += 256) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c < dim) {
                    v = *reinterpret_cast<const f32x4*>(src + c);
                    v.x = v.x / denom;
                    v.y = v.y / denom;
                    v.z = v.z / denom;
                    v.w = v.w / denom;
                }
                *reinterpret_cast<f32x4*>(dst + c) = v;
            }
        } else {
            for (int c = lane; c < (int)out_stride; c += 64) dst[c] = (c < dim) ? src[c] / denom : 0.f;
        }
    }
}

hipError_t launch_normalize_rows_f32(const float* in, int64_t in_stride, float* out, int64_t out_stride,
                                     int64_t n, int dim, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)blocks), dim3(kRowThreads), 0, stream, in, in_stride,
                       out, out_stride, n, dim);
    return hipGetLastError();
}

__global__ void zero_rows_kernel(float* dst, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = 0.f;
}

hipError_t launch_zero_rows(float* dst, int64_t stride, int n_rows, hipStream_t stream) {
    const int64_t total = stride * n_rows;
    if (total <= 0) return hipSuccess;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, total);
    return hipGetLastError();
}

__global__ void fill_i32_kernel(int32_t* dst, int64_t n, int32_t value) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = value;
}

hipError_t launch_fill_i32(int32_t* dst, int64_t n, int32_t value, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, n, value);
    return hipGetLastError();
}

// ---- Philox4x32-10 (Salmon et al., SC'11), counter = (row_lo, row_hi, col/4, 0), key = seed.
struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ f32x4 normals4(uint64_t seed, int64_t row, int chunk) {
    const U4 r = philox4x32_10(U4{(uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)chunk, 0u},
                               (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(r.w >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca, sb, cb;
    sincosf(6.28318530717958647692f * u1, &sa, &ca);
    sincosf(6.28318530717958647692f * u3, &sb, &cb);
    return f32x4{ra * ca, ra * sa, rb * cb, rb * sb};
}

__global__ __launch_bounds__(kRowThreads) void fill_synthetic_kernel(float* __restrict__ out, int64_t stride,
                                                                     int64_t n, int dim, uint64_t seed,
                                                                     int64_t row_id_base) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const int64_t gid = row_id_base + r;
        float ss = 0.f;
        for (int c = lane * 4; c < dim; c += 256) {
            const f32x4 v = normals4(seed, gid, c >> 2);
            ss = fmaf(v.x, v.x, ss);
            if (c + 1 < dim) ss = fmaf(v.y, v.y, ss);
            if (c + 2 < dim) ss = fmaf(v.z, v.z, ss);
            if (c + 3 < dim) ss = fmaf(v.w, v.w, ss);
        }
        ss = wave_sum(ss);
        const float denom = sqrtf(ss) + 1e-9f;
        float* dst = out + r * stride;
        for (int c = lane * 4; c < (int)stride; c += 256) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < dim) {
                const f32x4 g = normals4(seed, gid, c >> 2);
                v.x = g.x / denom;
                if (c + 1 < dim) v.y = g.y / denom;
                if (c + 2 < dim) v.z = g.z / denom;
                if (c + 3 < dim) v.w = g.w / denom;
            }
            *reinterpret_cast<f32x4*>(dst + c) = v;
        }
    }
}

hipError_t launch_fill_synthetic_f32(float* out, int64_t stride, int64_t n, int dim, uint64_t seed,
                                     int64_t row_id_base, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(fill_synthetic_kernel, dim3((unsigned)blocks), dim3(kRowThreads), 0, stream, out, stride, n,
                       dim, seed, row_id_base);
    return hipGetLastError();
}

}  // namespace rass
