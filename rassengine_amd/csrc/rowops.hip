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
                                                                     int64_t out_stride, int64_t n, int dim,
                                                                     int64_t n_total) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // rows [n, n_total) are zero-filled (query padding up to a multiple of 16 rows)
    for (int64_t r = n + (int64_t)blockIdx.x * 4 + wave; r < n_total; r += (int64_t)gridDim.x * 4) {
        float* dst = out + r * out_stride;
        for (int c = lane; c < (int)out_stride; c += 64) dst[c] = 0.f;
    }
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
                                     int64_t n, int dim, hipStream_t stream, int64_t n_total) {
    if (n_total < n) n_total = n;
    if (n_total <= 0) return hipSuccess;
    int64_t blocks = (n_total + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)blocks), dim3(kRowThreads), 0, stream, in, in_stride,
                       out, out_stride, n, dim, n_total);
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

__global__ void iota_i64_kernel(int64_t* dst, int64_t n, int64_t base) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = base + i;
}

hipError_t launch_iota_i64(int64_t* dst, int64_t n, int64_t base, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iota_i64_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, n, base);
    return hipGetLastError();
}

// Row tags handed over from device memory cannot be validated on the host without a sync: negative
// values (the tombstone code is reserved for rass_index_delete, which also keeps the counters) are
// stored as 0 = "no patient".
__global__ void copy_tags_clamped_kernel(int32_t* dst, const int32_t* src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t t = src[i];
        dst[i] = t < 0 ? 0 : t;
    }
}

hipError_t launch_copy_tags_clamped(int32_t* dst, const int32_t* src, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_tags_clamped_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, src, n);
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

// ---------------------------------------------------------------------------- tile16 layout
// (kernels.h) One wave owns one 16-row block: lane (m = lane&15, g = lane>>4) handles row
// 16b+m, columns 16j + 4g .. +3 of every chunk j, so each chunk is written / read as one
// coalesced 1 KiB burst in exactly the order the scan kernel's MFMA A operand wants.

__device__ __forceinline__ float block_row_sum(float v) {
    // the 4 lanes that share a row are lane, lane^16, lane^32, lane^48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ f32x4 load_row_piece(const float* src, int c, int dim, bool vec) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (vec && c + 3 < dim) {
        v = *reinterpret_cast<const f32x4*>(src + c);
    } else {
        if (c < dim) v.x = src[c];
        if (c + 1 < dim) v.y = src[c + 1];
        if (c + 2 < dim) v.z = src[c + 2];
        if (c + 3 < dim) v.w = src[c + 3];
    }
    return v;
}

// rows [first_row, first_row + n) of the packed slab <- in[0..n) (row-major), optionally
// L2-normalised with the reference formula (app/main.py:1249-1251).  Rows of a touched
// block outside that range are left as they are.
__global__ __launch_bounds__(kRowThreads) void pack_rows_tile16_kernel(const float* __restrict__ in,
                                                                       int64_t in_stride,
                                                                       float* __restrict__ packed, int64_t stride,
                                                                       int64_t first_row, int64_t n, int dim,
                                                                       int normalize) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nchunks = (int)(stride >> 4);
    const int64_t b0 = first_row >> 4, b1 = (first_row + n + 15) >> 4;
    const bool vec = ((in_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    for (int64_t b = b0 + (int64_t)blockIdx.x * 4 + wave; b < b1; b += (int64_t)gridDim.x * 4) {
        const int64_t row = b * 16 + m;
        const bool valid = row >= first_row && row < first_row + n;
        const float* src = in + (valid ? (row - first_row) : 0) * in_stride;
        float denom = 1.f;
        if (normalize) {
            float ss = 0.f;
            for (int j = 0; j < nchunks; ++j) {
                const f32x4 v = valid ? load_row_piece(src, 16 * j + 4 * g, dim, vec) : f32x4{0.f, 0.f, 0.f, 0.f};
                ss = fmaf(v.x, v.x, ss);
                ss = fmaf(v.y, v.y, ss);
                ss = fmaf(v.z, v.z, ss);
                ss = fmaf(v.w, v.w, ss);
            }
            denom = sqrtf(block_row_sum(ss)) + 1e-9f;
        }
        float* dst = packed + b * 16 * stride + lane * 4;
        for (int j = 0; j < nchunks; ++j) {
            if (valid) {
                f32x4 v = load_row_piece(src, 16 * j + 4 * g, dim, vec);
                if (normalize) {
                    v.x = v.x / denom;
                    v.y = v.y / denom;
                    v.z = v.z / denom;
                    v.w = v.w / denom;
                }
                *reinterpret_cast<f32x4*>(dst + j * 256) = v;
            }
        }
    }
}

hipError_t launch_pack_rows_tile16(const float* in, int64_t in_stride, float* packed, int64_t stride,
                                   int64_t first_row, int64_t n, int dim, int normalize, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int64_t nblk = ((first_row + n + 15) >> 4) - (first_row >> 4);
    int64_t blocks = (nblk + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pack_rows_tile16_kernel, dim3((unsigned)blocks), dim3(kRowThreads), 0, stream, in, in_stride,
                       packed, stride, first_row, n, dim, normalize);
    return hipGetLastError();
}

// out[0..n) (row-major, out_stride) <- rows [first_row, first_row + n) of the packed slab
__global__ __launch_bounds__(kRowThreads) void unpack_rows_tile16_kernel(const float* __restrict__ packed,
                                                                         int64_t stride, int64_t first_row,
                                                                         int64_t n, int dim, float* __restrict__ out,
                                                                         int64_t out_stride) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nchunks = (int)(stride >> 4);
    const int64_t b0 = first_row >> 4, b1 = (first_row + n + 15) >> 4;
    for (int64_t b = b0 + (int64_t)blockIdx.x * 4 + wave; b < b1; b += (int64_t)gridDim.x * 4) {
        const int64_t row = b * 16 + m;
        const bool valid = row >= first_row && row < first_row + n;
        const float* src = packed + b * 16 * stride + lane * 4;
        float* dst = out + (valid ? (row - first_row) : 0) * out_stride;
        for (int j = 0; j < nchunks; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + j * 256);
            const int c = 16 * j + 4 * g;
            if (valid) {
                if (c < dim) dst[c] = v.x;
                if (c + 1 < dim) dst[c + 1] = v.y;
                if (c + 2 < dim) dst[c + 2] = v.z;
                if (c + 3 < dim) dst[c + 3] = v.w;
            }
        }
    }
}

hipError_t launch_unpack_rows_tile16(const float* packed, int64_t stride, int64_t first_row, int64_t n, int dim,
                                     float* out, int64_t out_stride, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int64_t nblk = ((first_row + n + 15) >> 4) - (first_row >> 4);
    int64_t blocks = (nblk + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(unpack_rows_tile16_kernel, dim3((unsigned)blocks), dim3(kRowThreads), 0, stream, packed,
                       stride, first_row, n, dim, out, out_stride);
    return hipGetLastError();
}

// out[i] (row-major, out_stride) <- row row_ids[i] of the packed slab, i < n: one workgroup per row, a thread per
// (16-column chunk, 4-column group).  The k-means seeding picks nlist scattered sample rows: as single-row unpack launches
// that was 38 us per row (one wave walking 64 chunks), 0.16 s of a 1.4-s training at 4 096 lists.
__global__ __launch_bounds__(kRowThreads) void gather_rows_tile16_kernel(const float* __restrict__ packed, int64_t stride,
                                                                         const int64_t* __restrict__ row_ids, int64_t n_rows,
                                                                         int dim, float* __restrict__ out, int64_t out_stride) {
    const int64_t i = blockIdx.x;
    const int64_t row = row_ids[i];
    if (row < 0 || row >= n_rows) return;   // (the caller clamps; an out-of-range id leaves its output row untouched)
    const int nchunks = (int)(stride >> 4);
    const float* src = packed + (row >> 4) * 16 * stride + (row & 15) * 4;
    float* dst = out + i * out_stride;
    for (int t = threadIdx.x; t < nchunks * 4; t += kRowThreads) {
        const int j = t >> 2, g = t & 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)j * 256 + g * 64);
        const int c = 16 * j + 4 * g;
        if (c < dim) dst[c] = v.x;
        if (c + 1 < dim) dst[c + 1] = v.y;
        if (c + 2 < dim) dst[c + 2] = v.z;
        if (c + 3 < dim) dst[c + 3] = v.w;
    }
}

hipError_t launch_gather_rows_tile16(const float* packed, int64_t stride, const int64_t* row_ids, int64_t n, int64_t n_rows,
                                     int dim, float* out, int64_t out_stride, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_tile16_kernel, dim3((unsigned)n), dim3(kRowThreads), 0, stream, packed, stride, row_ids,
                       n_rows, dim, out, out_stride);
    return hipGetLastError();
}

// Synthetic unit rows [first_row, first_row + n) written straight into the packed slab.
__global__ __launch_bounds__(kRowThreads) void fill_synthetic_kernel(float* __restrict__ packed, int64_t stride,
                                                                     int64_t first_row, int64_t n, int dim,
                                                                     uint64_t seed, int64_t row_id_base) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nchunks = (int)(stride >> 4);
    const int64_t b0 = first_row >> 4, b1 = (first_row + n + 15) >> 4;
    for (int64_t b = b0 + (int64_t)blockIdx.x * 4 + wave; b < b1; b += (int64_t)gridDim.x * 4) {
        const int64_t row = b * 16 + m;
        const bool valid = row >= first_row && row < first_row + n;
        const int64_t gid = row_id_base + row;
        float ss = 0.f;
        for (int j = 0; j < nchunks; ++j) {
            const int c = 16 * j + 4 * g;
            if (c < dim) {
                const f32x4 v = normals4(seed, gid, c >> 2);
                ss = fmaf(v.x, v.x, ss);
                if (c + 1 < dim) ss = fmaf(v.y, v.y, ss);
                if (c + 2 < dim) ss = fmaf(v.z, v.z, ss);
                if (c + 3 < dim) ss = fmaf(v.w, v.w, ss);
            }
        }
        const float denom = sqrtf(block_row_sum(ss)) + 1e-9f;
        float* dst = packed + b * 16 * stride + lane * 4;
        for (int j = 0; j < nchunks; ++j) {
            const int c = 16 * j + 4 * g;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < dim) {
                const f32x4 x = normals4(seed, gid, c >> 2);
                v.x = x.x / denom;
                if (c + 1 < dim) v.y = x.y / denom;
                if (c + 2 < dim) v.z = x.z / denom;
                if (c + 3 < dim) v.w = x.w / denom;
            }
            if (valid) *reinterpret_cast<f32x4*>(dst + j * 256) = v;
        }
    }
}

hipError_t launch_fill_synthetic_f32(float* packed, int64_t stride, int64_t first_row, int64_t n, int dim,
                                     uint64_t seed, int64_t row_id_base, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int64_t nblk = ((first_row + n + 15) >> 4) - (first_row >> 4);
    int64_t blocks = (nblk + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(fill_synthetic_kernel, dim3((unsigned)blocks), dim3(kRowThreads), 0, stream, packed, stride,
                       first_row, n, dim, seed, row_id_base);
    return hipGetLastError();
}

}  // namespace rass
