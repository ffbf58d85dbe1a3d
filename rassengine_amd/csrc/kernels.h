// kernels.h — internal launcher interface between the C-ABI layer (api.hip) and the
// gfx950 kernels.  Not part of the public ABI (include/rass_engine.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rass {

struct ScanArgs {
    const float* corpus;      // [n_rows][row_stride] fp32, rows L2-normalised, zero padded past dim
    const int32_t* row_tag;   // [n_rows] or nullptr; -1 = tombstone, >= 0 = patientId code
    const float* q_padded;    // [16*NT][row_stride] normalised queries, zero rows past nq
    const int32_t* q_filter;  // [nq] or nullptr; -1 = no filter
    float* part_scores;       // [grid][nq][k]
    int64_t* part_ids;        // [grid][nq][k]
    int64_t row_stride;       // elements, multiple of 128
    int64_t id_base;          // added to local row ids
    int n_rows;
    int nq;
    int k;
};

bool scan_supported_stride(int64_t row_stride);
hipError_t launch_scan_topk_f32(const ScanArgs& a, int grid, hipStream_t stream);

// [n_lists][nq][k] sorted candidate lists -> [nq][k]; n_lists * k <= kMergeMaxCandidates.
constexpr int kMergeMaxCandidates = 8192;
hipError_t launch_merge_topk(const float* scores, const int64_t* ids, int n_lists, int nq, int k,
                             float* out_scores, int64_t* out_ids, hipStream_t stream);

// out[r][0..dim) = in[r] / (||in[r]|| + 1e-9); out[r][dim..out_stride) = 0.
hipError_t launch_normalize_rows_f32(const float* in, int64_t in_stride, float* out, int64_t out_stride,
                                     int64_t n, int dim, hipStream_t stream);

// Zero `n_pad_rows` rows of `stride` floats starting at `dst` (query padding).
hipError_t launch_zero_rows(float* dst, int64_t stride, int n_rows, hipStream_t stream);

// Synthetic corpus: rows [0,n) of out get iid N(0,1) from Philox4x32-10 keyed by
// (seed, row_id_base + r, col), then L2-normalised in place; padding zeroed.
hipError_t launch_fill_synthetic_f32(float* out, int64_t stride, int64_t n, int dim, uint64_t seed,
                                     int64_t row_id_base, hipStream_t stream);

hipError_t launch_fill_i32(int32_t* dst, int64_t n, int32_t value, hipStream_t stream);

}  // namespace rass
