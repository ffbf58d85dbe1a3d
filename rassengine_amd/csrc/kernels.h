// kernels.h — internal launcher interface between the C-ABI layer (api.hip) and the
// gfx950 kernels.  Not part of the public ABI (include/rass_engine.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rass {

constexpr int kMaxSampleGroups = 256;  // workgroups of a sample pass (ScanArgs::sample_best)

struct ScanArgs {
    const float* corpus;      // tile16-packed fp32 slab (see below), rows L2-normalised, zero padded past dim
    const int32_t* row_tag;   // [n_rows] or nullptr; -1 = tombstone, >= 0 = patientId code
    const float* q_padded;    // [16*NT][row_stride] normalised queries, zero rows past nq
    const int32_t* q_filter;  // [nq] or nullptr; -1 = no filter
    // Extended per-query filters (all nullptr for the plain scan; any non-null selects the EXT kernel variant):
    //   q_filter_mask[q]: a row matches when (tag & mask) == q_filter[q] (tag = patient code | doc_type << 24,
    //                     so one compare serves `term: patientId`, `term: doc_type` or both); nullptr = exact compare
    //   q_after_score/q_after_id[q]: continuation bound of a multi-pass top-k (k > 32): only rows that rank
    //                     strictly AFTER (score, global id) under (score desc, id asc) are eligible
    const int32_t* q_filter_mask = nullptr;
    const float* q_after_score = nullptr;
    const int64_t* q_after_id = nullptr;
    float* part_scores;       // [grid][nq][k]
    int64_t* part_ids;        // [grid][nq][k]
    int64_t row_stride;       // elements, multiple of 128
    int64_t id_base;          // added to local row ids
    int n_rows;
    int nq;
    int k;
    // IVF probe plan (all nullptr for the flat scan): work item i = slab tile work_tile[i] with
    // work_rows[i] valid rows, visible to the queries whose bit is set in work_mask[i]
    const int32_t* work_tile = nullptr;
    const int32_t* work_rows = nullptr;
    const uint32_t* work_mask = nullptr;
    const int32_t* n_work = nullptr;  // device scalar: number of work items
    // cross-index batch (all nullptr otherwise): item i scans tile work_tile[i] of the slab work_base[i] with the
    // row tags work_tags[i] (may be null per item); ids are rows of THAT slab
    const float* const* work_base = nullptr;
    const int32_t* const* work_tags = nullptr;
    // The sample floor (nullptr = none).  sample_best[32][kMaxSampleGroups] holds, for g < sample_groups, the best
    // score per query that workgroup g of a SAMPLE PASS found (the same scan over a small prefix of the slab, run
    // first with sample_pass = true, which writes exactly this array through part_scores).  Each wave of the big
    // scan takes, per query, the k-th largest of those as a floor: k different rows reach it, hence so does the
    // final k-th best, and rows scoring below it are dropped before the sorted insertion.  That insertion is where
    // the 32-query scan spends its non-MFMA time: every workgroup list otherwise takes ~k(1 + ln(rows/k))
    // insertions, ~70 at k = 10 over 3,900 rows, most of them in the first iterations.  Results do not depend on
    // the floor (rows tying with it are kept).
    const float* sample_best = nullptr;
    int sample_groups = 0;
    bool sample_pass = false;  // this launch IS the sample pass (flat, > 16 queries): same code, its own kernel name
    // XCD skew (0 = plain round-robin).  Workgroups land on XCD blockIdx % 8; measured on MI355X
    // (scripts/microbench/scan_tail.hip) the odd XCDs stream ~14 % slower than the even ones when
    // the scan is purely HBM-bound (B <= 16).  With skew s > 0 the even workgroups take s+1 items
    // for every s the odd ones take, so all of them finish together.  Needs an even grid.
    int xcd_skew = 0;
    // Grouped flat scan (kFlatGroups; 0 = off): workgroups [g * wgs_per_group, (g + 1) * wgs_per_group) serve launch group
    // g — its 32 (zero-padded) queries at q_padded + g * q_group_stride, its lists at part_* + g * part_group_stride
    // ([wgs_per_group][nq][k], nq = 32 for every group), its filters at q_filter + 32 g.  No sample floor, no EXT.
    int wgs_per_group = 0;
    int64_t q_group_stride = 0, part_group_stride = 0;
    // Grouped IVF fine scan (kIvfGroups; work_tile != nullptr and wgs_per_group > 0): group g walks the work list at
    // work_* + g * work_group_stride with n_work[g] items; it has min(32, nq_total - 32 g) queries and writes dense lists
    // [wgs_per_group][that many][k] at part_* + g * part_group_stride.
    int64_t work_group_stride = 0;
    int nq_total = 0;
};

bool scan_supported_stride(int64_t row_stride);
hipError_t launch_scan_topk_f32(const ScanArgs& a, int grid, hipStream_t stream);

// [n_lists][nq][k] sorted candidate lists -> [nq][k]; n_lists * k <= kMergeMaxCandidates.
constexpr int kMergeMaxCandidates = 8192;
// `groups` (optional): ONE launch merges the lists of several launch groups of <= size queries each; nq is then
// the total query count, blockIdx / size the group, and every pointer advances by its *_stride per group (in
// elements).  lists_are_dense: a group's lists are [n_lists][nq_g][k] with nq_g its own query count (the scan's
// per-workgroup lists); otherwise the caller's list strides hold for every group (gathered packed records).
struct MergeGroups {
    int size = 0;  // 0 = ungrouped
    int nq_total = 0;
    bool lists_are_dense = false;
    int64_t score_stride = 0, id_stride = 0, out_score_stride = 0, out_id_stride = 0;
};
hipError_t launch_merge_topk(const float* scores, const int64_t* ids, int n_lists, int nq, int k,
                             float* out_scores, int64_t* out_ids, hipStream_t stream,
                             const int64_t* id_map = nullptr, int64_t score_list_stride = 0,
                             int64_t id_list_stride = 0, const MergeGroups* groups = nullptr);

// ---- bf16 candidate scan + exact re-rank (scan_bf16.hip)
struct ScanBf16Args {
    const unsigned short* corpus;   // tile16b bf16 slab
    const int32_t* row_tag;
    const unsigned short* q_bf16;   // [16*NT][row_stride] normalised queries as bf16
    const int32_t* q_filter;
    float* part_scores;             // [grid][nq][k]
    int64_t* part_ids;              // [grid][nq][k]  LOCAL rows
    int64_t row_stride;             // elements, multiple of 256
    int n_rows;
    int nq;
    int k;                          // candidates kept per query (<= 32)
    int64_t id_base = 0;            // added to the reported rows (0 for the prefilter's candidate scan)
    // EXT variant (as ScanArgs): masked tag compare and the continuation bound of a k > 32 pass
    const int32_t* q_filter_mask = nullptr;
    const float* q_after_score = nullptr;
    const int64_t* q_after_id = nullptr;
    // IVF probe plan over a bf16 slab (all nullptr for the flat scan; as ScanArgs, in 64-row tiles): work item i = slab
    // tile work_tile[i] with work_rows[i] valid rows, visible to the queries whose bit is set in work_mask[i]
    const int32_t* work_tile = nullptr;
    const int32_t* work_rows = nullptr;
    const uint32_t* work_mask = nullptr;
    const int32_t* n_work = nullptr;
    // the sample floor of the flat scan (as ScanI8Args): part_scores [sample_groups][nq][1] of a SAMPLE launch — this kernel with
    // k = 1 over the slab's first 64 * sample_groups rows, same queries and filters; nullptr = none
    const float* sample_best = nullptr;
    int sample_groups = 0;
};
hipError_t launch_scan_bf16_topk(const ScanBf16Args& a, int grid, hipStream_t stream);
// fp32 tile16 blocks -> bf16 tile16b blocks [block0, block1) of dst.  src_block0 (default = block0): the source
// block that lands in block0 (a staging slab); only destination rows in [row_lo, row_hi) are written.
hipError_t launch_convert_tile16_bf16(const float* src, void* dst, int64_t stride, int64_t block0, int64_t block1,
                                      hipStream_t stream, int64_t src_block0 = -1, int64_t row_lo = 0,
                                      int64_t row_hi = -1);
hipError_t launch_unpack_rows_tile16b(const void* slab, int64_t stride, int64_t first_row, int64_t n, int dim,
                                      float* out, int64_t out_stride, hipStream_t stream);
hipError_t launch_queries_to_bf16(const float* src, void* dst, int64_t n, hipStream_t stream);
// out_*_group_stride (elements; 0 = contiguous [nq][k]): query q's results go to out + (q / 32) * group_stride + (q % 32) * k
hipError_t launch_rerank_f32(const float* slab, int64_t stride, const float* q_padded, const int64_t* cand_rows, int nq,
                             int n_cand, int k, int64_t id_base, float* out_scores, int64_t* out_ids,
                             hipStream_t stream, int64_t out_scores_group_stride = 0, int64_t out_ids_group_stride = 0,
                             const int64_t* id_map = nullptr);   // id_map: reported id (and tie order) of slab row r = id_map[r]

// ---- int8 candidate scan of the prefilter mode (scan_i8.hip, SURVEY §8f-4 "or int8")
struct ScanI8Args {
    const signed char* corpus;   // tile16i int8 slab
    const float* row_scale;      // [n_rows] max|x| / 127 of every row
    const int32_t* row_tag;      // [n_rows] or nullptr
    const signed char* q_i8;     // [16*NT][row_stride] quantised queries, row-major
    const int32_t* q_filter;     // [nq] or nullptr; -1 = no filter (exact tag compare)
    float* part_scores;          // [grid][nq][k]  (float)(integer dot) * row scale
    int64_t* part_ids;           // [grid][nq][k]  LOCAL rows
    int64_t row_stride;          // bytes per row of the int8 slab: 512 or 1024
    int n_rows;
    int nq;
    int k;                       // candidates kept per query (<= 32)
    // the sample floor (nullptr = none): part_scores [sample_groups][nq][1] of a SAMPLE launch — this kernel with k = 1 over the
    // slab's first 64 * sample_groups rows, same queries and filters; see scan_i8.hip
    const float* sample_best = nullptr;
    int sample_groups = 0;
    const int32_t* q_filter_mask = nullptr;   // [nq] or nullptr: a row matches when (tag & mask) == q_filter
    // grouped launch (flat scans; 0 = off): workgroups [g * wgs_per_group, (g + 1) * wgs_per_group) serve launch group g — its 32
    // queries at q_i8 + g * q_group_stride (bytes), its filters at q_filter + 32 g, its lists at part_scores + g *
    // part_group_stride (elements; part_ids likewise if given).  Every group has nq queries.  Used for the sample launches of a
    // batch call (one launch instead of one per group).
    int wgs_per_group = 0;
    int64_t q_group_stride = 0, part_group_stride = 0;
    // IVF probe plan over an int8 slab (all nullptr for the flat scan; as ScanBf16Args, 64-row tiles)
    const int32_t* work_tile = nullptr;
    const int32_t* work_rows = nullptr;
    const uint32_t* work_mask = nullptr;
    const int32_t* n_work = nullptr;
};
hipError_t launch_scan_i8_topk(const ScanI8Args& a, int grid, hipStream_t stream);
// fp32 tile16 blocks [block0, block1) -> tile16i blocks of dst (+ one scale per row)
hipError_t launch_quantize_tile16_i8(const float* src, void* dst, float* scale, int64_t stride, int64_t stride_i8, int64_t block0,
                                     int64_t block1, hipStream_t stream);
hipError_t launch_queries_to_i8(const float* src, void* dst, int nq_pad, int64_t stride, int64_t stride_i8, hipStream_t stream);

// ---- peer-store exchange of per-shard top-k (peer.hip)
hipError_t launch_peer_post(const void* local, size_t bytes, void* remote_slot, void* remote_flag, uint64_t seq,
                            hipStream_t stream);
hipError_t launch_peer_wait(const void* flags, int n, int flag_stride_bytes, uint64_t seq, int* status,
                            int64_t max_spins, hipStream_t stream);

// ---- k-means of the IVF build (kmeans.hip)
struct AssignArgs {
    const float* rows;       // tile16 slab holding the rows to assign (normalised)
    const float* centroids;  // tile16 slab of nlist normalised centroids (whole 16-row blocks)
    int32_t* assign;         // [n_blocks * 32] list of row (b, r) at b*32 + r (compact over the processed blocks)
    float* best;             // optional [n_blocks * 32]: the winning cosine
    int64_t row_stride;      // elements, 128 * {1..8}
    int64_t slab_rows;       // rows ALLOCATED in `rows` (multiple of 16): halves past it read as zero
    int64_t first_block;     // first 32-row block processed ...
    int64_t block_step;      // ... and the stride between processed blocks (a strided training sample)
    int n_blocks;
    int nlist;
};
hipError_t launch_kmeans_assign_f32(const AssignArgs& a, int n_cus, hipStream_t stream);
// sums[assign][0..dim) += row, counts[assign] += 1 over the processed blocks; rows >= n_valid are skipped
hipError_t launch_kmeans_accumulate(const float* rows, int64_t stride, int64_t first_block, int64_t block_step,
                                    int n_blocks, int64_t n_valid, const int32_t* assign, float* sums, float* counts,
                                    int dim, int nlist, hipStream_t stream);

// ---- IVF (ivf.hip)
hipError_t launch_plan_probe(const int64_t* probe_ids, int nq, int nprobe, int nlist, const int32_t* list_tile0,
                             const int32_t* list_len, int32_t* work_tile, int32_t* work_rows, uint32_t* work_mask,
                             int32_t* n_work, int64_t* scanned_rows, hipStream_t stream,
                             const uint32_t* preset_mask = nullptr, int tile_rows = 32);
// The plan of a whole batch of launch groups straight from the grouped coarse scan's per-workgroup lists (ivf.hip).
hipError_t launch_plan_probe_groups(const float* cpart_scores, const int64_t* cpart_ids, int n_clists, int nprobe, int groups,
                                    int nq_total, int64_t cpart_group_stride, int nlist, const int32_t* list_tile0,
                                    const int32_t* list_len, int32_t* work_tile, int32_t* work_rows, uint32_t* work_mask,
                                    int64_t work_cap, int32_t* n_work, int64_t* scanned_rows, hipStream_t stream,
                                    int tile_rows);
hipError_t launch_ivf_threshold(const float* part_scores, const int64_t* part_ids, int n_ctiles, int nq, int nprobe,
                                uint32_t* tau_key, hipStream_t stream);
hipError_t launch_ivf_mask_from_scores(const float* part_scores, const int64_t* part_ids, int n_ctiles, int nq,
                                       int nlist, const uint32_t* tau_key, uint32_t* mask, hipStream_t stream);
hipError_t launch_permute_rows_tile16(const float* src, float* dst, int64_t stride, const int64_t* src_of,
                                      int64_t dst_rows, hipStream_t stream);
// the same rows rounded to bf16 into a tile16b slab (the IVF over a bf16 slab)
hipError_t launch_permute_rows_tile16_bf16(const float* src, void* dst, int64_t stride, const int64_t* src_of,
                                           int64_t dst_rows, hipStream_t stream);

// out[r][0..dim) = in[r] / (||in[r]|| + 1e-9); out[r][dim..out_stride) = 0 for r < n;
// rows [n, n_total) of out are zero-filled (query padding), all in one launch.
hipError_t launch_normalize_rows_f32(const float* in, int64_t in_stride, float* out, int64_t out_stride,
                                     int64_t n, int dim, hipStream_t stream, int64_t n_total = 0);

// Zero `n_pad_rows` rows of `stride` floats starting at `dst` (query padding).
hipError_t launch_zero_rows(float* dst, int64_t stride, int n_rows, hipStream_t stream);

// ---- "tile16" corpus layout (what the scan kernel streams) ---------------------------------
// Rows live in 16-row blocks of 16*stride floats.  Inside block b, chunk j (columns
// 16j..16j+15) of the 16 rows is ONE contiguous 1 KiB in MFMA lane order:
//   element (row r, col c) -> (r>>4)*16*stride + (c>>4)*256 + ((((c>>2)&3)*16 + (r&15))*4) + (c&3)
// so a wave's `base + lane*16 B` load is fully coalesced AND already is the A operand of
// v_mfma_f32_16x16x4_f32 (lane = g*16 + m holds row m, k-group g).  A slab holds whole blocks.
inline int64_t tile16_offset(int64_t row, int64_t col, int64_t stride) {
    return (row >> 4) * 16 * stride + (col >> 4) * 256 + ((((col >> 2) & 3) * 16 + (row & 15)) * 4) + (col & 3);
}

// packed rows [first_row, first_row+n) <- in[0..n) row-major; optional reference normalise.
hipError_t launch_pack_rows_tile16(const float* in, int64_t in_stride, float* packed, int64_t stride,
                                   int64_t first_row, int64_t n, int dim, int normalize, hipStream_t stream);
// out[0..n) row-major <- packed rows [first_row, first_row+n).
hipError_t launch_unpack_rows_tile16(const float* packed, int64_t stride, int64_t first_row, int64_t n, int dim,
                                     float* out, int64_t out_stride, hipStream_t stream);
// out[i] <- row row_ids[i] (device array) of the packed slab; ids outside [0, n_rows) leave their output row untouched
hipError_t launch_gather_rows_tile16(const float* packed, int64_t stride, const int64_t* row_ids, int64_t n, int64_t n_rows,
                                     int dim, float* out, int64_t out_stride, hipStream_t stream);

// Synthetic corpus: packed rows [first_row, first_row+n) get iid N(0,1) from Philox4x32-10
// keyed by (seed, row_id_base + row, col/4), L2-normalised; padding columns zero.
hipError_t launch_fill_synthetic_f32(float* packed, int64_t stride, int64_t first_row, int64_t n, int dim,
                                     uint64_t seed, int64_t row_id_base, hipStream_t stream);

hipError_t launch_fill_i32(int32_t* dst, int64_t n, int32_t value, hipStream_t stream);
// dst[i] = base + i
hipError_t launch_iota_i64(int64_t* dst, int64_t n, int64_t base, hipStream_t stream);
// dst[i] = max(src[i], 0): device-source row tags (negative = reserved tombstone code -> 0)
hipError_t launch_copy_tags_clamped(int32_t* dst, const int32_t* src, int64_t n, hipStream_t stream);

}  // namespace rass
