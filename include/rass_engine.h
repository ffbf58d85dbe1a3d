/*
 * rass_engine.h — C ABI of the MI355X-native embedding + vector-search engine
 * that drops in behind RASSEngine's embed_*() functions and OpenSearchIndexer
 * k-NN lookup.
 *
 * The reference has no FFI for this path: its boundary is HTTP/JSON to Ollama
 * (app/main.py:225-237) and to the OpenSearch k-NN plugin (app/main.py:1552).
 * Every entry point below names the reference call site whose arithmetic it
 * replaces.  The Python shim (rassengine_amd/) binds these with ctypes and is
 * the only caller; see INTEGRATION.md for the reference-side rebinding.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary
 *   - every function returns RASS_OK (0) or a negative rass_status; the text of
 *     the last failure on the calling thread is rass_last_error()
 *   - nothing throws across the ABI
 *   - "d_" parameters are device (HBM) pointers, everything else is host memory
 *   - `stream` parameters are a hipStream_t passed as void* (NULL = the
 *     engine's own stream / the null stream for the stateless kernels)
 *   - one engine drives ONE GPU; multi-GPU is one process per GPU (RCCL via
 *     torch.distributed in the Python layer), never several devices per engine
 */
#ifndef RASS_ENGINE_H
#define RASS_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RASS_ABI_VERSION 1

typedef enum rass_status {
    RASS_OK = 0,
    RASS_ERR_INVALID = -1,     /* bad argument (shape, k, dtype, NULL) */
    RASS_ERR_HIP = -2,         /* a HIP runtime call failed */
    RASS_ERR_OOM = -3,         /* device or host allocation failed */
    RASS_ERR_NOT_FOUND = -4,   /* unknown index name / row */
    RASS_ERR_UNSUPPORTED = -5, /* valid request this build cannot serve */
    RASS_ERR_IO = -6           /* save/load failure */
} rass_status;

/* Corpus storage dtype.  RASS_F32 is the parity path (exact fp32 MFMA, bit-equal to the
 * oracle's fmaf-order emulation).  RASS_BF16 is a first-class bf16-ONLY corpus: rows are normalised in fp32, rounded
 * to bf16 and stored in the "tile16b" layout (half the HBM per row, half the bytes per scan); a search rounds the
 * normalised queries to bf16 and runs v_mfma_f32_16x16x32_bf16 with fp32 accumulation, so a returned score is the
 * fp32-accumulated dot product of the two bf16-rounded unit vectors (|error| vs the fp32 cosine ~1e-3, the
 * north_star tolerance; recall@k vs the fp32 index is measured, bench.py --corpus-dtype bf16).  Needs dim padded to
 * a multiple of 256; masked filters and k > RASS_MAX_K work as on an fp32 index (the bf16 scan's EXT variant); the
 * prefilter mode and the IVF build are fp32-only. */
typedef enum rass_dtype {
    RASS_F32 = 0,
    RASS_BF16 = 1,
    RASS_I8 = 2     /* ONLY as the slab_dtype of rass_ivf_build_ex / _prefix (an index is fp32 or bf16) */
} rass_dtype;

/* Limits of the fused scan kernel. */
#define RASS_MAX_K 32        /* top-k kept in one half-wave sorted list */
#define RASS_MAX_QBATCH 32   /* queries per scan launch (two 16-wide MFMA N tiles) */
#define RASS_MAX_K_MULTIPASS 4096 /* rass_index_search_ex: k > RASS_MAX_K is served in passes of RASS_MAX_K */
#define RASS_MAX_DEVICE_BATCH 4096 /* queries per rass_index_search_device_batch call */
/* Row tag layout used by the Python shim (the engine itself only compares integers): bits 0..23 the
 * patientId dictionary code (0 = none), bits 24..30 the doc_type code (0 = none).  A masked filter
 * (rass_index_search_ex) selects on either field or both with one compare. */
#define RASS_TAG_PATIENT_MASK 0x00ffffff
#define RASS_TAG_DOCTYPE_SHIFT 24
#define RASS_TAG_DOCTYPE_MASK 0x7f000000
#define RASS_ROW_TAG_DELETED (-1)
#define RASS_QFILTER_NONE (-1)

typedef struct rass_engine rass_engine_t;
typedef struct rass_index rass_index_t;

/* ------------------------------------------------------------------ misc */

int rass_abi_version(void);
/* Thread-local text of the last error on this thread ("" if none). */
const char* rass_last_error(void);
/* Number of visible HIP devices, or a negative rass_status. */
int rass_device_count(void);

/* ---------------------------------------------------------------- engine */

/* One engine per process per GPU.  `dim` = EMBED_DIM (app/main.py:80), 1 .. 2048.  Up to 1024 columns every feature of
 * this header applies; a WIDE-row engine (1024 < dim <= 2048: what an encoder of hidden size 1536 / 2048 emits) serves
 * fp32 flat indices — add / delete / get / save / load and every search entry point incl. masked filters and k > 32 —
 * (and, for k <= 16, by the int8 prefilter mode: rass_index_set_prefilter), while a bf16 corpus, the bf16 prefilter, the IVF
 * build and cross-index batches answer RASS_ERR_UNSUPPORTED. */
int rass_engine_create(int device, int dim, rass_engine_t** out);
void rass_engine_destroy(rass_engine_t* eng);
int rass_engine_dim(const rass_engine_t* eng);
int rass_engine_device(const rass_engine_t* eng);
/* Run all engine work on a caller-owned hipStream_t (e.g. torch's current
 * stream).  The value is taken literally: NULL is HIP's legacy null stream. */
int rass_engine_set_stream(rass_engine_t* eng, void* stream);
/* Go back to the engine's own (non-blocking) stream. */
int rass_engine_reset_stream(rass_engine_t* eng);
/* The hipStream_t engine work is currently enqueued on. */
void* rass_engine_get_stream(rass_engine_t* eng);
int rass_engine_synchronize(rass_engine_t* eng);

/* ----------------------------------------------------------------- index */

/* Replaces ensure_index_exists()'s knn_vector field (app/main.py:350-579,
 * vector field 563-572; name from get_index_name 346-347): look up the named
 * cosine index, creating it when absent.  O(1) when it exists, because the
 * reference constructs OpenSearchIndexer per request (app/main.py:2802). */
int rass_index_open(rass_engine_t* eng, const char* name, rass_dtype dtype,
                    int64_t initial_capacity_rows, rass_index_t** out);
/* Drop an index and free its HBM. */
int rass_index_drop(rass_engine_t* eng, const char* name);
/* Replaces OpenSearchIndexer.has_any_data's count (app/main.py:1470-1478):
 * live (non-deleted) rows. */
int64_t rass_index_count(const rass_index_t* idx);
/* Rows ever appended (live + tombstoned) = next row id. */
int64_t rass_index_rows(const rass_index_t* idx);
int rass_index_dim(const rass_index_t* idx);
int rass_index_row_stride(const rass_index_t* idx); /* elements: dim padded to 128 (dim <= 1024) or to 256 (above) */
/* RASS_F32 / RASS_BF16 as given to rass_index_open (or read back by rass_index_load). */
int rass_index_dtype(const rass_index_t* idx);
/* != 0 once any row carries a caller-assigned GLOBAL id (rass_index_add_ex: a shard of a multi-GPU index).  Together
 * with the dtype and the row count this is what decides whether an index may join a cross-index batch
 * (rass_index_search_multi: fp32, plain row ids, <= 65 536 tiles of 32 rows per batch). */
int rass_index_has_global_ids(const rass_index_t* idx);

/* Replaces the vector half of store_fhir_docs_in_opensearch (app/main.py:
 * 1245-1282: normalise 1249-1251 + bulk index).  Appends n rows of `dim`
 * floats (host memory).  When `normalize` != 0 rows are L2-normalised on the
 * GPU with the reference's formula e / (||e|| + 1e-9).  `tags` (may be NULL =
 * all 0) is one int32 per row: the dictionary code of the row's patientId
 * (the reference's `_routing` / term-filter key, app/main.py:1263, 1549);
 * must be >= 0.  Row ids are insertion ordinals; *first_row receives the id of
 * the first appended row.  Append-only: overwrite = rass_index_delete + add. */
int rass_index_add(rass_index_t* idx, const float* vecs, const int32_t* tags,
                   int64_t n, int normalize, int64_t* first_row);
/* Same, from device memory (the encoder's pooled output), async on the
 * engine stream.  Device tags cannot be validated without a sync: negative values are stored as
 * 0 (the tombstone code belongs to rass_index_delete). */
int rass_index_add_device(rass_index_t* idx, const float* d_vecs,
                          const int32_t* d_tags, int64_t n, int normalize,
                          int64_t* first_row);
/* Append with CALLER-ASSIGNED ids: row i of the batch is reported by searches as
 * first_global_id + i instead of its ordinal (first_global_id = -1: ordinals, as rass_index_add).
 * This is what makes an index one SHARD of a multi-GPU index whose batches are dealt round-robin to
 * the ranks (rassengine_amd/serving.py; the reference's analogue is OpenSearch routing docs to
 * SHARD_COUNT shards, app/main.py:89, 357): ids must ascend with the append order, so the per-shard
 * (score desc, id asc) order is the global one and the cross-shard merge reproduces the single-index
 * result.  rass_index_delete / get_row(s) keep addressing rows by ORDINAL (*first_row); the caller
 * maps global ids to (rank, ordinal).  `device_source` != 0: vecs / tags are device pointers.
 * On such an index id_base is ignored, k <= RASS_MAX_K, and the prefilter mode is not used. */
int rass_index_add_ex(rass_index_t* idx, const float* vecs, const int32_t* tags,
                      int64_t n, int normalize, int64_t first_global_id,
                      int device_source, int64_t* first_row);
/* Tombstone a row (the overwrite semantics of `_id=doc_id`, app/main.py:1260). */
int rass_index_delete(rass_index_t* idx, int64_t row);
/* Copy one stored (normalised) row back to the host as fp32 (dim floats): the
 * reference returns the embedding inside `_source` (app/main.py:1555-1557). */
int rass_index_get_row(rass_index_t* idx, int64_t row, float* out);
/* Same for rows [first_row, first_row + n): n x dim floats, row-major. */
int rass_index_get_rows(rass_index_t* idx, int64_t first_row, int64_t n,
                        float* out);

/* Replaces the knn query of OpenSearchIndexer.semantic_search (app/main.py:
 * 1527-1560) and the knn sub-clauses of hybrid_search (1595),
 * hybrid_structured_search (1754), multi_intent_search (2003): exact cosine
 * top-k.  `queries` is nq x dim fp32 (host); they are re-normalised on the
 * GPU exactly as the reference does (1536-1537).  `q_filter` (may be NULL) is
 * one int32 per query: RASS_QFILTER_NONE or the patientId code to restrict to
 * (the `term: patientId` filter, 1549), applied as a pre-filter.
 * Outputs: out_scores[nq*k] raw cosine, best first; out_ids[nq*k] global row
 * ids (id_base + local row), -1 (score -inf) where fewer than k rows match.
 * Order: score descending, ties by id ascending.  1 <= k <= RASS_MAX_K; any nq
 * (scanned in batches of RASS_MAX_QBATCH). */
int rass_index_search(rass_index_t* idx, const float* queries, int nq, int k,
                      const int32_t* q_filter, float* out_scores,
                      int64_t* out_ids);
/* Extended host search.  (1) `q_filter_mask` (may be NULL = exact compare): a row matches query q
 * when (row_tag & q_filter_mask[q]) == q_filter[q], so one compare serves `term: patientId`
 * (app/main.py:1549), the `term: doc_type` of hybrid_structured_search (app/main.py:1765) or both.
 * (2) 1 <= k <= RASS_MAX_K_MULTIPASS: the reference passes the caller's top_k straight through
 * (app/main.py:2882, 3008); k > RASS_MAX_K is served exactly, in ceil(k/32) passes, pass p ranking only
 * the rows strictly after pass p-1's last hit under (score desc, id asc).
 * Thread-safe: concurrent calls (same or different indices) overlap their host round trips; the
 * engine lock is held while enqueuing only. */
int rass_index_search_ex(rass_index_t* idx, const float* queries, int nq, int k,
                         const int32_t* q_filter, const int32_t* q_filter_mask,
                         float* out_scores, int64_t* out_ids);
/* CROSS-INDEX batch: query i is answered over index idxs[i] (the reference keeps one index per user,
 * app/main.py:346-347, and serves users concurrently: their queries then share ONE scan launch instead of one
 * launch per user).  All indices must live on one engine, be fp32 and carry plain row ids; any nq (groups of 32).
 * Each distinct index of a group is streamed once, whatever the number of its queries; out_ids are rows of the
 * query's own index.  A group may cover at most 65 536 32-row tiles (2 M rows) over its distinct indices. */
int rass_index_search_multi(rass_index_t* const* idxs, const float* queries, int nq, int k,
                            const int32_t* q_filter, const int32_t* q_filter_mask,
                            float* out_scores, int64_t* out_ids);
/* Device-resident variant for the multi-GPU path and the benchmark: queries
 * and outputs live in HBM, nothing is synchronised; nq <= RASS_MAX_QBATCH.
 * `id_base` is added to local row ids (row-sharded corpus). */
int rass_index_search_device(rass_index_t* idx, const float* d_queries, int nq,
                             int k, const int32_t* d_q_filter, int64_t id_base,
                             float* d_out_scores, int64_t* d_out_ids);

/* Same with the masked filter of rass_index_search_ex ((tag & mask) == filter); on an index
 * with caller-assigned ids (rass_index_add_ex) those ids are reported and id_base is ignored. */
int rass_index_search_device_ex(rass_index_t* idx, const float* d_queries,
                                int nq, int k, const int32_t* d_q_filter,
                                const int32_t* d_q_filter_mask, int64_t id_base,
                                float* d_out_scores, int64_t* d_out_ids);

/* One pass of a k > RASS_MAX_K search over a SHARD (rassengine_amd/serving.py): query q ranks only the rows strictly
 * after (d_after_score[q], row d_after_row[q]) in (score desc, row asc) order, i.e. score < after_score, or equal
 * and row ordinal > after_row (-1: every tying row counts).  A multi-GPU front asks every shard for its next 32
 * behind the previous pass's last GLOBAL hit — each rank translates that hit's id into its own row ordinals, which
 * ascend with the ids — and merges; the reference passes the caller's top_k straight through (app/main.py:2882,
 * 3008).  Reported ids are the index's own (caller-assigned ids where rass_index_add_ex gave them, else ordinals). */
int rass_index_search_device_after(rass_index_t* idx, const float* d_queries, int nq, int k,
                                   const int32_t* d_q_filter, const int32_t* d_q_filter_mask,
                                   const float* d_after_score, const int64_t* d_after_row,
                                   float* d_out_scores, int64_t* d_out_ids);

/* Many launch groups in one call (nq <= RASS_MAX_DEVICE_BATCH, k <= RASS_MAX_K): the result is, bit for bit,
 * that of rass_index_search_device on consecutive groups of RASS_MAX_QBATCH queries, but on an fp32 index the
 * batch shares ONE normalise launch and ONE merge launch and runs the groups' score-floor sample passes back to
 * back ahead of the big scans, so the serial tail a lone group pays after its scan is paid once.
 * This is the shape of the reference's load: embed_texts_in_batches / ask() under concurrent users hand the
 * engine many queries at once (app/main.py:1536-1560 is called per request; the micro-batcher coalesces them).
 * Group g's results go to d_out_scores + g * out_scores_group_stride (floats) and d_out_ids + g *
 * out_ids_group_stride (int64s); 0 = contiguous [nq][k].  d_q_filter: nq tags or NULL. */
int rass_index_search_device_batch(rass_index_t* idx, const float* d_queries, int nq, int k,
                                   const int32_t* d_q_filter, int64_t id_base,
                                   float* d_out_scores, int64_t* d_out_ids,
                                   int64_t out_scores_group_stride, int64_t out_ids_group_stride);

/* Prefilter mode (SURVEY §8f-4 "bf16 (or int8)"; the reference's own index is approximate, app/main.py:563-572), OFF by
 * default.  `enable` = RASS_PREFILTER_BF16 (1): keep a bf16 copy of the slab, scan IT (half the HBM bytes per pass, bf16
 * MFMA) for the 32 best candidates per query; RASS_PREFILTER_INT8 (2): keep an int8 copy (a quarter of the bytes; per row
 * q = rint(x * 127 / max|x|) and one fp32 scale, queries quantised the same way, v_mfma_i32_16x16x64_i8; a candidate's score
 * is (float)(exact integer dot) * row scale).  Either way those candidates' scores are then recomputed exactly from the fp32
 * slab in the flat kernel's fmaf order and the exact top-k among them is returned: returned scores are bit-identical to the
 * flat path; the id set equals the flat result whenever the true top-k lies inside the candidate top-32 (measured as recall,
 * not guaranteed).  Used for k <= 16 only (k > 16 takes the exact flat scan).  bf16 needs dim padded to a multiple of 256 and
 * dim <= 1024; int8 serves every dim an index takes (wide rows, 1024 < dim <= 2048, included).  0 = off (frees the copy);
 * switching modes rebuilds the copy from the fp32 rows. */
#define RASS_PREFILTER_OFF 0
#define RASS_PREFILTER_BF16 1
#define RASS_PREFILTER_INT8 2
int rass_index_set_prefilter(rass_index_t* idx, int enable);
int rass_index_get_prefilter(const rass_index_t* idx);   /* the mode: 0 / 1 / 2 */
/* The candidate lists of the active prefilter mode for <= 32 device queries, BEFORE the exact re-rank: [nq][32] candidate
 * scores (bf16: fp32-accumulated dot of the bf16-rounded operands; int8: as above) and LOCAL rows, (score desc, row asc),
 * -inf / -1 past the end.  Parity hook of the integer path (tests compare it with the oracle bit for bit); stream-ordered. */
int rass_index_candidates_device(rass_index_t* idx, const float* d_queries, int nq, const int32_t* d_q_filter,
                                 float* d_cand_scores, int64_t* d_cand_rows);

/* Shard persistence: raw rows + tags + manifest header. */
int rass_index_save(rass_index_t* idx, const char* path);
int rass_index_load(rass_engine_t* eng, const char* name, const char* path,
                    rass_index_t** out);

/* Fill rows [first, first+n) of the index with synthetic unit vectors
 * generated ON DEVICE (counter-based RNG keyed by (seed, global row id), so
 * any shard regenerates identically; SURVEY §8d cfg 2/4).  Appends when
 * first == rass_index_rows().  `row_id_base` offsets the RNG key for sharding. */
int rass_index_fill_synthetic(rass_index_t* idx, int64_t n, uint64_t seed,
                              int64_t row_id_base);

/* --------------------------------------------- stateless kernel launchers */

/* Bytes of scratch the scan needs for (nq, k). */
size_t rass_scan_workspace_bytes(int nq, int k);

/* Corpus layout in HBM ("tile16"): rows live in 16-row blocks of 16*row_stride
 * floats; inside a block, chunk j (columns 16j..16j+15) of the 16 rows is one
 * contiguous 1 KiB in MFMA lane order, i.e. element (row r, col c) sits at
 *   (r>>4)*16*row_stride + (c>>4)*256 + ((((c>>2)&3)*16 + (r&15))*4) + (c&3)
 * floats from the slab base.  Every wave-level load of the scan is then a fully
 * coalesced 1 KiB burst that already is the MFMA A operand.  row_stride = dim
 * rounded up to 128 (to 256 above 1024 columns), zero padded; a slab holds
 * whole blocks (rows rounded up to 16).  The two converters below move between
 * row-major and tile16. */
int rass_pack_rows_f32(const float* d_in, int64_t in_stride, float* d_packed,
                       int64_t row_stride, int64_t first_row, int64_t n, int dim,
                       int normalize, void* stream);
int rass_unpack_rows_f32(const float* d_packed, int64_t row_stride,
                         int64_t first_row, int64_t n, int dim, float* d_out,
                         int64_t out_stride, void* stream);
/* d_out[i] (row-major, out_stride) <- row d_row_ids[i] of a tile16 slab holding n_rows rows, i < n (device arrays): the
 * scattered form of rass_unpack_rows_f32 (k-means seeds are nlist sample rows all over the slab). */
int rass_gather_rows_f32(const float* d_packed, int64_t row_stride, int64_t n_rows,
                         const int64_t* d_row_ids, int64_t n, int dim, float* d_out,
                         int64_t out_stride, void* stream);

/* K1+K2: fused flat cosine scan + per-workgroup top-k + merge over a tile16
 * fp32 corpus slab in HBM.  Rows must already be normalised (rass_pack_rows_f32
 * with normalize=1 does both); d_queries (nq x dim, row-major) are normalised
 * by the launcher.  row_stride is in elements, 128 * {1..8} or (wide rows)
 * 256 * {5..8}, with zero padding beyond dim; the slab must hold
 * ceil(n_rows/16) whole blocks.
 * d_row_tag / d_q_filter may be NULL. */
int rass_scan_topk_f32(const float* d_corpus, int64_t n_rows, int dim,
                       int64_t row_stride, const int32_t* d_row_tag,
                       const float* d_queries, int nq,
                       const int32_t* d_q_filter, int k, int64_t id_base,
                       float* d_out_scores, int64_t* d_out_ids,
                       void* d_workspace, size_t workspace_bytes,
                       void* stream);

/* K2/K3 merge: n_lists sorted candidate lists per query, laid out
 * [n_lists][nq][k] (score f32, id i64; id < 0 = empty), -> [nq][k] with the
 * same total order.  Used after the RCCL all-gather of per-shard top-k. */
int rass_topk_merge(const float* d_scores, const int64_t* d_ids, int n_lists,
                    int nq, int k, float* d_out_scores, int64_t* d_out_ids,
                    void* stream);

/* Same with explicit list strides (in elements): lets the merge read the G
 * per-rank records of ONE all-gather, each packed as nq*k f32 scores followed
 * by nq*k i64 ids, without unpacking (rassengine_amd/dist.py). */
int rass_topk_merge_strided(const float* d_scores, const int64_t* d_ids,
                            int64_t score_list_stride, int64_t id_list_stride,
                            int n_lists, int nq, int k, float* d_out_scores,
                            int64_t* d_out_ids, void* stream);

/* The same merge for several launch groups in ONE launch: query q of nq_total belongs to group q / group_size,
 * whose lists start g * score_group_stride / id_group_stride elements after group 0's (the gathered
 * [rank][group][record] buffer of dist.ShardedSearch.search_batch); output contiguous [nq_total][k]. */
int rass_topk_merge_strided_batch(const float* d_scores, const int64_t* d_ids,
                                  int64_t score_list_stride, int64_t id_list_stride,
                                  int n_lists, int nq_total, int group_size,
                                  int64_t score_group_stride, int64_t id_group_stride, int k,
                                  float* d_out_scores, int64_t* d_out_ids, void* stream);

/* the cross-shard exchange without a collective.  Rank 0 creates a buffer in its HBM and hands
 * the 64-byte HIP IPC handle to the other ranks' processes (one process per GPU); every rank then STORES its
 * packed per-shard top-k record into its slot (over xGMI between GPUs) and releases a per-rank sequence flag at
 * system scope (rass_peer_post: copy, fence, flag); rank 0 enqueues rass_peer_wait (one workgroup acquiring the
 * n flags, BOUNDED: after max_spins polls it gives up and stores 1 + the missing rank in *d_status instead of
 * hanging the GPU) in front of its rass_topk_merge_strided.  Replaces the OpenSearch shard -> coordinator
 * response (SHARD_COUNT, app/main.py:89, 357).  Layout and step protocol: rassengine_amd/dist.py
 * (PeerMergeSearch).  The processes need HSA_ENABLE_IPC_MODE_LEGACY=0 on this platform. */
int rass_peer_buffer_create(int device, size_t bytes, void** d_ptr, unsigned char* handle64);
int rass_peer_buffer_open(int device, const unsigned char* handle64, void** d_ptr);
int rass_peer_buffer_close(void* d_ptr, int opened_from_handle);
int rass_peer_post(const void* d_record, size_t bytes, void* d_remote_slot, void* d_remote_flag,
                   uint64_t seq, void* stream);
int rass_peer_wait(const void* d_flags, int n, int flag_stride_bytes, uint64_t seq, int* d_status,
                   int64_t max_spins, void* stream);

/* a4 (app/main.py:1249-1251, 1536-1537): out = in / (||in||_2 + 1e-9), rows
 * of `dim` floats read at in_stride, written at out_stride (elements); the
 * out_stride - dim tail of each output row is zero-filled. */
int rass_normalize_rows_f32(const float* d_in, int64_t in_stride, float* d_out,
                            int64_t out_stride, int64_t n, int dim,
                            void* stream);

/* -------------------------------------------------------------------- IVF
 * K9: inverted-file cosine index (nlist coarse centroids + contiguous lists)
 * for shards where a flat scan per query batch is too much (BASELINE cfg 5:
 * 100 M rows, IVF-4096).  Stands in for the sub-linear behaviour of the
 * reference's HNSW (app/main.py:563-572) while staying an HBM-streaming
 * kernel: probing = the flat fused scan over the centroid slab, a plan kernel,
 * and the same fused scan over the union of the batch's probed lists.
 * Approximate by construction: recall@k vs the flat index is a function of
 * nprobe and is measured, never assumed (nprobe = nlist is exact). */
typedef struct rass_ivf rass_ivf_t;

/* K9(i): the two O(rows) steps of spherical k-means, over rows already resident in `idx`'s slab, asynchronous
 * on the engine stream.  The processed rows are 32-row blocks first_block, first_block + block_step, ...
 * (n_blocks of them: a strided training sample, or every block with block_step = 1).
 *   rass_kmeans_assign: d_assign[b*32 + r] = arg max over the nlist centroids of the cosine with row r of the
 *     b-th processed block (exact fp32 MFMA, ties -> lowest list); d_best (may be NULL) the winning cosine.
 *     `d_centroids_tile16`: nlist normalised centroids as a tile16 slab (rass_pack_rows_f32, normalize = 1),
 *     row_stride = rass_index_row_stride(idx), whole 16-row blocks.  Entries of rows past rass_index_rows() or
 *     tombstoned are computed like any other and must be ignored by the caller.
 *   rass_kmeans_accumulate: d_sums[list][0..dim) += row, d_counts[list] += 1 for every processed row below
 *     rass_index_rows() (fp32 atomics; d_sums is nlist x dim row-major, caller-zeroed).
 * The all-reduce over ranks, the normalisation of the sums and the re-seeding of empty lists are O(nlist x dim)
 * and stay with the caller (rassengine_amd/ivf.py). */
int rass_kmeans_assign(rass_index_t* idx, int64_t first_block, int64_t block_step,
                       int64_t n_blocks, const float* d_centroids_tile16, int nlist,
                       int32_t* d_assign, float* d_best);
int rass_kmeans_accumulate(rass_index_t* idx, int64_t first_block, int64_t block_step,
                           int64_t n_blocks, const int32_t* d_assign, float* d_sums,
                           float* d_counts, int nlist);

/* Build from a flat index: `centroids` nlist x dim fp32 (host; normalised
 * here), `assign[r]` = list of source row r (host, one per appended row;
 * tombstoned rows are skipped).  Training / assignment are offline and live in
 * the Python layer (rassengine_amd/ivf.py).  Result ids are the source index's
 * row ids.  The source index may be dropped afterwards. */
int rass_ivf_build(rass_index_t* src, const float* centroids, int nlist,
                   const int32_t* assign, rass_ivf_t** out);
/* The same with the IVF's list-ordered copy of the rows held as `slab_dtype`: RASS_F32 = rass_ivf_build;
 * RASS_BF16 = the rows rounded to bf16 (half the HBM bytes per probed row; needs the row stride to be a multiple of
 * 256): the fine scan is then the bf16 scan (queries rounded to bf16, fp32 accumulation) and a probe returns what a
 * flat RASS_BF16 index over the same rows returns, restricted to the probed lists.  The source index is fp32 either
 * way (k-means and the assignment read it) and may be dropped afterwards.  (SURVEY §8f-4's bf16 path for cfg 5.)
 * RASS_I8 = an fp32 slab (lists on 64-row tiles) PLUS its int8 copy (per-row-scaled, as rass_index_set_prefilter mode 2): the
 * fine scan reads the int8 copy (a quarter of the bytes per probed row) and keeps 32 candidates per query, which are then
 * rescored exactly from the fp32 slab — returned scores are the fp32 IVF's bit for bit, the id set equals it whenever the
 * probed lists' true top-k lies inside the int8 top-32 (measured as recall).  Serves k <= 16 (RASS_ERR_UNSUPPORTED beyond). */
int rass_ivf_build_ex(rass_index_t* src, const float* centroids, int nlist,
                      const int32_t* assign, rass_dtype slab_dtype, rass_ivf_t** out);
void rass_ivf_destroy(rass_ivf_t* ivf);
/* IVF persistence (SURVEY §8f-3 for the IVF shard): the whole device state — list table, slab ids, tags,
 * centroid slab, permuted row slab — so a load needs neither the flat index nor a re-training.  The file is
 * fsynced before close (write to a temporary name and rename for crash safety, as docstore.py does). */
int rass_ivf_save(rass_ivf_t* ivf, const char* path);
int rass_ivf_load(rass_engine_t* eng, const char* path, rass_ivf_t** out);
int64_t rass_ivf_rows(const rass_ivf_t* ivf);
int rass_ivf_nlist(const rass_ivf_t* ivf);
int rass_ivf_dtype(const rass_ivf_t* ivf); /* rass_dtype of the row slab; -1 for NULL */
/* Same contract as rass_index_search; nprobe >= 1 lists per query (capped at
 * nlist; nprobe > 32 selects by a per-query score threshold, ties may add lists).
 * *scanned_rows (may be NULL) receives the rows the fine scans touched. */
int rass_ivf_search(rass_ivf_t* ivf, const float* queries, int nq, int k,
                    int nprobe, const int32_t* q_filter, float* out_scores,
                    int64_t* out_ids, int64_t* scanned_rows);
int rass_ivf_search_device(rass_ivf_t* ivf, const float* d_queries, int nq,
                           int k, int nprobe, const int32_t* d_q_filter,
                           float* d_out_scores, int64_t* d_out_ids);
/* Up to 1 024 queries (32 launch groups) per call, device-resident: bit-identical to rass_ivf_search_device on
 * consecutive groups of 32 queries, with ONE normalise, ONE grouped coarse scan over the centroid slab, ONE plan launch
 * (the coarse lists are merged inside it) and ONE grouped merge for the whole batch — 4 + G launches instead of 5 G,
 * which is what an IVF probe at nprobe 1-2 (where two-level training puts recall 1.0) is bound by.  Outputs contiguous
 * [nq][k]; d_scanned_per_group (may be NULL): int64 per launch group, the rows its fine scan touched.  nprobe > 32 runs
 * group by group.  (The reference's k-NN lookup under concurrent load: app/main.py:1552; BASELINE configs[4].) */
int rass_ivf_search_device_batch(rass_ivf_t* ivf, const float* d_queries, int nq, int k, int nprobe,
                                 const int32_t* d_q_filter, float* d_out_scores, int64_t* d_out_ids,
                                 int64_t* d_scanned_per_group);

/* ---- IVF + flat delta: the approximate index that stays incrementally insertable.
 * Replaces what the reference gets from OpenSearch's knn_vector field: an approximate (HNSW) index
 * (app/main.py:563-572) that takes bulk inserts of 64 docs at any time (app/main.py:1253-1282) and honours
 * `_id` overwrites / deletes.  Here: the IVF is a snapshot of source rows [0, covered); rows the source index
 * takes afterwards (the DELTA) are scanned exactly from its own slab in the same call and merged with the
 * probe's list by the same merge kernel; tombstones are honoured on both sides.
 *
 * rass_ivf_build_prefix: rass_ivf_build_ex over the first `n_rows` source rows only (-1 = all).  An IVF that
 * is to be searched with a delta must cover a multiple of 32 rows (the scan's tile), or every row.
 * rass_ivf_delete: tombstone source row `src_row` inside the IVF's slab (a no-op for rows it does not cover or
 * that are already gone) — call it next to rass_index_delete on the source index.
 * rass_ivf_search_delta[_device]: per query the exact top-k over (the rows of its nprobe best lists) U (source
 * rows >= covered), ties (score desc, source ordinal asc); ids = source ordinals, or the source index's
 * caller-assigned ids (rass_index_add_ex).  `q_filter` / `q_filter_mask` as rass_index_search_ex.  With
 * nprobe >= nlist the result equals rass_index_search_ex on the source index bit for bit (fp32 slab).
 * nq <= RASS_MAX_QBATCH for the device variant; k <= RASS_MAX_K (deeper lists: search the source index).
 * IVF files written since round 4 (versions 3 / 4) carry `covered`; older files load with covered = the
 * largest slab id + 1. */
int rass_ivf_build_prefix(rass_index_t* src, const float* centroids, int nlist,
                          const int32_t* assign, rass_dtype slab_dtype,
                          int64_t n_rows, rass_ivf_t** out);
int64_t rass_ivf_covered_rows(const rass_ivf_t* ivf);
int rass_ivf_delete(rass_ivf_t* ivf, int64_t src_row);
int rass_ivf_search_delta(rass_ivf_t* ivf, rass_index_t* src, const float* queries,
                          int nq, int k, int nprobe, const int32_t* q_filter,
                          const int32_t* q_filter_mask, float* out_scores,
                          int64_t* out_ids, int64_t* scanned_rows);
int rass_ivf_search_delta_device(rass_ivf_t* ivf, rass_index_t* src,
                                 const float* d_queries, int nq, int k, int nprobe,
                                 const int32_t* d_q_filter, const int32_t* d_q_filter_mask,
                                 float* d_out_scores, int64_t* d_out_ids);

/* ---------------------------------------------------------------- encoder
 * Replaces ollama_embed_text / embed_texts_in_batches / embed_query's HTTP hop
 * to Ollama (app/main.py:225-274): a BERT-class post-LN sentence encoder
 * (mxbai-embed-large class, OLLAMA_EMBED_MODEL app/main.py:67) run as one
 * batched, varlen-packed forward of hand-written gfx950 kernels (bf16 MFMA
 * GEMMs with fused bias/GELU/residual epilogues, LDS-resident attention,
 * LayerNorm, pooling). */
typedef struct rass_encoder rass_encoder_t;

typedef struct rass_encoder_config {
    int32_t vocab_size;     /* 30522 */
    int32_t hidden;         /* 1024 = EMBED_DIM; must be heads * 64 */
    int32_t layers;         /* 24 */
    int32_t heads;          /* 16 */
    int32_t intermediate;   /* 4096 */
    int32_t max_positions;  /* <= 512 */
    int32_t pooling;        /* 0 = cls, 1 = mean over tokens */
    int32_t normalize;      /* != 0: L2-normalise the pooled vector, e/(||e||+1e-9) */
    float layer_norm_eps;   /* 1e-12 */
} rass_encoder_config;

int rass_encoder_create(int device, const rass_encoder_config* cfg,
                        rass_encoder_t** out);
void rass_encoder_destroy(rass_encoder_t* enc);
int rass_encoder_hidden(const rass_encoder_t* enc);
/* One call per tensor, by its Hugging Face BERT name ("embeddings.word_
 * embeddings.weight", "encoder.layer.7.attention.self.query.weight", ...),
 * fp32 host data in the checkpoint's layout ([out][in] for Linear). */
int rass_encoder_set_weight(rass_encoder_t* enc, const char* name,
                            const float* data, int64_t numel);
/* Checks that every tensor of the architecture was supplied. */
int rass_encoder_finalize(rass_encoder_t* enc);
/* token_ids: all sequences back to back (each already [CLS] ... [SEP],
 * <= max_positions tokens); cu_seqlens[nseq+1] prefix sums starting at 0.
 * out: nseq x hidden fp32, order = input order. */
int rass_encode(rass_encoder_t* enc, const int32_t* token_ids,
                const int32_t* cu_seqlens, int nseq, float* out);
/* Device-resident, asynchronous on `stream`.  NULL means the encoder's OWN
 * (non-blocking) stream, rass_encoder_get_stream() — NOT HIP's legacy null
 * stream, with which it does not synchronise.  The output can be handed
 * straight to rass_index_add_device when the engine runs on the same stream
 * (rass_engine_set_stream(eng, rass_encoder_get_stream(enc)), or one caller
 * stream passed to both); on different streams the caller must order them. */
int rass_encode_device(rass_encoder_t* enc, const int32_t* d_token_ids,
                       const int32_t* d_cu_seqlens, int nseq, int total_tokens,
                       int max_seqlen, float* d_out, void* stream);
/* The hipStream_t rass_encode() and rass_encode_device(stream = NULL) run on. */
void* rass_encoder_get_stream(rass_encoder_t* enc);
/* Counters since create: out[0] forwards launched (rass_encode + rass_encode_device), out[1] sequences, out[2]
 * tokens.  What the embed micro-batcher's tests read: N concurrent embed_query coroutines (app/main.py:2800, up to
 * MAX_EMBED_CONCURRENCY in flight, 250-260) must arrive as one or two forwards, not N. */
int rass_encoder_stats(rass_encoder_t* enc, int64_t out[3]);
/* The encoder's GEMM on its own: Y[m,n] = epi(X[m,k] W[n,k]^T + bias), bf16
 * operands; epilogue 0 bias, 1 bias+residual, 2 bias+GELU(erf).  m_pad
 * (multiple of 128) rows must be allocated; n % 128 == 0, k % 64 == 0. */
int rass_gemm_bf16(const void* d_x, const void* d_w, const float* d_bias,
                   const void* d_residual, void* d_y, int m, int m_pad, int n,
                   int k, int epilogue, void* stream);
/* Same with a caller-owned fp32 scratch (d_ws, ws_bytes >= 2 * m_pad * n * 4): a GEMM over few rows (m_pad <= 256)
 * is then split over K so that enough workgroups stream the weights — the path the encoder takes for embed_query /
 * ollama_embed_text (app/main.py:225-237, 266-274); slices are summed in fixed order (deterministic). */
int rass_gemm_bf16_ws(const void* d_x, const void* d_w, const float* d_bias,
                      const void* d_residual, void* d_y, int m, int m_pad, int n, int k,
                      int epilogue, void* d_ws, size_t ws_bytes, void* stream);
/* The encoder's self-attention on its own (what llama.cpp computes per layer behind the reference's POST to Ollama,
 * app/main.py:225-237): d_qkv bf16 [tokens][3*hidden] (q | k | v per token), sequences packed back to back with
 * d_cu_seqlens[nseq + 1] token offsets (total_tokens = the last one, as the host knows it), heads of 64
 * (hidden = 64 * heads), every sequence <= max_seqlen <= 512, softmax(q k^T / 8) v in fp32, d_ctx bf16
 * [total_tokens][hidden]. */
int rass_attention_bf16(const void* d_qkv, const int32_t* d_cu_seqlens, int nseq, int total_tokens,
                        int max_seqlen, int hidden, int heads, void* d_ctx, void* stream);
/* Query time (embed_query / ollama_embed_text, app/main.py:225-237, 266-274: one text, a dozen tokens): the attention and
 * the attention-output projection in ONE launch — d_y = attention(d_qkv) d_w^T + d_bias + d_residual, d_w bf16 [n][hidden],
 * d_residual / d_y bf16 [total_tokens][n].  Every workgroup of the projection recomputes the attention (its context rows
 * have rass_attention_bf16's bits); a launch of a one-query forward costs 4-5 us whatever it does, this saves one per layer.
 * 1 <= total_tokens <= 32 over all nseq sequences, hidden = 1024, heads = 16, n % 16 == 0; anything else (or
 * RASS_ATTN_FUSE=0 in the environment) -> RASS_ERR_UNSUPPORTED.  rass_encode_device takes it by itself. */
int rass_attention_out_bf16(const void* d_qkv, const int32_t* d_cu_seqlens, int nseq, int total_tokens, int hidden,
                            int heads, const void* d_w, const float* d_bias, const void* d_residual, void* d_y, int n,
                            void* stream);

/* -------------------------------------------------------------- tokenizer
 * BERT (uncased) BasicTokenizer + WordPiece on the host (C++), replacing the
 * tokenisation Ollama / llama.cpp performs on every chunk the reference posts
 * (app/main.py:225-237): lower-case, NFD + accent strip, punctuation split,
 * CJK spacing, greedy longest-match-first pieces, [CLS] ... [SEP], truncated
 * to max_len.  Unicode tables are generated from Python's unicodedata. */
typedef struct rass_tokenizer rass_tokenizer_t;
int rass_tokenizer_create(const char* vocab_path, int lower_case,
                          rass_tokenizer_t** out);
void rass_tokenizer_destroy(rass_tokenizer_t* tok);
int rass_tokenizer_vocab_size(const rass_tokenizer_t* tok);
/* Returns the number of ids written (<= max_len) or a negative status. */
int rass_tokenizer_encode(const rass_tokenizer_t* tok, const char* text,
                          int64_t text_len, int max_len, int32_t* out_ids);
/* n UTF-8 texts -> packed ids (capacity n*max_len) + cu_seqlens[n+1], fanned
 * out over n_threads (<= 0: all cores).  Returns the total token count. */
int64_t rass_tokenizer_encode_batch(const rass_tokenizer_t* tok,
                                    const char* const* texts,
                                    const int64_t* lens, int n, int max_len,
                                    int32_t* out_ids, int32_t* out_cu,
                                    int n_threads);

/* HIP-event timing on an explicit stream (bench.py measures kernels on the
 * stream they run on; torch.cuda.Event only sees torch's current stream). */
typedef struct rass_timer rass_timer_t;
int rass_timer_create(rass_timer_t** out);
void rass_timer_destroy(rass_timer_t* t);
int rass_timer_start(rass_timer_t* t, void* stream);
int rass_timer_stop(rass_timer_t* t, void* stream);
/* Blocks until the stop event has completed; milliseconds between the two. */
int rass_timer_elapsed_ms(rass_timer_t* t, float* ms);

/* Bracket every scan-kernel launch the engine makes with a hipEvent pair on
 * the engine stream (up to max_launches launches), then read back the summed
 * kernel time: bench.py's live per-kernel duration for the roofline figure. */
int rass_engine_kernel_timing_begin(rass_engine_t* eng, int max_launches);
int rass_engine_kernel_timing_end(rass_engine_t* eng, double* total_ms,
                                  int* launches);

/* Device pointers of an index's tile16 slab / tag array (zero-copy interop,
 * e.g. rass_scan_topk_f32 over a row prefix).  Invalidated by growth. */
void* rass_index_device_rows(rass_index_t* idx);
void* rass_index_device_tags(rass_index_t* idx);

/* Name of the scan kernel variant a (dim, nq) request dispatches to, for
 * matching rocprofv3 kernel-trace rows ("" if unsupported). */
const char* rass_scan_kernel_name(int dim, int nq);

#ifdef __cplusplus
}
#endif
#endif /* RASS_ENGINE_H */
