// api.hip — the C ABI of include/rass_engine.h: engine / index objects that own HBM, and
// the stateless launchers.  Host-side C++ only; the kernels live in the other .hip files.
//
// Ownership model: the engine singleton of a process owns the corpus slabs
// for process lifetime; callers own every host buffer they pass in or get filled.
// Threading: add/delete/grow take the index mutex; searches take the engine mutex only
// while ENQUEUING (all GPU work of an engine is ordered on one stream, so the shared device
// scratch and staging are safe by stream order) and wait for their results on a per-call
// event outside of it, on a pinned host slot taken from a small pool: searches on different
// indices (users) overlap their host round trips instead of serialising on a stream sync.
// rows / deleted / has_tags are atomics: searches read them without the index mutex.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rass_engine.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    // A failed HIP call (e.g. an out-of-memory hipMalloc) stays behind as the runtime's "last error" and the
    // NEXT kernel launch's hipGetLastError() would report it as its own: every failure path ends here, clear it.
    (void)hipGetLastError();
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            char _b[512];                                                                               \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                     __LINE__);                                                                         \
            return fail(_e == hipErrorOutOfMemory ? RASS_ERR_OOM : RASS_ERR_HIP, _b);                   \
        }                                                                                               \
    } while (0)

constexpr int kPrefilterMaxK = 16;     // prefilter keeps 32 bf16 candidates: only k <= 16 uses it, wider k scans fp32
constexpr int kMaxGrid = 1024;        // upper bound on scan workgroups (sizing of scratch)
constexpr int kMaxStride = 2048;      // dim_padded limit of the fused scan (128 * {1..8}; wide rows: 256 * {5..8})
constexpr int kNarrowStride = 1024;   // above it a row is "wide": flat fp32 scans of <= 16 queries per launch only
constexpr int64_t kStageRows = 8192;  // host -> device staging granule for add()

int64_t pad128(int64_t d) { return (d + 127) / 128 * 128; }
// The row stride of an index of `dim` columns: whole 128-column units (8 waves x one 16-column chunk); above 1 024
// columns whole 256-column units (the wide-row scan walks a wave's slice in an even number of chunks per panel).
int64_t pad_stride(int64_t d) { return d <= kNarrowStride ? pad128(d) : (d + 255) / 256 * 256; }
const char* kStrideMsg = "row_stride must be 128*{1..8} elements (dim <= 1024) or 256*{5..8} (dim <= 2048)";

int device_cus(int device) {
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    cache[device] = cus;
    return cus;
}

struct IvfPlan {
    const int32_t* work_tile;
    const int32_t* work_rows;
    const uint32_t* work_mask;
    const int32_t* n_work;
    int64_t max_tiles;
};

// Extended per-query filters of a scan (kernels.h ScanArgs): all-null = the plain kernel variant.
struct ScanExt {
    const int32_t* d_q_mask = nullptr;
    const float* d_after_s = nullptr;
    const int64_t* d_after_i = nullptr;
};

struct ScratchLayout {
    size_t q_padded, part_scores, part_ids, q_bf16, cand_scores, cand_ids, sample_best, total;
};

ScratchLayout scratch_layout(int nq, int k) {
    ScratchLayout L;
    size_t off = 0;
    L.q_padded = off;
    off += (size_t)RASS_MAX_QBATCH * kMaxStride * sizeof(float);
    L.part_scores = off;
    off += (size_t)kMaxGrid * nq * k * sizeof(float);
    off = (off + 255) / 256 * 256;
    L.part_ids = off;
    off += (size_t)kMaxGrid * nq * k * sizeof(int64_t);
    off = (off + 255) / 256 * 256;
    // prefilter mode: bf16 queries + the 32 candidates per query handed to the exact re-rank
    L.q_bf16 = off;
    off += (size_t)RASS_MAX_QBATCH * kMaxStride * 2;
    L.cand_scores = off;
    off += (size_t)RASS_MAX_QBATCH * RASS_MAX_K * sizeof(float);
    off = (off + 255) / 256 * 256;
    L.cand_ids = off;
    off += (size_t)RASS_MAX_QBATCH * RASS_MAX_K * sizeof(int64_t);
    off = (off + 255) / 256 * 256;
    L.sample_best = off;  // the sample pass's per-workgroup best scores [32][kMaxSampleGroups]
    off += (size_t)rass::kMaxSampleGroups * 32 * sizeof(float);
    L.total = off;
    return L;
}

// One host search call in flight: pinned staging for a batch of <= 32 queries and its results, and
// the event recorded behind the batch's last copy.
struct HostSlot {
    float* h_q = nullptr;          // [32][dim]
    int32_t* h_filter = nullptr;   // [32]
    int32_t* h_mask = nullptr;     // [32]
    float* h_after_s = nullptr;    // [32]
    int64_t* h_after_i = nullptr;  // [32]
    float* h_out_s = nullptr;      // [32][32]
    int64_t* h_out_i = nullptr;    // [32][32]
    int64_t* h_scanned = nullptr;  // [1]
    void* base = nullptr;          // the one hipHostMalloc behind all of the above
    void* h_items = nullptr;       // pinned work list of a cross-index batch (lazily allocated, kMultiMaxItems)
    hipEvent_t done = nullptr;
    bool busy = false;
};
constexpr int kHostSlots = 8;
constexpr int kMultiMaxItems = 65536;  // 32-row tiles per cross-index batch (2 M rows over all its indices)

}  // namespace

struct rass_engine {
    int device = 0;
    int dim = 0;
    int n_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::map<std::string, rass_index*> indices;
    // scratch for searches (sized for nq = RASS_MAX_QBATCH, k = RASS_MAX_K)
    unsigned char* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // scratch of rass_index_search_device_batch (every launch group's queries, lists and sample bests), grown on demand
    unsigned char* d_batch = nullptr;
    size_t batch_bytes = 0;
    // host-API staging
    float* d_qraw = nullptr;        // [32][dim]
    int32_t* d_qfilter = nullptr;   // [32]
    float* d_out_scores = nullptr;  // [32][32]
    int64_t* d_out_ids = nullptr;   // [32][32]
    float* d_stage = nullptr;       // [kStageRows][dim]
    int32_t* d_stage_tags = nullptr;
    // cross-index batches (rass_index_search_multi): device work list, lazily allocated
    int32_t *d_mw_tile = nullptr, *d_mw_rows = nullptr, *d_mw_n = nullptr;
    uint32_t* d_mw_mask = nullptr;
    const float** d_mw_base = nullptr;
    const int32_t** d_mw_tags = nullptr;
    float* d_stage_t16 = nullptr;   // bf16 indices: (kStageRows + 32) x kMaxStride fp32 tile16 staging, lazily allocated
    int32_t* d_qmask = nullptr;     // [32] masked-filter masks
    float* d_after_s = nullptr;     // [32] continuation bound of a multi-pass top-k (k > 32)
    int64_t* d_after_i = nullptr;   // [32]
    // pinned host slots of the host search API (one per call in flight)
    std::vector<HostSlot> slots;
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    // optional HIP-event bracket around every scan kernel launch (bench.py's roofline leg)
    std::vector<hipEvent_t> ev_pool;  // pairs: [2i] before, [2i+1] after
    int ev_used = 0;                  // pairs recorded since timing_begin
    bool ev_on = false;
};

struct rass_index {
    rass_engine* eng = nullptr;
    std::string name;
    rass_dtype dtype = RASS_F32;
    int dim = 0;
    int64_t stride = 0;
    std::atomic<int64_t> rows{0};      // published after the rows' pack kernels are enqueued
    int64_t capacity = 0;
    std::atomic<int64_t> deleted{0};
    std::atomic<bool> has_tags{false};  // any non-zero tag ever stored
    float* d_rows = nullptr;
    int32_t* d_tags = nullptr;
    int64_t* d_gid = nullptr;               // [capacity] id reported for a row: its ordinal, or the caller's
                                            // GLOBAL id (rass_index_add_ex: a shard of a multi-GPU index)
    std::atomic<bool> has_gid{false};       // any row carries a caller-assigned id
    unsigned short* d_rows_bf16 = nullptr;  // tile16b copy for the prefilter mode (nullptr = off)
    int prefilter = 0;                      // 0 off | 1 bf16 candidate copy | 2 int8 candidate copy (+ a scale per row)
    signed char* d_rows_i8 = nullptr;       // tile16i copy (prefilter mode 2), rows of stride_i8 bytes
    float* d_row_scale = nullptr;           // [capacity] max|x| / 127 of every row (prefilter mode 2)
    int64_t stride_i8 = 0;                  // stride rounded up to 512
    std::vector<uint8_t> host_deleted;  // tombstone bitmap mirror (host)
    std::mutex mu;
};

namespace {

int set_device(const rass_engine* eng) {
    HIP_TRY(hipSetDevice(eng->device));
    return RASS_OK;
}

int index_reserve(rass_index* idx, int64_t need_rows) {
    if (need_rows <= idx->capacity) return RASS_OK;
    int64_t cap = std::max<int64_t>(idx->capacity * 2, std::max<int64_t>(need_rows, 1024));
    cap = (cap + 15) / 16 * 16;  // tile16: whole 16-row blocks
    float* nrows = nullptr;
    int32_t* ntags = nullptr;
    hipStream_t st = idx->eng->stream;
    const bool want_f32 = idx->dtype == RASS_F32;                  // a bf16 index holds the bf16 slab ONLY
    const bool want_b16 = idx->dtype == RASS_BF16 || idx->prefilter == 1;
    const bool want_i8 = idx->prefilter == 2;
    const size_t elem = want_f32 ? sizeof(float) : 2;
    void* nmain = nullptr;  // the dtype's own slab
    hipError_t e = hipMalloc(&nmain, (size_t)cap * idx->stride * elem);
    if (e != hipSuccess) {
        // retry with the exact size before giving up
        cap = (need_rows + 15) / 16 * 16;
        e = hipMalloc(&nmain, (size_t)cap * idx->stride * elem);
        if (e != hipSuccess) return fail(RASS_ERR_OOM, "index grow: hipMalloc of corpus slab failed");
    }
    if (want_f32) nrows = static_cast<float*>(nmain);
    int64_t* ngid = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&ntags), (size_t)cap * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ngid), (size_t)cap * sizeof(int64_t));
    if (e != hipSuccess) {
        (void)hipFree(nmain);
        if (ntags) (void)hipFree(ntags);
        return fail(RASS_ERR_OOM, "index grow: hipMalloc of tag / id arrays failed");
    }
    // rows of a block past the last appended one must read as finite zeros (they are masked,
    // never ranked): zero the part of the slab the copy below does not overwrite
    const int64_t used_rows = (idx->rows + 15) / 16 * 16;
    unsigned short* nb16 = want_f32 ? nullptr : static_cast<unsigned short*>(nmain);
    signed char* ni8 = nullptr;
    float* nscale = nullptr;
    auto copy_over = [&]() -> hipError_t {
        hipError_t c = hipSuccess;
        if (want_f32) {
            c = hipMemsetAsync(nrows + used_rows * idx->stride, 0, (size_t)(cap - used_rows) * idx->stride * sizeof(float), st);
            if (c == hipSuccess && idx->rows > 0)
                c = hipMemcpyAsync(nrows, idx->d_rows, (size_t)used_rows * idx->stride * sizeof(float),
                                   hipMemcpyDeviceToDevice, st);
        }
        if (c == hipSuccess && idx->rows > 0)
            c = hipMemcpyAsync(ntags, idx->d_tags, (size_t)idx->rows * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
        if (c == hipSuccess && idx->rows > 0)
            c = hipMemcpyAsync(ngid, idx->d_gid, (size_t)idx->rows * sizeof(int64_t), hipMemcpyDeviceToDevice, st);
        if (c == hipSuccess && want_b16) {
            if (want_f32) c = hipMalloc(reinterpret_cast<void**>(&nb16), (size_t)cap * idx->stride * 2);
            if (c == hipSuccess) c = hipMemsetAsync(nb16, 0, (size_t)cap * idx->stride * 2, st);
            if (c == hipSuccess && idx->rows > 0)
                c = hipMemcpyAsync(nb16, idx->d_rows_bf16, (size_t)used_rows * idx->stride * 2,
                                   hipMemcpyDeviceToDevice, st);
        }
        if (c == hipSuccess && want_i8) {
            c = hipMalloc(reinterpret_cast<void**>(&ni8), (size_t)cap * idx->stride_i8);
            if (c == hipSuccess) c = hipMalloc(reinterpret_cast<void**>(&nscale), (size_t)cap * sizeof(float));
            if (c == hipSuccess) c = hipMemsetAsync(ni8, 0, (size_t)cap * idx->stride_i8, st);
            if (c == hipSuccess) c = hipMemsetAsync(nscale, 0, (size_t)cap * sizeof(float), st);
            if (c == hipSuccess && idx->rows > 0)
                c = hipMemcpyAsync(ni8, idx->d_rows_i8, (size_t)used_rows * idx->stride_i8, hipMemcpyDeviceToDevice, st);
            if (c == hipSuccess && idx->rows > 0)
                c = hipMemcpyAsync(nscale, idx->d_row_scale, (size_t)used_rows * sizeof(float), hipMemcpyDeviceToDevice, st);
        }
        if (c == hipSuccess) c = hipStreamSynchronize(st);
        return c;
    };
    e = copy_over();
    if (e != hipSuccess) {  // nothing of the old index was touched: release the new allocations and report
        (void)hipStreamSynchronize(st);
        (void)hipFree(nmain);
        (void)hipFree(ntags);
        (void)hipFree(ngid);
        if (nb16 && want_f32) (void)hipFree(nb16);
        if (ni8) (void)hipFree(ni8);
        if (nscale) (void)hipFree(nscale);
        return fail(e == hipErrorOutOfMemory ? RASS_ERR_OOM : RASS_ERR_HIP,
                    std::string("index grow failed: ") + hipGetErrorString(e));
    }
    if (idx->d_rows) (void)hipFree(idx->d_rows);
    if (idx->d_tags) (void)hipFree(idx->d_tags);
    if (idx->d_gid) (void)hipFree(idx->d_gid);
    if (idx->d_rows_bf16) (void)hipFree(idx->d_rows_bf16);
    if (idx->d_rows_i8) (void)hipFree(idx->d_rows_i8);
    if (idx->d_row_scale) (void)hipFree(idx->d_row_scale);
    idx->d_rows_i8 = ni8;
    idx->d_row_scale = nscale;
    idx->d_rows = nrows;
    idx->d_tags = ntags;
    idx->d_gid = ngid;
    idx->d_rows_bf16 = nb16;
    idx->capacity = cap;
    return RASS_OK;
}

// XCD skew of the scan's tile order (kernels.h ScanArgs::xcd_skew; measured in
// scripts/microbench/scan_tail.hip and with bench.py).  RASS_SCAN_XCD_SKEW="a" or "a,b" overrides
// the defaults for query batches <= 16 / > 16 (0 = plain round-robin).
int scan_xcd_skew(int nq) {
    struct Skew {
        int b16 = 4, b32 = 0;  // bench.py sweeps: B<=16 603 -> 582 us at skew 4; no gain at B=32 (MFMA/power-bound)
        Skew() {
            if (const char* e = getenv("RASS_SCAN_XCD_SKEW")) {
                int a = 0, b = 0;
                const int n = sscanf(e, "%d,%d", &a, &b);
                if (n == 1) b = a;
                if (n >= 1 && a >= 0 && b >= 0 && a <= 4096 && b <= 4096) b16 = a, b32 = b;
            }
        }
    };
    static const Skew skew;  // C++11: initialised once, thread-safe
    return nq <= 16 ? skew.b16 : skew.b32;
}

// The sample floor (ScanArgs::sample_best): before a large flat scan with more than 16 queries, the first tile pair of
// every workgroup (64 * grid rows, 16,384 on MI355X) is scanned on its own, keeping only each workgroup's best score
// per query, and the k-th largest of those becomes the big scan's floor: k different rows reach it, so the final
// k-th best does too.  A row of the slab ranks above that floor with probability ~k / 16,384, so a 1M-row scan feeds
// ~600 candidates per query to the sorted insertion instead of ~18,000 (256 lists x ~70), for one short extra launch
// (the sample scan, which does no sorted insertion; the selection runs in the big scan's prologue).  Only worth it where the insertion is on the critical path: with <= 16 queries the scan
// is HBM-bound and the ranking hides under the loads.  RASS_SCAN_SAMPLE_FLOOR=0 switches it off and =force lowers
// the size threshold to twice the sample (A/B and tests; results are identical either way; read at every launch so
// that one process can compare the settings).
int64_t scan_sample_floor_min_share() {  // the slab must hold at least this many samples; 0 = never sample
    const char* e = getenv("RASS_SCAN_SAMPLE_FLOOR");
    if (e && e[0] == '0') return 0;
    if (e && e[0] == 'f') return 2;
    return 32;
}

int scan_launch(const float* d_corpus, int64_t n_rows, int64_t stride, const int32_t* d_row_tag,
                const float* d_queries, int q_dim, int64_t q_stride, int nq, const int32_t* d_q_filter, int k,
                int64_t id_base, float* d_out_scores, int64_t* d_out_ids, unsigned char* ws, size_t ws_bytes,
                int n_cus, hipStream_t st, rass_engine* timing = nullptr, const IvfPlan* plan = nullptr,
                const int64_t* id_map = nullptr, const ScanExt* ext = nullptr, bool queries_prepared = false) {
    if (nq < 1 || nq > RASS_MAX_QBATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, RASS_MAX_QBATCH]");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    if (n_rows < 0 || n_rows > 0x7fffffc0LL) return fail(RASS_ERR_INVALID, "n_rows out of range for one scan");
    if (!rass::scan_supported_stride(stride) || stride > kMaxStride)
        return fail(RASS_ERR_UNSUPPORTED, kStrideMsg);
    if (q_dim > stride) return fail(RASS_ERR_INVALID, "dim exceeds row_stride");
    if (stride > kNarrowStride) {
        if (plan) return fail(RASS_ERR_UNSUPPORTED, "IVF needs dim <= 1024");
        if (nq > 16) {
            // the wide-row kernel answers 16 queries per launch (its query fragments fill the registers): two launches,
            // one after the other on the stream (they share the workspace)
            ScanExt lo, hi;
            if (ext) {
                lo = *ext;
                hi.d_q_mask = ext->d_q_mask ? ext->d_q_mask + 16 : nullptr;
                hi.d_after_s = ext->d_after_s ? ext->d_after_s + 16 : nullptr;
                hi.d_after_i = ext->d_after_i ? ext->d_after_i + 16 : nullptr;
            }
            int rc = scan_launch(d_corpus, n_rows, stride, d_row_tag, d_queries, q_dim, q_stride, 16, d_q_filter, k, id_base,
                                 d_out_scores, d_out_ids, ws, ws_bytes, n_cus, st, timing, nullptr, id_map, ext ? &lo : nullptr);
            if (rc != RASS_OK) return rc;
            return scan_launch(d_corpus, n_rows, stride, d_row_tag, d_queries + 16 * q_stride, q_dim, q_stride, nq - 16,
                               d_q_filter ? d_q_filter + 16 : nullptr, k, id_base, d_out_scores + (int64_t)16 * k,
                               d_out_ids + (int64_t)16 * k, ws, ws_bytes, n_cus, st, timing, nullptr, id_map,
                               ext ? &hi : nullptr);
        }
    }
    const ScratchLayout L = scratch_layout(nq, k);
    if (ws == nullptr || ws_bytes < L.total) return fail(RASS_ERR_INVALID, "scan workspace too small");
    if ((reinterpret_cast<uintptr_t>(d_corpus) & 15) != 0) return fail(RASS_ERR_INVALID, "corpus not 16-B aligned");

    float* q_padded = reinterpret_cast<float*>(ws + L.q_padded);
    float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
    int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
    const int nq_pad = nq <= 16 ? 16 : 32;

    // a4 on the query side (reference app/main.py:1536-1537), written zero-padded.  (`queries_prepared`: the workspace
    // already holds these very queries normalised at this stride — the fine scan of an IVF probe right after its coarse scan.)
    if (!queries_prepared)
        HIP_TRY(rass::launch_normalize_rows_f32(d_queries, q_stride, q_padded, stride, nq, q_dim, st, nq_pad));

    // IVF: the number of work tiles is only known on the device; size the grid by the slab
    const int64_t n_tiles = plan ? plan->max_tiles : (n_rows + 31) / 32;
    int grid = (int)std::min<int64_t>(std::max<int64_t>(n_tiles, 1), std::min(n_cus, kMaxGrid));
    if ((int64_t)grid * k > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / k;

    rass::ScanArgs a;
    a.corpus = d_corpus;
    a.row_tag = d_row_tag;
    a.q_padded = q_padded;
    a.q_filter = d_q_filter;
    a.part_scores = part_scores;
    a.part_ids = part_ids;
    a.row_stride = stride;
    a.id_base = id_base;
    a.n_rows = (int)n_rows;
    a.nq = nq;
    a.k = k;
    // one workgroup per CU on every XCD: only then does blockIdx parity = XCD parity
    a.xcd_skew = (grid == n_cus && grid % 8 == 0) ? scan_xcd_skew(nq) : 0;
    if (ext) {
        a.q_filter_mask = ext->d_q_mask;
        a.q_after_score = ext->d_after_s;
        a.q_after_id = ext->d_after_i;
    }
    if (plan) {
        a.work_tile = plan->work_tile;
        a.work_rows = plan->work_rows;
        a.work_mask = plan->work_mask;
        a.n_work = plan->n_work;
    }
    const int64_t sample_rows = (int64_t)64 * grid;
    const int64_t min_share = scan_sample_floor_min_share();
    if (min_share > 0 && !plan && nq > 16 && grid <= rass::kMaxSampleGroups && n_rows >= min_share * sample_rows) {
        // same filters, same continuation bound, same id space: only the row count differs
        rass::ScanArgs s = a;
        s.n_rows = (int)sample_rows;
        s.xcd_skew = 0;
        s.sample_pass = true;
        s.part_scores = reinterpret_cast<float*>(ws + L.sample_best);
        s.part_ids = nullptr;
        HIP_TRY(rass::launch_scan_topk_f32(s, grid, st));
        a.sample_best = s.part_scores;
        a.sample_groups = grid;
    }
    const bool timed = timing && timing->ev_on && (size_t)(2 * timing->ev_used + 1) < timing->ev_pool.size();
    if (timed) HIP_TRY(hipEventRecord(timing->ev_pool[2 * timing->ev_used], st));
    HIP_TRY(rass::launch_scan_topk_f32(a, grid, st));
    if (timed) {
        HIP_TRY(hipEventRecord(timing->ev_pool[2 * timing->ev_used + 1], st));
        timing->ev_used += 1;
    }
    HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, nq, k, d_out_scores, d_out_ids, st, id_map));
    return RASS_OK;
}

// The sample floor of the int8 and bf16 scans (scan_i8.hip, scan_bf16.hip): worth a short extra launch when the slab is many samples long.
// RASS_I8_SAMPLE_FLOOR=0 turns it off (the A/B; results do not depend on it).
bool i8_sample_floor(int64_t rows, int grid) {
    const char* e = getenv("RASS_I8_SAMPLE_FLOOR");   // read per call: the tests switch it
    if (e && atoi(e) == 0) return false;
    return grid <= rass::kMaxSampleGroups && rows >= (int64_t)8 * 64 * grid;
}

// A bf16 corpus (RASS_BF16): the bf16 scan IS the search — normalise the queries, round them to bf16,
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation over the bf16 slab, per-workgroup top-k, merge.
int bf16_scan_launch(rass_index* idx, const float* d_queries, int nq, const int32_t* d_q_filter, int k, int64_t id_base,
                     float* d_out_scores, int64_t* d_out_ids, const int32_t* d_row_tag, rass_engine* eng, hipStream_t st,
                     const int64_t* id_map, const ScanExt* ext = nullptr) {
    if (nq < 1 || nq > RASS_MAX_QBATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, RASS_MAX_QBATCH]");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    const int64_t stride = idx->stride;
    const int64_t rows = idx->rows.load(std::memory_order_acquire);
    const ScratchLayout L = scratch_layout(RASS_MAX_QBATCH, RASS_MAX_K);
    unsigned char* ws = eng->d_scratch;
    float* q_padded = reinterpret_cast<float*>(ws + L.q_padded);
    float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
    int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
    unsigned short* q_bf16 = reinterpret_cast<unsigned short*>(ws + L.q_bf16);
    const int nq_pad = nq <= 16 ? 16 : 32;
    HIP_TRY(rass::launch_normalize_rows_f32(d_queries, idx->dim, q_padded, stride, nq, idx->dim, st, nq_pad));
    HIP_TRY(rass::launch_queries_to_bf16(q_padded, q_bf16, (int64_t)nq_pad * stride, st));
    const int64_t n_tiles = (rows + 63) / 64;
    int grid = (int)std::min<int64_t>(std::max<int64_t>(n_tiles, 1), std::min(eng->n_cus, kMaxGrid));
    if ((int64_t)grid * k > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / k;
    rass::ScanBf16Args a;
    a.corpus = idx->d_rows_bf16 ? idx->d_rows_bf16 : reinterpret_cast<const unsigned short*>(eng->d_scratch);
    a.row_tag = d_row_tag;
    a.q_bf16 = q_bf16;
    a.q_filter = d_q_filter;
    a.part_scores = part_scores;
    a.part_ids = part_ids;
    a.row_stride = stride;
    a.n_rows = (int)rows;
    a.nq = nq;
    a.k = k;
    a.id_base = id_map ? 0 : id_base;
    if (ext) {
        a.q_filter_mask = ext->d_q_mask;
        a.q_after_score = ext->d_after_s;
        a.q_after_id = ext->d_after_i;
    }
    // the sample floor pays where many candidates are kept (k = 10: 96.1 k queries/s without it, 89.9 k with its extra launch;
    // the prefilter's 32 candidates: 85.5 k -> 92.4 k); RASS_I8_SAMPLE_FLOOR=0: the A/B for both scans
    if (!ext && k >= 24 && i8_sample_floor(rows, grid)) {
        rass::ScanBf16Args sa = a;
        sa.n_rows = 64 * grid;
        sa.k = 1;
        sa.part_scores = reinterpret_cast<float*>(ws + L.sample_best);
        sa.part_ids = nullptr;
        HIP_TRY(rass::launch_scan_bf16_topk(sa, grid, st));
        a.sample_best = sa.part_scores;
        a.sample_groups = grid;
    }
    const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
    if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
    HIP_TRY(rass::launch_scan_bf16_topk(a, grid, st));
    if (timed) {
        HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
        eng->ev_used += 1;
    }
    HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, nq, k, d_out_scores, d_out_ids, st, id_map));
    return RASS_OK;
}

// Prefilter mode: bf16 (mode 1) or int8 (mode 2) candidate scan (32 per query) -> merge -> exact fp32 re-rank.
// d_cand_scores / d_cand_rows (optional, [nq][32]): the merged candidate lists as well (rass_index_candidates_device).
// d_q_filter_mask: masked tag compare ((tag & mask) == filter); id_map: the id reported for row r (a shard's caller-assigned
// global ids, ascending with the row: the tie order is unchanged) instead of id_base + r.
int prefilter_launch(rass_index* idx, const float* d_queries, int nq, const int32_t* d_q_filter, int k,
                     int64_t id_base, float* d_out_scores, int64_t* d_out_ids, const int32_t* d_row_tag,
                     rass_engine* eng, hipStream_t st, float* d_cand_scores = nullptr, int64_t* d_cand_rows = nullptr,
                     const int32_t* d_q_filter_mask = nullptr, const int64_t* id_map = nullptr) {
    if (nq < 1 || nq > RASS_MAX_QBATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, RASS_MAX_QBATCH]");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    const int64_t stride = idx->stride;
    const ScratchLayout L = scratch_layout(RASS_MAX_QBATCH, RASS_MAX_K);
    unsigned char* ws = eng->d_scratch;
    float* q_padded = reinterpret_cast<float*>(ws + L.q_padded);
    float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
    int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
    unsigned short* q_bf16 = reinterpret_cast<unsigned short*>(ws + L.q_bf16);   // int8 queries live here too (half the bytes)
    float* cand_scores = d_cand_scores ? d_cand_scores : reinterpret_cast<float*>(ws + L.cand_scores);
    int64_t* cand_ids = d_cand_rows ? d_cand_rows : reinterpret_cast<int64_t*>(ws + L.cand_ids);
    const int nq_pad = nq <= 16 ? 16 : 32;
    const int kc = RASS_MAX_K;  // candidates per query
    HIP_TRY(rass::launch_normalize_rows_f32(d_queries, idx->dim, q_padded, stride, nq, idx->dim, st, nq_pad));
    const int64_t n_tiles = (idx->rows + 63) / 64;
    int grid = (int)std::min<int64_t>(std::max<int64_t>(n_tiles, 1), std::min(eng->n_cus, kMaxGrid));
    if ((int64_t)grid * kc > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / kc;
    const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
    if (idx->prefilter == 2) {
        HIP_TRY(rass::launch_queries_to_i8(q_padded, q_bf16, nq_pad, stride, idx->stride_i8, st));
        rass::ScanI8Args a;
        a.corpus = idx->d_rows_i8;
        a.row_scale = idx->d_row_scale;
        a.row_tag = d_row_tag;
        a.q_i8 = reinterpret_cast<const signed char*>(q_bf16);
        a.q_filter = d_q_filter;
        a.q_filter_mask = d_q_filter_mask;
        a.part_scores = part_scores;
        a.part_ids = part_ids;
        a.row_stride = idx->stride_i8;
        a.n_rows = (int)idx->rows;
        a.nq = nq;
        a.k = kc;
        if (i8_sample_floor(idx->rows, grid)) {   // the sample launch: the first 64 * grid rows, the best score per workgroup
            rass::ScanI8Args sa = a;
            sa.n_rows = 64 * grid;
            sa.k = 1;
            sa.part_scores = reinterpret_cast<float*>(ws + L.sample_best);
            sa.part_ids = nullptr;
            HIP_TRY(rass::launch_scan_i8_topk(sa, grid, st));
            a.sample_best = sa.part_scores;
            a.sample_groups = grid;
        }
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        HIP_TRY(rass::launch_scan_i8_topk(a, grid, st));
    } else {
        HIP_TRY(rass::launch_queries_to_bf16(q_padded, q_bf16, (int64_t)nq_pad * stride, st));
        rass::ScanBf16Args a;
        a.corpus = idx->d_rows_bf16;
        a.row_tag = d_row_tag;
        a.q_bf16 = q_bf16;
        a.q_filter = d_q_filter;
        a.q_filter_mask = d_q_filter_mask;
        a.part_scores = part_scores;
        a.part_ids = part_ids;
        a.row_stride = stride;
        a.n_rows = (int)idx->rows;
        a.nq = nq;
        a.k = kc;
        if (!d_q_filter_mask && i8_sample_floor(idx->rows, grid)) {
            rass::ScanBf16Args sa = a;
            sa.n_rows = 64 * grid;
            sa.k = 1;
            sa.part_scores = reinterpret_cast<float*>(ws + L.sample_best);
            sa.part_ids = nullptr;
            HIP_TRY(rass::launch_scan_bf16_topk(sa, grid, st));
            a.sample_best = sa.part_scores;
            a.sample_groups = grid;
        }
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        HIP_TRY(rass::launch_scan_bf16_topk(a, grid, st));
    }
    if (timed) {
        HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
        eng->ev_used += 1;
    }
    HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, nq, kc, cand_scores, cand_ids, st));
    HIP_TRY(rass::launch_rerank_f32(idx->d_rows, stride, q_padded, cand_ids, nq, kc, k, id_map ? 0 : id_base, d_out_scores,
                                    d_out_ids, st, 0, 0, id_map));
    return RASS_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------ IVF (K9)
struct rass_ivf {
    rass_engine* eng = nullptr;
    int dim = 0, nlist = 0;
    int64_t stride = 0, rows = 0, slab_rows = 0, total_tiles = 0;
    int dtype = RASS_F32;           // RASS_F32: d_slab (tile16, 32-row tiles) | RASS_BF16: d_slab_b16 (tile16b, 64-row tiles)
    int tile_rows = 32;             // rows per plan tile = the fine scan kernel's tile
    float* d_slab = nullptr;        // tile16, lists contiguous, each starting on a 32-row tile
    unsigned short* d_slab_b16 = nullptr;  // bf16 slab: the rows rounded to bf16, lists starting on 64-row tiles
    // RASS_I8: d_slab (fp32, lists on 64-row tiles: read by the exact re-rank only) + its int8 copy (tile16i) and row scales:
    // the fine scan keeps 32 int8 candidates per query, the re-rank rescores them exactly and returns the best k <= 16
    signed char* d_slab_i8 = nullptr;
    float* d_slab_scale = nullptr;
    int64_t stride_i8 = 0;
    float* d_cand_scores = nullptr;        // [32][32] candidates of one launch group (RASS_I8)
    int64_t* d_cand_rows = nullptr;
    int32_t* d_tags = nullptr;      // [slab_rows] permuted row tags (0 on padding)
    int64_t* d_ids = nullptr;       // [slab_rows] source row id, -1 on padding
    float* d_centroids = nullptr;   // tile16 slab of nlist normalised centroids
    int32_t *d_list_tile0 = nullptr, *d_list_len = nullptr;
    int32_t *d_work_tile = nullptr, *d_work_rows = nullptr, *d_n_work = nullptr;
    uint32_t* d_work_mask = nullptr;
    int64_t* d_scanned = nullptr;   // rows touched by the last fine scan
    float* d_probe_scores = nullptr;  // [32][32]
    int64_t* d_probe_ids = nullptr;   // [32][32]
    uint32_t* d_tau = nullptr;        // [32] nprobe > 32: per-query threshold keys
    uint32_t* d_list_mask = nullptr;  // [nlist] nprobe > 32: probe masks from the score matrix
    bool any_tags = false;
    // IVF + flat delta (rass_ivf_search_delta*): the IVF covers source rows [0, src_rows); rows the source index took
    // afterwards are scanned exactly from its own slab and merged with the probe's list
    int64_t src_rows = 0;
    std::vector<int32_t> pos_of;      // host: slab position of source row r (< src_rows), -1 = not in the slab (tombstoned)
    float* d_pair_scores = nullptr;   // [2][32][32] the probe's list and the delta scan's list of one launch group
    int64_t* d_pair_ids = nullptr;
    unsigned char* d_batch = nullptr; // rass_ivf_search_device_batch: queries, coarse / fine lists and work lists of <= 32 groups
    size_t batch_bytes = 0;
};

extern "C" void rassint_set_last_error(const char* msg) { g_err = msg ? msg : ""; }

extern "C" {

int rass_abi_version(void) { return RASS_ABI_VERSION; }

const char* rass_last_error(void) { return g_err.c_str(); }

int rass_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RASS_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int rass_engine_create(int device, int dim, rass_engine_t** out) {
    if (out == nullptr) return fail(RASS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (dim < 1 || pad_stride(dim) > kMaxStride)
        return fail(RASS_ERR_UNSUPPORTED, "dim must be in [1, 2048] for the fused scan");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RASS_ERR_INVALID, "no such HIP device");
    rass_engine* eng = new (std::nothrow) rass_engine();
    if (!eng) return fail(RASS_ERR_OOM, "host allocation failed");
    eng->device = device;
    eng->dim = dim;
    auto init = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        eng->n_cus = device_cus(device);
        HIP_TRY(hipStreamCreateWithFlags(&eng->own_stream, hipStreamNonBlocking));
        eng->stream = eng->own_stream;
        eng->scratch_bytes = scratch_layout(RASS_MAX_QBATCH, RASS_MAX_K).total;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_scratch), eng->scratch_bytes));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_qraw), (size_t)RASS_MAX_QBATCH * dim * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_qfilter), RASS_MAX_QBATCH * sizeof(int32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_out_scores), RASS_MAX_QBATCH * RASS_MAX_K * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_out_ids), RASS_MAX_QBATCH * RASS_MAX_K * sizeof(int64_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_stage), (size_t)kStageRows * dim * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_stage_tags), (size_t)kStageRows * sizeof(int32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_qmask), RASS_MAX_QBATCH * sizeof(int32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_after_s), RASS_MAX_QBATCH * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_after_i), RASS_MAX_QBATCH * sizeof(int64_t)));
        eng->slots.resize(kHostSlots);
        for (HostSlot& sl : eng->slots) {
            // one pinned block per slot: [out_i 32x32 i64][after_i 32 i64][scanned i64][q 32xdim f32][out_s][after_s][filter][mask]
            const size_t B = RASS_MAX_QBATCH, K = RASS_MAX_K;
            const size_t bytes = B * K * 8 + B * 8 + 8 + B * (size_t)dim * 4 + B * K * 4 + B * 4 + B * 4 + B * 4;
            HIP_TRY(hipHostMalloc(&sl.base, bytes, hipHostMallocDefault));
            unsigned char* p = static_cast<unsigned char*>(sl.base);
            sl.h_out_i = reinterpret_cast<int64_t*>(p), p += B * K * 8;
            sl.h_after_i = reinterpret_cast<int64_t*>(p), p += B * 8;
            sl.h_scanned = reinterpret_cast<int64_t*>(p), p += 8;
            sl.h_q = reinterpret_cast<float*>(p), p += B * (size_t)dim * 4;
            sl.h_out_s = reinterpret_cast<float*>(p), p += B * K * 4;
            sl.h_after_s = reinterpret_cast<float*>(p), p += B * 4;
            sl.h_filter = reinterpret_cast<int32_t*>(p), p += B * 4;
            sl.h_mask = reinterpret_cast<int32_t*>(p), p += B * 4;
            HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        }
        return RASS_OK;
    };
    const int rc = init();
    if (rc != RASS_OK) {
        const std::string why = g_err;  // destroy() must not lose the reason
        rass_engine_destroy(eng);
        return fail(rc, why);
    }
    *out = eng;
    return RASS_OK;
}

void rass_engine_destroy(rass_engine_t* eng) {
    if (!eng) return;
    (void)hipSetDevice(eng->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : eng->indices) {
        rass_index* idx = kv.second;
        if (idx->d_rows) (void)hipFree(idx->d_rows);
        if (idx->d_tags) (void)hipFree(idx->d_tags);
        if (idx->d_gid) (void)hipFree(idx->d_gid);
        if (idx->d_rows_bf16) (void)hipFree(idx->d_rows_bf16);
        if (idx->d_rows_i8) (void)hipFree(idx->d_rows_i8);
        if (idx->d_row_scale) (void)hipFree(idx->d_row_scale);
        delete idx;
    }
    eng->indices.clear();
    (void)hipFree(eng->d_scratch);
    if (eng->d_batch) (void)hipFree(eng->d_batch);
    (void)hipFree(eng->d_qraw);
    (void)hipFree(eng->d_qfilter);
    (void)hipFree(eng->d_out_scores);
    (void)hipFree(eng->d_out_ids);
    (void)hipFree(eng->d_stage);
    (void)hipFree(eng->d_stage_tags);
    if (eng->d_stage_t16) (void)hipFree(eng->d_stage_t16);
    for (void* p : {(void*)eng->d_mw_tile, (void*)eng->d_mw_rows, (void*)eng->d_mw_n, (void*)eng->d_mw_mask,
                    (void*)eng->d_mw_base, (void*)eng->d_mw_tags})
        if (p) (void)hipFree(p);
    (void)hipFree(eng->d_qmask);
    (void)hipFree(eng->d_after_s);
    (void)hipFree(eng->d_after_i);
    for (HostSlot& sl : eng->slots) {
        if (sl.base) (void)hipHostFree(sl.base);
        if (sl.h_items) (void)hipHostFree(sl.h_items);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (hipEvent_t e : eng->ev_pool) (void)hipEventDestroy(e);
    if (eng->own_stream) (void)hipStreamDestroy(eng->own_stream);
    delete eng;
}

int rass_engine_dim(const rass_engine_t* eng) { return eng ? eng->dim : fail(RASS_ERR_INVALID, "engine is NULL"); }
int rass_engine_device(const rass_engine_t* eng) {
    return eng ? eng->device : fail(RASS_ERR_INVALID, "engine is NULL");
}

int rass_engine_set_stream(rass_engine_t* eng, void* stream) {
    if (!eng) return fail(RASS_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> lk(eng->mu);
    HIP_TRY(hipSetDevice(eng->device));
    HIP_TRY(hipStreamSynchronize(eng->stream));
    eng->stream = reinterpret_cast<hipStream_t>(stream);
    return RASS_OK;
}

int rass_engine_reset_stream(rass_engine_t* eng) {
    if (!eng) return fail(RASS_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> lk(eng->mu);
    HIP_TRY(hipSetDevice(eng->device));
    HIP_TRY(hipStreamSynchronize(eng->stream));
    eng->stream = eng->own_stream;
    return RASS_OK;
}

void* rass_engine_get_stream(rass_engine_t* eng) { return eng ? reinterpret_cast<void*>(eng->stream) : nullptr; }

int rass_engine_synchronize(rass_engine_t* eng) {
    if (!eng) return fail(RASS_ERR_INVALID, "engine is NULL");
    HIP_TRY(hipSetDevice(eng->device));
    HIP_TRY(hipStreamSynchronize(eng->stream));
    return RASS_OK;
}

int rass_index_open(rass_engine_t* eng, const char* name, rass_dtype dtype, int64_t initial_capacity_rows,
                    rass_index_t** out) {
    if (!eng || !name || !out) return fail(RASS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (dtype != RASS_F32 && dtype != RASS_BF16) return fail(RASS_ERR_INVALID, "unknown corpus dtype");
    if (dtype == RASS_BF16 && pad128(eng->dim) % 256 != 0)
        return fail(RASS_ERR_UNSUPPORTED, "a bf16 corpus needs dim padded to a multiple of 256 (the bf16 scan's K split)");
    if (dtype == RASS_BF16 && eng->dim > kNarrowStride)
        return fail(RASS_ERR_UNSUPPORTED, "a bf16 corpus needs dim <= 1024 (wide rows are served by the fp32 scan only)");
    std::lock_guard<std::mutex> lk(eng->mu);
    auto it = eng->indices.find(name);
    if (it != eng->indices.end()) {
        if (it->second->dtype != dtype) return fail(RASS_ERR_INVALID, "index exists with another dtype");
        *out = it->second;
        return RASS_OK;
    }
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    rass_index* idx = new (std::nothrow) rass_index();
    if (!idx) return fail(RASS_ERR_OOM, "host allocation failed");
    idx->eng = eng;
    idx->name = name;
    idx->dtype = dtype;
    idx->dim = eng->dim;
    idx->stride = pad_stride(eng->dim);
    if (initial_capacity_rows > 0) {
        rc = index_reserve(idx, initial_capacity_rows);
        if (rc != RASS_OK) {
            delete idx;
            return rc;
        }
    }
    eng->indices[name] = idx;
    *out = idx;
    return RASS_OK;
}

int rass_index_drop(rass_engine_t* eng, const char* name) {
    if (!eng || !name) return fail(RASS_ERR_INVALID, "NULL argument");
    std::unique_lock<std::mutex> lk(eng->mu);
    auto it = eng->indices.find(name);
    if (it == eng->indices.end()) return fail(RASS_ERR_NOT_FOUND, "no such index");
    HIP_TRY(hipSetDevice(eng->device));
    rass_index* idx = it->second;
    eng->indices.erase(it);
    lk.unlock();
    {   // an add / delete / get_row / save that already holds the index runs to its end first (lock order
        // everywhere: index mutex, then engine mutex).  Using the handle AFTER drop returns is the caller's bug.
        std::lock_guard<std::mutex> ilk(idx->mu);
        std::lock_guard<std::mutex> elk(eng->mu);
        (void)hipStreamSynchronize(eng->stream);
        if (idx->d_rows) (void)hipFree(idx->d_rows);
        if (idx->d_tags) (void)hipFree(idx->d_tags);
        if (idx->d_gid) (void)hipFree(idx->d_gid);
        if (idx->d_rows_bf16) (void)hipFree(idx->d_rows_bf16);
        if (idx->d_rows_i8) (void)hipFree(idx->d_rows_i8);
        if (idx->d_row_scale) (void)hipFree(idx->d_row_scale);
        idx->d_rows_i8 = nullptr;
        idx->d_row_scale = nullptr;
        idx->d_rows = nullptr;
        idx->d_tags = nullptr;
        idx->d_gid = nullptr;
        idx->d_rows_bf16 = nullptr;
    }
    delete idx;
    return RASS_OK;
}

int rass_index_set_prefilter(rass_index_t* idx, int enable) {
    if (!idx) return fail(RASS_ERR_INVALID, "index is NULL");
    if (enable < 0 || enable > 2) return fail(RASS_ERR_INVALID, "prefilter mode must be 0 (off), 1 (bf16) or 2 (int8)");
    rass_engine* eng = idx->eng;
    std::lock_guard<std::mutex> lk(idx->mu);
    std::lock_guard<std::mutex> elk(eng->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    hipStream_t st = eng->stream;
    if (enable && idx->dtype == RASS_BF16) return fail(RASS_ERR_UNSUPPORTED, "a bf16 corpus IS the bf16 scan: no prefilter mode");
    if (enable == idx->prefilter) return RASS_OK;
    if (enable == 1 && idx->stride % 256 != 0) return fail(RASS_ERR_UNSUPPORTED, "prefilter needs dim padded to a multiple of 256");
    if (enable == 1 && idx->stride > kNarrowStride) return fail(RASS_ERR_UNSUPPORTED, "the bf16 prefilter needs dim <= 1024 (wide rows: int8 candidates or the fp32 flat scan)");
    // leave the current mode (a switch between the two candidate copies goes through "off")
    HIP_TRY(hipStreamSynchronize(st));
    if (idx->d_rows_bf16) (void)hipFree(idx->d_rows_bf16);
    if (idx->d_rows_i8) (void)hipFree(idx->d_rows_i8);
    if (idx->d_row_scale) (void)hipFree(idx->d_row_scale);
    idx->d_rows_bf16 = nullptr;
    idx->d_rows_i8 = nullptr;
    idx->d_row_scale = nullptr;
    idx->prefilter = 0;
    if (!enable) return RASS_OK;
    if (enable == 1 && idx->capacity > 0) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&idx->d_rows_bf16), (size_t)idx->capacity * idx->stride * 2));
        HIP_TRY(hipMemsetAsync(idx->d_rows_bf16, 0, (size_t)idx->capacity * idx->stride * 2, st));
        HIP_TRY(rass::launch_convert_tile16_bf16(idx->d_rows, idx->d_rows_bf16, idx->stride, 0, (idx->rows + 15) >> 4,
                                                 st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (enable == 2) {
        idx->stride_i8 = (idx->stride + 511) / 512 * 512;
        if (idx->capacity > 0) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&idx->d_rows_i8), (size_t)idx->capacity * idx->stride_i8));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&idx->d_row_scale), (size_t)idx->capacity * sizeof(float)));
            HIP_TRY(hipMemsetAsync(idx->d_rows_i8, 0, (size_t)idx->capacity * idx->stride_i8, st));
            HIP_TRY(hipMemsetAsync(idx->d_row_scale, 0, (size_t)idx->capacity * sizeof(float), st));
            HIP_TRY(rass::launch_quantize_tile16_i8(idx->d_rows, idx->d_rows_i8, idx->d_row_scale, idx->stride, idx->stride_i8, 0,
                                                    (idx->rows + 15) >> 4, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    idx->prefilter = enable;
    return RASS_OK;
}

int rass_index_get_prefilter(const rass_index_t* idx) { return idx ? idx->prefilter : 0; }

int64_t rass_index_count(const rass_index_t* idx) { return idx ? idx->rows.load() - idx->deleted.load() : 0; }
int64_t rass_index_rows(const rass_index_t* idx) { return idx ? idx->rows.load() : (int64_t)0; }
int rass_index_dim(const rass_index_t* idx) { return idx ? idx->dim : fail(RASS_ERR_INVALID, "index is NULL"); }
int rass_index_dtype(const rass_index_t* idx) { return idx ? (int)idx->dtype : fail(RASS_ERR_INVALID, "index is NULL"); }
int rass_index_has_global_ids(const rass_index_t* idx) {
    return idx ? (idx->has_gid.load() ? 1 : 0) : fail(RASS_ERR_INVALID, "index is NULL");
}
int rass_index_row_stride(const rass_index_t* idx) {
    return idx ? (int)idx->stride : fail(RASS_ERR_INVALID, "index is NULL");
}

static int add_common(rass_index_t* idx, const float* vecs, const int32_t* tags, int64_t n, int normalize,
                      int64_t* first_row, bool device_src, int64_t first_global_id = -1) {
    if (!idx) return fail(RASS_ERR_INVALID, "index is NULL");
    if (n < 0 || (n > 0 && !vecs)) return fail(RASS_ERR_INVALID, "bad vecs / n");
    rass_engine* eng = idx->eng;
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    if (!device_src && tags) {
        for (int64_t i = 0; i < n; ++i)
            if (tags[i] < 0) return fail(RASS_ERR_INVALID, "row tags must be >= 0");
    }
    if (first_row) *first_row = idx->rows;
    if (n == 0) return RASS_OK;
    {
        std::lock_guard<std::mutex> elk(eng->mu);  // grow swaps pointers searches read
        rc = index_reserve(idx, idx->rows + n);
        if (rc != RASS_OK) return rc;
    }
    hipStream_t st = eng->stream;
    const int dim = idx->dim;
    for (int64_t done = 0; done < n;) {
        const int64_t m = std::min<int64_t>(kStageRows, n - done);
        int32_t* tdst = idx->d_tags + idx->rows + done;
        const float* src = vecs + done * dim;
        const float* dsrc = src;
        // host source: the staging buffer is ENGINE scratch, shared with the other indices (users) of
        // this engine and with get_row(s) / save — hold the engine lock for the chunk's round trip
        std::unique_lock<std::mutex> elk(eng->mu, std::defer_lock);
        if (!device_src) elk.lock();
        if (!device_src) {
            HIP_TRY(hipMemcpyAsync(eng->d_stage, src, (size_t)m * dim * sizeof(float), hipMemcpyHostToDevice, st));
            dsrc = eng->d_stage;
        }
        if (idx->dtype == RASS_BF16) {
            // normalise + pack into the fp32 staging slab at the destination's phase inside a 16-row block, then round
            // the chunk's rows (and only them) into the bf16 slab
            if (!elk.owns_lock()) elk.lock();
            if (!eng->d_stage_t16)
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_stage_t16), (size_t)(kStageRows + 32) * kMaxStride * sizeof(float)));
            const int64_t at = idx->rows + done, phase = at & 15;
            HIP_TRY(rass::launch_pack_rows_tile16(dsrc, dim, eng->d_stage_t16, idx->stride, phase, m, dim,
                                                  normalize ? 1 : 0, st));
            HIP_TRY(rass::launch_convert_tile16_bf16(eng->d_stage_t16, idx->d_rows_bf16, idx->stride, at >> 4,
                                                     (at + m + 15) >> 4, st, 0, at, at + m));
        } else
        HIP_TRY(rass::launch_pack_rows_tile16(dsrc, dim, idx->d_rows, idx->stride, idx->rows + done, m, dim,
                                              normalize ? 1 : 0, st));
        if (tags && device_src) {
            HIP_TRY(rass::launch_copy_tags_clamped(tdst, tags + done, m, st));
        } else if (tags) {
            HIP_TRY(hipMemcpyAsync(tdst, tags + done, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, st));
        } else {
            HIP_TRY(rass::launch_fill_i32(tdst, m, 0, st));
        }
        // the staging buffer is reused by the next chunk
        if (!device_src || idx->dtype == RASS_BF16) HIP_TRY(hipStreamSynchronize(st));
        done += m;
    }
    if (idx->prefilter == 1 && idx->dtype == RASS_F32)
        HIP_TRY(rass::launch_convert_tile16_bf16(idx->d_rows, idx->d_rows_bf16, idx->stride, idx->rows >> 4,
                                                 (idx->rows + n + 15) >> 4, st));
    if (idx->prefilter == 2)   // whole blocks: the earlier rows of a partially filled block quantise to the same bytes again
        HIP_TRY(rass::launch_quantize_tile16_i8(idx->d_rows, idx->d_rows_i8, idx->d_row_scale, idx->stride, idx->stride_i8,
                                                idx->rows >> 4, (idx->rows + n + 15) >> 4, st));
    // the id a search reports for these rows: their ordinal, or the caller's global ids (ascending with the
    // ordinal, so the (score desc, id asc) tie order inside the shard is the global one)
    HIP_TRY(rass::launch_iota_i64(idx->d_gid + idx->rows, n, first_global_id >= 0 ? first_global_id : idx->rows.load(), st));
    if (first_global_id >= 0 && first_global_id != idx->rows) idx->has_gid = true;
    if (tags) idx->has_tags = true;
    idx->rows += n;
    idx->host_deleted.resize((size_t)((idx->rows + 7) / 8), 0);
    return RASS_OK;
}

int rass_index_add(rass_index_t* idx, const float* vecs, const int32_t* tags, int64_t n, int normalize,
                   int64_t* first_row) {
    return add_common(idx, vecs, tags, n, normalize, first_row, false);
}

int rass_index_add_device(rass_index_t* idx, const float* d_vecs, const int32_t* d_tags, int64_t n, int normalize,
                          int64_t* first_row) {
    return add_common(idx, d_vecs, d_tags, n, normalize, first_row, true);
}

int rass_index_add_ex(rass_index_t* idx, const float* vecs, const int32_t* tags, int64_t n, int normalize,
                      int64_t first_global_id, int device_source, int64_t* first_row) {
    if (first_global_id < -1) return fail(RASS_ERR_INVALID, "first_global_id must be >= 0, or -1 for row ordinals");
    return add_common(idx, vecs, tags, n, normalize, first_row, device_source != 0, first_global_id);
}

int rass_index_delete(rass_index_t* idx, int64_t row) {
    if (!idx) return fail(RASS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (row < 0 || row >= idx->rows) return fail(RASS_ERR_NOT_FOUND, "row out of range");
    uint8_t& byte = idx->host_deleted[(size_t)(row >> 3)];
    const uint8_t bit = (uint8_t)(1u << (row & 7));
    if (byte & bit) return RASS_OK;  // idempotent
    int rc = set_device(idx->eng);
    if (rc != RASS_OK) return rc;
    {
        // A search holds eng->mu for its whole enqueue sequence (sample-floor pass, scan, further passes of a k > 32
        // search): without it this fill could land BETWEEN them — the floor was computed from k rows, one of which is
        // now a tombstone, and the scan would reject live rows below it and return fewer than k hits.  Lock order
        // idx->mu then eng->mu, as in add_common: a search sees a tombstone before its first pass or after its merge.
        std::lock_guard<std::mutex> elk(idx->eng->mu);
        HIP_TRY(rass::launch_fill_i32(idx->d_tags + row, 1, RASS_ROW_TAG_DELETED, idx->eng->stream));
    }
    byte |= bit;
    idx->deleted += 1;
    return RASS_OK;
}

int rass_index_get_row(rass_index_t* idx, int64_t row, float* out) {
    if (!idx || !out) return fail(RASS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (row < 0 || row >= idx->rows) return fail(RASS_ERR_NOT_FOUND, "row out of range");
    int rc = set_device(idx->eng);
    if (rc != RASS_OK) return rc;
    hipStream_t st = idx->eng->stream;
    {
        std::lock_guard<std::mutex> elk(idx->eng->mu);  // d_stage is shared engine scratch
        if (idx->dtype == RASS_BF16)
            HIP_TRY(rass::launch_unpack_rows_tile16b(idx->d_rows_bf16, idx->stride, row, 1, idx->dim, idx->eng->d_stage,
                                                     idx->dim, st));
        else
        HIP_TRY(rass::launch_unpack_rows_tile16(idx->d_rows, idx->stride, row, 1, idx->dim, idx->eng->d_stage,
                                                idx->dim, st));
        HIP_TRY(hipMemcpyAsync(out, idx->eng->d_stage, (size_t)idx->dim * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return RASS_OK;
}

int rass_index_get_rows(rass_index_t* idx, int64_t first_row, int64_t n, float* out) {
    if (!idx || (n > 0 && !out)) return fail(RASS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (first_row < 0 || n < 0 || first_row + n > idx->rows) return fail(RASS_ERR_NOT_FOUND, "rows out of range");
    int rc = set_device(idx->eng);
    if (rc != RASS_OK) return rc;
    hipStream_t st = idx->eng->stream;
    std::lock_guard<std::mutex> elk(idx->eng->mu);  // d_stage is shared engine scratch
    for (int64_t done = 0; done < n; done += kStageRows) {
        const int64_t m = std::min<int64_t>(kStageRows, n - done);
        if (idx->dtype == RASS_BF16)
            HIP_TRY(rass::launch_unpack_rows_tile16b(idx->d_rows_bf16, idx->stride, first_row + done, m, idx->dim,
                                                     idx->eng->d_stage, idx->dim, st));
        else
        HIP_TRY(rass::launch_unpack_rows_tile16(idx->d_rows, idx->stride, first_row + done, m, idx->dim,
                                                idx->eng->d_stage, idx->dim, st));
        HIP_TRY(hipMemcpyAsync(out + done * idx->dim, idx->eng->d_stage, (size_t)m * idx->dim * sizeof(float),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return RASS_OK;
}

namespace {

// One launch group (<= 32 queries) of a device search; the caller holds eng->mu and has set the device.
int search_device_group(rass_index* idx, const float* d_queries, int nq, int k, const int32_t* d_q_filter,
                        const int32_t* d_q_filter_mask, int64_t id_base, float* d_out_scores, int64_t* d_out_ids,
                        const float* d_after_score = nullptr, const int64_t* d_after_row = nullptr) {
    rass_engine* eng = idx->eng;
    const int64_t rows = idx->rows.load(std::memory_order_acquire);
    const bool need_tags = (idx->deleted.load(std::memory_order_acquire) > 0) || (d_q_filter != nullptr);
    const bool gid = idx->has_gid.load(std::memory_order_acquire);  // caller-assigned ids: reported instead of
    if (idx->dtype == RASS_BF16) {
        ScanExt bext;
        bext.d_q_mask = d_q_filter_mask;
        bext.d_after_s = d_after_score;
        bext.d_after_i = d_after_row;
        return bf16_scan_launch(idx, d_queries, nq, d_q_filter, k, (gid || d_after_score) ? 0 : id_base, d_out_scores,
                                d_out_ids, need_tags ? idx->d_tags : nullptr, eng, eng->stream,
                                gid ? idx->d_gid : nullptr, (d_q_filter_mask || d_after_score) ? &bext : nullptr);
    }
    if (idx->prefilter && rows > 0 && k <= kPrefilterMaxK && !d_after_score)
        return prefilter_launch(idx, d_queries, nq, d_q_filter, k, id_base, d_out_scores, d_out_ids,
                                need_tags ? idx->d_tags : nullptr, eng, eng->stream, nullptr, nullptr, d_q_filter_mask,
                                gid ? idx->d_gid : nullptr);
    ScanExt ext;
    ext.d_q_mask = d_q_filter_mask;
    ext.d_after_s = d_after_score;
    ext.d_after_i = d_after_row;
    // the continuation bound names ROWS of this index (the kernel compares id_base + row): the scan runs with
    // id_base 0 and the ids are translated afterwards, as for caller-assigned ids
    return scan_launch(idx->d_rows ? idx->d_rows : reinterpret_cast<const float*>(eng->d_scratch), rows,
                       idx->stride, need_tags ? idx->d_tags : nullptr, d_queries, idx->dim, idx->dim, nq,
                       d_q_filter, k, (gid || d_after_score) ? 0 : id_base, d_out_scores, d_out_ids, eng->d_scratch,
                       eng->scratch_bytes, eng->n_cus, eng->stream, eng, nullptr, gid ? idx->d_gid : nullptr,
                       (d_q_filter_mask || d_after_score) ? &ext : nullptr);
}

struct BatchLayout {
    size_t q_padded, part_scores, part_ids, sample_best, total;
    size_t part_per_group;  // elements of one group's [grid][32][k] lists
};

BatchLayout batch_layout(int groups, int grid, int k, int64_t stride) {
    BatchLayout L;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t off = 0;
    L.q_padded = off;
    off = up(off + (size_t)groups * 32 * stride * sizeof(float));
    L.part_per_group = (size_t)grid * 32 * k;
    L.part_scores = off;
    off = up(off + (size_t)groups * L.part_per_group * sizeof(float));
    L.part_ids = off;
    off = up(off + (size_t)groups * L.part_per_group * sizeof(int64_t));
    L.sample_best = off;
    off = up(off + (size_t)groups * 32 * rass::kMaxSampleGroups * sizeof(float));
    L.total = off;
    return L;
}

static bool scan_batch_one_sample() {
    const char* e = getenv("RASS_SCAN_BATCH_SAMPLE");
    return !(e && e[0] == 'g');
}

// The fused batch of rass_index_search_device_batch on an fp32 flat index: the per-group steps of scan_launch, but
// ONE normalise launch and ONE merge launch for the whole batch, and the groups' sample passes back to back (their
// 64 * grid rows stay in the Infinity Cache between them) ahead of the big scans.  Per 32 queries the serial tail
// of a search (normalise 4.8 us + merge 17 us on 32 of 256 CUs + launch gaps) shrinks to the sample pass.
int scan_launch_batch(rass_index* idx, const float* d_queries, int nq, int k, const int32_t* d_q_filter, int64_t id_base,
                      float* d_out_scores, int64_t* d_out_ids, int64_t out_scores_group_stride,
                      int64_t out_ids_group_stride) {
    rass_engine* eng = idx->eng;
    hipStream_t st = eng->stream;
    const int64_t rows = idx->rows.load(std::memory_order_acquire);
    const bool need_tags = (idx->deleted.load(std::memory_order_acquire) > 0) || (d_q_filter != nullptr);
    const bool gid = idx->has_gid.load(std::memory_order_acquire);
    const int64_t stride = idx->stride;
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    if (rows < 0 || rows > 0x7fffffc0LL) return fail(RASS_ERR_INVALID, "n_rows out of range for one scan");
    if (!rass::scan_supported_stride(stride) || stride > kMaxStride)
        return fail(RASS_ERR_UNSUPPORTED, kStrideMsg);
    const int groups = (nq + RASS_MAX_QBATCH - 1) / RASS_MAX_QBATCH;
    if (stride > kNarrowStride) {
        // wide rows: group by group through scan_launch (16 queries per kernel launch; no fused normalise / merge)
        for (int g = 0; g < groups; ++g) {
            const int b = std::min(RASS_MAX_QBATCH, nq - g * RASS_MAX_QBATCH);
            const int64_t so = out_scores_group_stride > 0 ? out_scores_group_stride : (int64_t)RASS_MAX_QBATCH * k;
            const int64_t io = out_ids_group_stride > 0 ? out_ids_group_stride : (int64_t)RASS_MAX_QBATCH * k;
            int rc = scan_launch(idx->d_rows ? idx->d_rows : reinterpret_cast<const float*>(eng->d_scratch), rows, stride,
                                 need_tags ? idx->d_tags : nullptr, d_queries + (int64_t)g * RASS_MAX_QBATCH * idx->dim, idx->dim,
                                 idx->dim, b, d_q_filter ? d_q_filter + g * RASS_MAX_QBATCH : nullptr, k, gid ? 0 : id_base,
                                 d_out_scores + g * so, d_out_ids + g * io, eng->d_scratch, eng->scratch_bytes, eng->n_cus, st,
                                 eng, nullptr, gid ? idx->d_gid : nullptr, nullptr);
            if (rc != RASS_OK) return rc;
        }
        return RASS_OK;
    }
    const int64_t n_tiles = (rows + 31) / 32;
    int grid = (int)std::min<int64_t>(std::max<int64_t>(n_tiles, 1), std::min(eng->n_cus, kMaxGrid));
    if ((int64_t)grid * k > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / k;

    const BatchLayout L = batch_layout(groups, grid, k, stride);
    if (eng->batch_bytes < L.total) {
        // earlier batches on this stream may still read the old block: hipFree waits for the device
        if (eng->d_batch) HIP_TRY(hipFree(eng->d_batch));
        eng->d_batch = nullptr;
        eng->batch_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_batch), L.total));
        eng->batch_bytes = L.total;
    }
    unsigned char* ws = eng->d_batch;
    float* q_all = reinterpret_cast<float*>(ws + L.q_padded);
    float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
    int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
    float* sample_best = reinterpret_cast<float*>(ws + L.sample_best);

    HIP_TRY(rass::launch_normalize_rows_f32(d_queries, idx->dim, q_all, stride, nq, idx->dim, st, (int64_t)groups * 32));

    const float* corpus = idx->d_rows ? idx->d_rows : reinterpret_cast<const float*>(eng->d_scratch);
    auto group_args = [&](int g) {
        rass::ScanArgs a;
        a.corpus = corpus;
        a.row_tag = need_tags ? idx->d_tags : nullptr;
        a.q_padded = q_all + (int64_t)g * 32 * stride;
        a.q_filter = d_q_filter ? d_q_filter + g * 32 : nullptr;
        a.part_scores = part_scores + (int64_t)g * L.part_per_group;
        a.part_ids = part_ids + (int64_t)g * L.part_per_group;
        a.row_stride = stride;
        a.id_base = gid ? 0 : id_base;
        a.n_rows = (int)rows;
        a.nq = std::min(RASS_MAX_QBATCH, nq - g * 32);
        a.k = k;
        a.xcd_skew = (grid == eng->n_cus && grid % 8 == 0) ? scan_xcd_skew(a.nq) : 0;
        return a;
    };
    const int64_t sample_rows = (int64_t)64 * grid;
    const int64_t min_share = scan_sample_floor_min_share();
    const bool sample = min_share > 0 && grid <= rass::kMaxSampleGroups && rows >= min_share * sample_rows;
    // The sample passes: ONE launch for all groups (kFlatSampleGroups, 32 workgroups of 16 tiles per group: the same
    // 64 * grid sample rows, the floor = the k-th largest of 32 block maxima instead of `grid` of them) when the batch has
    // several full groups; group by group otherwise (RASS_SCAN_BATCH_SAMPLE=groups: the A/B).  Results do not depend on the
    // floor (rows tying with it are kept).
    const bool one_sample = sample && groups >= 2 && nq % 32 == 0 && scan_batch_one_sample() && grid >= 32;
    const int sample_wgs = one_sample ? 32 : grid;
    if (one_sample) {
        rass::ScanArgs s = group_args(0);
        s.n_rows = (int)sample_rows;
        s.xcd_skew = 0;
        s.sample_pass = true;
        s.nq = 32;
        s.part_scores = sample_best;
        s.part_ids = nullptr;
        s.wgs_per_group = sample_wgs;
        s.q_group_stride = 32 * stride;
        s.part_group_stride = (int64_t)32 * rass::kMaxSampleGroups;
        s.nq_total = nq;
        HIP_TRY(rass::launch_scan_topk_f32(s, groups * sample_wgs, st));
    } else if (sample)
        for (int g = 0; g < groups; ++g) {
            rass::ScanArgs s = group_args(g);
            if (s.nq <= 16) continue;
            s.n_rows = (int)sample_rows;
            s.xcd_skew = 0;
            s.sample_pass = true;
            s.part_scores = sample_best + (int64_t)g * 32 * rass::kMaxSampleGroups;
            s.part_ids = nullptr;
            HIP_TRY(rass::launch_scan_topk_f32(s, grid, st));
        }
    for (int g = 0; g < groups; ++g) {
        rass::ScanArgs a = group_args(g);
        if (sample && a.nq > 16) {
            a.sample_best = sample_best + (int64_t)g * 32 * rass::kMaxSampleGroups;
            a.sample_groups = sample_wgs;
        }
        const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        HIP_TRY(rass::launch_scan_topk_f32(a, grid, st));
        if (timed) {
            HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
            eng->ev_used += 1;
        }
    }
    rass::MergeGroups mg;
    mg.size = RASS_MAX_QBATCH;
    mg.nq_total = nq;
    mg.lists_are_dense = true;
    mg.score_stride = mg.id_stride = (int64_t)L.part_per_group;
    mg.out_score_stride = out_scores_group_stride > 0 ? out_scores_group_stride : (int64_t)RASS_MAX_QBATCH * k;
    mg.out_id_stride = out_ids_group_stride > 0 ? out_ids_group_stride : (int64_t)RASS_MAX_QBATCH * k;
    HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, nq, k, d_out_scores, d_out_ids, st,
                                    gid ? idx->d_gid : nullptr, 0, 0, &mg));
    return RASS_OK;
}

// The prefilter mode's batch (rass_index_search_device_batch on an index in mode 1 / 2): ONE normalise, ONE query
// conversion, the groups' candidate scans back to back, ONE grouped merge of their [grid][32][32] lists and ONE re-rank
// launch over all queries.  Group by group the serial tail of a 32-query search (normalise 5 + convert 7 + merge 34 on 32 of
// 256 CUs + re-rank 26 us) was a quarter of the int8 mode's time.  Same results as the group-by-group path, bit for bit.
int prefilter_launch_batch(rass_index* idx, const float* d_queries, int nq, int k, const int32_t* d_q_filter, int64_t id_base,
                           float* d_out_scores, int64_t* d_out_ids, int64_t gs, int64_t gi) {
    rass_engine* eng = idx->eng;
    hipStream_t st = eng->stream;
    const int64_t rows = idx->rows.load(std::memory_order_acquire);
    const bool need_tags = (idx->deleted.load(std::memory_order_acquire) > 0) || (d_q_filter != nullptr);
    const int64_t stride = idx->stride;
    const int kc = RASS_MAX_K;
    const int groups = (nq + RASS_MAX_QBATCH - 1) / RASS_MAX_QBATCH;
    const int64_t n_tiles = (rows + 63) / 64;
    int grid = (int)std::min<int64_t>(std::max<int64_t>(n_tiles, 1), std::min(eng->n_cus, kMaxGrid));
    if ((int64_t)grid * kc > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / kc;
    const BatchLayout L = batch_layout(groups, grid, kc, stride);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t off_qsmall = L.total;                                                   // bf16 / int8 queries
    const size_t off_cs = up(off_qsmall + (size_t)groups * 32 * kMaxStride * 2);        // [groups][32][32] candidate scores
    const size_t off_ci = up(off_cs + (size_t)groups * 32 * kc * sizeof(float));        // ... and rows
    const size_t total = up(off_ci + (size_t)groups * 32 * kc * sizeof(int64_t));
    if (eng->batch_bytes < total) {
        if (eng->d_batch) HIP_TRY(hipFree(eng->d_batch));   // waits for earlier batches on the device
        eng->d_batch = nullptr;
        eng->batch_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_batch), total));
        eng->batch_bytes = total;
    }
    unsigned char* ws = eng->d_batch;
    float* q_all = reinterpret_cast<float*>(ws + L.q_padded);
    float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
    int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
    unsigned char* q_small = ws + off_qsmall;
    float* cand_scores = reinterpret_cast<float*>(ws + off_cs);
    int64_t* cand_rows = reinterpret_cast<int64_t*>(ws + off_ci);
    const int nq_pad = groups * 32;
    HIP_TRY(rass::launch_normalize_rows_f32(d_queries, idx->dim, q_all, stride, nq, idx->dim, st, nq_pad));
    const bool i8 = idx->prefilter == 2;
    const int64_t qs_stride = i8 ? idx->stride_i8 : stride * 2;   // bytes per converted query
    if (i8) HIP_TRY(rass::launch_queries_to_i8(q_all, q_small, nq_pad, stride, idx->stride_i8, st));
    else HIP_TRY(rass::launch_queries_to_bf16(q_all, q_small, (int64_t)nq_pad * stride, st));
    // the int8 scans' sample launches: ONE grouped launch for all groups when every group is full (32 launches of ~10 us each
    // otherwise: 5 % of a 1 024-query step)
    const bool floor_on = i8_sample_floor(rows, grid);
    const bool one_sample = i8 && floor_on && groups >= 2 && nq % 32 == 0;
    if (one_sample) {
        rass::ScanI8Args sa;
        sa.corpus = idx->d_rows_i8;
        sa.row_scale = idx->d_row_scale;
        sa.row_tag = need_tags ? idx->d_tags : nullptr;
        sa.q_i8 = reinterpret_cast<const signed char*>(q_small);
        sa.q_filter = d_q_filter;
        sa.part_scores = reinterpret_cast<float*>(ws + L.sample_best);
        sa.part_ids = nullptr;
        sa.row_stride = idx->stride_i8;
        sa.n_rows = 64 * grid;
        sa.nq = 32;
        sa.k = 1;
        sa.wgs_per_group = grid;
        sa.q_group_stride = (int64_t)32 * qs_stride;
        sa.part_group_stride = (int64_t)32 * rass::kMaxSampleGroups;
        HIP_TRY(rass::launch_scan_i8_topk(sa, groups * grid, st));
    }
    for (int g = 0; g < groups; ++g) {
        const int b = std::min(RASS_MAX_QBATCH, nq - g * 32);
        const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        if (i8) {
            rass::ScanI8Args a;
            a.corpus = idx->d_rows_i8;
            a.row_scale = idx->d_row_scale;
            a.row_tag = need_tags ? idx->d_tags : nullptr;
            a.q_i8 = reinterpret_cast<const signed char*>(q_small + (int64_t)g * 32 * qs_stride);
            a.q_filter = d_q_filter ? d_q_filter + g * 32 : nullptr;
            a.part_scores = part_scores + (int64_t)g * L.part_per_group;
            a.part_ids = part_ids + (int64_t)g * L.part_per_group;
            a.row_stride = idx->stride_i8;
            a.n_rows = (int)rows;
            a.nq = b;
            a.k = kc;
            if (one_sample) {
                a.sample_best = reinterpret_cast<float*>(ws + L.sample_best) + (int64_t)g * 32 * rass::kMaxSampleGroups;
                a.sample_groups = grid;
            } else if (floor_on) {
                rass::ScanI8Args sa = a;
                sa.n_rows = 64 * grid;
                sa.k = 1;
                sa.part_scores = reinterpret_cast<float*>(ws + L.sample_best) + (int64_t)g * 32 * rass::kMaxSampleGroups;
                sa.part_ids = nullptr;
                HIP_TRY(rass::launch_scan_i8_topk(sa, grid, st));
                a.sample_best = sa.part_scores;
                a.sample_groups = grid;
            }
            HIP_TRY(rass::launch_scan_i8_topk(a, grid, st));
        } else {
            rass::ScanBf16Args a;
            a.corpus = idx->d_rows_bf16;
            a.row_tag = need_tags ? idx->d_tags : nullptr;
            a.q_bf16 = reinterpret_cast<const unsigned short*>(q_small + (int64_t)g * 32 * qs_stride);
            a.q_filter = d_q_filter ? d_q_filter + g * 32 : nullptr;
            a.part_scores = part_scores + (int64_t)g * L.part_per_group;
            a.part_ids = part_ids + (int64_t)g * L.part_per_group;
            a.row_stride = stride;
            a.n_rows = (int)rows;
            a.nq = b;
            a.k = kc;
            if (i8_sample_floor(rows, grid)) {
                rass::ScanBf16Args sa = a;
                sa.n_rows = 64 * grid;
                sa.k = 1;
                sa.part_scores = reinterpret_cast<float*>(ws + L.sample_best) + (int64_t)g * 32 * rass::kMaxSampleGroups;
                sa.part_ids = nullptr;
                HIP_TRY(rass::launch_scan_bf16_topk(sa, grid, st));
                a.sample_best = sa.part_scores;
                a.sample_groups = grid;
            }
            HIP_TRY(rass::launch_scan_bf16_topk(a, grid, st));
        }
        if (timed) {
            HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
            eng->ev_used += 1;
        }
    }
    rass::MergeGroups mg;
    mg.size = RASS_MAX_QBATCH;
    mg.nq_total = nq;
    mg.lists_are_dense = true;
    mg.score_stride = mg.id_stride = (int64_t)L.part_per_group;
    mg.out_score_stride = mg.out_id_stride = (int64_t)32 * kc;
    HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, nq, kc, cand_scores, cand_rows, st, nullptr, 0, 0, &mg));
    const bool gid = idx->has_gid.load(std::memory_order_acquire);
    HIP_TRY(rass::launch_rerank_f32(idx->d_rows, stride, q_all, cand_rows, nq, kc, k, gid ? 0 : id_base, d_out_scores, d_out_ids, st, gs,
                                    gi, gid ? idx->d_gid : nullptr));
    return RASS_OK;
}

}  // namespace

int rass_index_search_device_ex(rass_index_t* idx, const float* d_queries, int nq, int k, const int32_t* d_q_filter,
                                const int32_t* d_q_filter_mask, int64_t id_base, float* d_out_scores,
                                int64_t* d_out_ids) {
    if (!idx || !d_queries || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (d_q_filter_mask && !d_q_filter) return fail(RASS_ERR_INVALID, "q_filter_mask without q_filter");
    rass_engine* eng = idx->eng;
    std::lock_guard<std::mutex> lk(eng->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    return search_device_group(idx, d_queries, nq, k, d_q_filter, d_q_filter_mask, id_base, d_out_scores, d_out_ids);
}

int rass_index_search_device_after(rass_index_t* idx, const float* d_queries, int nq, int k, const int32_t* d_q_filter,
                                   const int32_t* d_q_filter_mask, const float* d_after_score,
                                   const int64_t* d_after_row, float* d_out_scores, int64_t* d_out_ids) {
    if (!idx || !d_queries || !d_out_scores || !d_out_ids || !d_after_score || !d_after_row)
        return fail(RASS_ERR_INVALID, "NULL argument");
    if (d_q_filter_mask && !d_q_filter) return fail(RASS_ERR_INVALID, "q_filter_mask without q_filter");
    rass_engine* eng = idx->eng;
    std::lock_guard<std::mutex> lk(eng->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    return search_device_group(idx, d_queries, nq, k, d_q_filter, d_q_filter_mask, 0, d_out_scores, d_out_ids,
                               d_after_score, d_after_row);
}

int rass_index_search_device_batch(rass_index_t* idx, const float* d_queries, int nq, int k, const int32_t* d_q_filter,
                                   int64_t id_base, float* d_out_scores, int64_t* d_out_ids,
                                   int64_t out_scores_group_stride, int64_t out_ids_group_stride) {
    if (!idx || !d_queries || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 1 || nq > RASS_MAX_DEVICE_BATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, RASS_MAX_DEVICE_BATCH]");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    const int64_t gs = out_scores_group_stride > 0 ? out_scores_group_stride : (int64_t)RASS_MAX_QBATCH * k;
    const int64_t gi = out_ids_group_stride > 0 ? out_ids_group_stride : (int64_t)RASS_MAX_QBATCH * k;
    if (gs < (int64_t)RASS_MAX_QBATCH * k || gi < (int64_t)RASS_MAX_QBATCH * k)
        return fail(RASS_ERR_INVALID, "output group strides must be >= 32 * k elements");
    rass_engine* eng = idx->eng;
    std::lock_guard<std::mutex> lk(eng->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    const bool fused = idx->dtype == RASS_F32 && !idx->prefilter && nq > RASS_MAX_QBATCH;
    if (fused) return scan_launch_batch(idx, d_queries, nq, k, d_q_filter, id_base, d_out_scores, d_out_ids, gs, gi);
    if (idx->prefilter && idx->dtype == RASS_F32 && nq > RASS_MAX_QBATCH && k <= kPrefilterMaxK &&
        idx->rows.load(std::memory_order_acquire) > 0)
        return prefilter_launch_batch(idx, d_queries, nq, k, d_q_filter, id_base, d_out_scores, d_out_ids, gs, gi);
    // bf16 / prefilter corpora and single groups: the same result group by group
    for (int g = 0; g * RASS_MAX_QBATCH < nq; ++g) {
        const int b = std::min(RASS_MAX_QBATCH, nq - g * RASS_MAX_QBATCH);
        rc = search_device_group(idx, d_queries + (int64_t)g * RASS_MAX_QBATCH * idx->dim, b, k,
                                 d_q_filter ? d_q_filter + g * RASS_MAX_QBATCH : nullptr, nullptr, id_base,
                                 d_out_scores + g * gs, d_out_ids + g * gi);
        if (rc != RASS_OK) return rc;
    }
    return RASS_OK;
}

int rass_index_search_device(rass_index_t* idx, const float* d_queries, int nq, int k, const int32_t* d_q_filter,
                             int64_t id_base, float* d_out_scores, int64_t* d_out_ids) {
    return rass_index_search_device_ex(idx, d_queries, nq, k, d_q_filter, nullptr, id_base, d_out_scores, d_out_ids);
}

int rass_index_candidates_device(rass_index_t* idx, const float* d_queries, int nq, const int32_t* d_q_filter,
                                 float* d_cand_scores, int64_t* d_cand_rows) {
    if (!idx || !d_queries || !d_cand_scores || !d_cand_rows) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 1 || nq > RASS_MAX_QBATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, RASS_MAX_QBATCH]");
    rass_engine* eng = idx->eng;
    std::lock_guard<std::mutex> lk(eng->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    if (!idx->prefilter || idx->dtype != RASS_F32) return fail(RASS_ERR_UNSUPPORTED, "the index is not in a prefilter mode");
    if (idx->rows.load(std::memory_order_acquire) <= 0) return fail(RASS_ERR_INVALID, "the index is empty");
    const bool need_tags = (idx->deleted.load(std::memory_order_acquire) > 0) || (d_q_filter != nullptr);
    const ScratchLayout L = scratch_layout(RASS_MAX_QBATCH, RASS_MAX_K);
    // the re-rank's own output (top-1 of every query) goes to the scratch's candidate area: not reported here
    return prefilter_launch(idx, d_queries, nq, d_q_filter, 1, 0, reinterpret_cast<float*>(eng->d_scratch + L.cand_scores),
                            reinterpret_cast<int64_t*>(eng->d_scratch + L.cand_ids), need_tags ? idx->d_tags : nullptr, eng,
                            eng->stream, d_cand_scores, d_cand_rows);
}

namespace {

// A pinned host slot for one search call in flight (blocks while all kHostSlots are taken).
HostSlot* slot_acquire(rass_engine* eng) {
    std::unique_lock<std::mutex> lk(eng->slot_mu);
    for (;;) {
        for (HostSlot& sl : eng->slots)
            if (!sl.busy) {
                sl.busy = true;
                return &sl;
            }
        eng->slot_cv.wait(lk);
    }
}

void slot_release(rass_engine* eng, HostSlot* sl) {
    {
        std::lock_guard<std::mutex> lk(eng->slot_mu);
        sl->busy = false;
    }
    eng->slot_cv.notify_one();
}

struct SlotGuard {
    rass_engine* eng;
    HostSlot* sl;
    SlotGuard(rass_engine* e) : eng(e), sl(slot_acquire(e)) {}
    ~SlotGuard() { slot_release(eng, sl); }
};

}  // namespace

int rass_index_search_ex(rass_index_t* idx, const float* queries, int nq, int k, const int32_t* q_filter,
                         const int32_t* q_filter_mask, float* out_scores, int64_t* out_ids) {
    if (!idx || !out_scores || !out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 0 || (nq > 0 && !queries)) return fail(RASS_ERR_INVALID, "bad queries / nq");
    if (k < 1 || k > RASS_MAX_K_MULTIPASS) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K_MULTIPASS]");
    if (q_filter_mask && !q_filter) return fail(RASS_ERR_INVALID, "q_filter_mask without q_filter");
    rass_engine* eng = idx->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    const int dim = idx->dim;
    SlotGuard guard(eng);
    HostSlot* sl = guard.sl;
    for (int done = 0; done < nq;) {
        const int b = std::min(RASS_MAX_QBATCH, nq - done);
        memcpy(sl->h_q, queries + (int64_t)done * dim, (size_t)b * dim * sizeof(float));
        if (q_filter) memcpy(sl->h_filter, q_filter + done, (size_t)b * sizeof(int32_t));
        if (q_filter_mask) memcpy(sl->h_mask, q_filter_mask + done, (size_t)b * sizeof(int32_t));
        // k > RASS_MAX_K: passes of <= 32; pass p ranks only the rows strictly AFTER pass p-1's last hit
        for (int kdone = 0; kdone < k;) {
            const int kk = std::min(RASS_MAX_K, k - kdone);
            const bool cont = kdone > 0;
            {
                // the engine lock is held while ENQUEUING only: device staging and scratch are shared by
                // stream order, the wait happens on this call's own event
                std::lock_guard<std::mutex> lk(eng->mu);
                hipStream_t st = eng->stream;
                HIP_TRY(hipMemcpyAsync(eng->d_qraw, sl->h_q, (size_t)b * dim * sizeof(float), hipMemcpyHostToDevice, st));
                const int32_t* d_filter = nullptr;
                ScanExt ext;
                if (q_filter) {
                    HIP_TRY(hipMemcpyAsync(eng->d_qfilter, sl->h_filter, (size_t)b * sizeof(int32_t),
                                           hipMemcpyHostToDevice, st));
                    d_filter = eng->d_qfilter;
                }
                if (q_filter_mask) {
                    HIP_TRY(hipMemcpyAsync(eng->d_qmask, sl->h_mask, (size_t)b * sizeof(int32_t), hipMemcpyHostToDevice, st));
                    ext.d_q_mask = eng->d_qmask;
                }
                if (cont) {
                    HIP_TRY(hipMemcpyAsync(eng->d_after_s, sl->h_after_s, (size_t)b * sizeof(float), hipMemcpyHostToDevice, st));
                    HIP_TRY(hipMemcpyAsync(eng->d_after_i, sl->h_after_i, (size_t)b * sizeof(int64_t), hipMemcpyHostToDevice, st));
                    ext.d_after_s = eng->d_after_s;
                    ext.d_after_i = eng->d_after_i;
                }
                const bool use_ext = q_filter_mask || cont;
                const int64_t rows = idx->rows.load(std::memory_order_acquire);
                const bool need_tags = (idx->deleted.load(std::memory_order_acquire) > 0) || (d_filter != nullptr);
                const bool gid = idx->has_gid.load(std::memory_order_acquire);
                if (gid && cont)  // the continuation bound compares row ordinals, the caller would hand back global ids
                    return fail(RASS_ERR_UNSUPPORTED, "k > RASS_MAX_K on an index with caller-assigned row ids");
                if (idx->dtype == RASS_BF16)
                    rc = bf16_scan_launch(idx, eng->d_qraw, b, d_filter, kk, 0, eng->d_out_scores, eng->d_out_ids,
                                          need_tags ? idx->d_tags : nullptr, eng, st, gid ? idx->d_gid : nullptr,
                                          use_ext ? &ext : nullptr);
                else if (idx->prefilter && rows > 0 && k <= kPrefilterMaxK && !cont)
                    rc = prefilter_launch(idx, eng->d_qraw, b, d_filter, kk, 0, eng->d_out_scores, eng->d_out_ids,
                                          need_tags ? idx->d_tags : nullptr, eng, st, nullptr, nullptr, ext.d_q_mask,
                                          gid ? idx->d_gid : nullptr);
                else
                    rc = scan_launch(idx->d_rows ? idx->d_rows : reinterpret_cast<const float*>(eng->d_scratch), rows,
                                     idx->stride, need_tags ? idx->d_tags : nullptr, eng->d_qraw, dim, dim, b, d_filter,
                                     kk, 0, eng->d_out_scores, eng->d_out_ids, eng->d_scratch, eng->scratch_bytes,
                                     eng->n_cus, st, eng, nullptr, gid ? idx->d_gid : nullptr, use_ext ? &ext : nullptr);
                if (rc != RASS_OK) return rc;
                HIP_TRY(hipMemcpyAsync(sl->h_out_s, eng->d_out_scores, (size_t)b * kk * sizeof(float),
                                       hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(sl->h_out_i, eng->d_out_ids, (size_t)b * kk * sizeof(int64_t),
                                       hipMemcpyDeviceToHost, st));
                HIP_TRY(hipEventRecord(sl->done, st));
            }
            HIP_TRY(hipEventSynchronize(sl->done));
            for (int q = 0; q < b; ++q) {
                memcpy(out_scores + (int64_t)(done + q) * k + kdone, sl->h_out_s + (int64_t)q * kk, (size_t)kk * sizeof(float));
                memcpy(out_ids + (int64_t)(done + q) * k + kdone, sl->h_out_i + (int64_t)q * kk, (size_t)kk * sizeof(int64_t));
                // continuation bound for the next pass: this pass's last hit, or "nothing left" (-inf) when the
                // pass came back short
                const int64_t last_id = sl->h_out_i[(int64_t)q * kk + kk - 1];
                sl->h_after_s[q] = last_id >= 0 ? sl->h_out_s[(int64_t)q * kk + kk - 1] : -INFINITY;
                sl->h_after_i[q] = last_id >= 0 ? last_id : INT64_MAX;
            }
            kdone += kk;
        }
        done += b;
    }
    return RASS_OK;
}

int rass_index_search_multi(rass_index_t* const* idxs, const float* queries, int nq, int k, const int32_t* q_filter,
                            const int32_t* q_filter_mask, float* out_scores, int64_t* out_ids) {
    if (!idxs || !out_scores || !out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 0 || (nq > 0 && !queries)) return fail(RASS_ERR_INVALID, "bad queries / nq");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    if (q_filter_mask && !q_filter) return fail(RASS_ERR_INVALID, "q_filter_mask without q_filter");
    if (nq == 0) return RASS_OK;
    rass_engine* eng = idxs[0] ? idxs[0]->eng : nullptr;
    for (int q = 0; q < nq; ++q) {
        if (!idxs[q]) return fail(RASS_ERR_INVALID, "NULL index");
        if (idxs[q]->eng != eng) return fail(RASS_ERR_INVALID, "the indices of one batch must share an engine (one GPU)");
        if (idxs[q]->dtype != RASS_F32) return fail(RASS_ERR_UNSUPPORTED, "cross-index batches are fp32-only");
        if (idxs[q]->has_gid.load()) return fail(RASS_ERR_UNSUPPORTED, "cross-index batches need plain row ids");
    }
    if (eng && eng->dim > kNarrowStride)
        return fail(RASS_ERR_UNSUPPORTED, "cross-index batches need dim <= 1024: search wide-row indices one by one");
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    const int dim = eng->dim;
    SlotGuard guard(eng);
    HostSlot* sl = guard.sl;
    const size_t item_bytes = 4 + 4 + 4 + 8 + 8;
    if (!sl->h_items) HIP_TRY(hipHostMalloc(&sl->h_items, (size_t)kMultiMaxItems * item_bytes, hipHostMallocDefault));
    int32_t* h_tile = static_cast<int32_t*>(sl->h_items);
    int32_t* h_rows = h_tile + kMultiMaxItems;
    uint32_t* h_mask = reinterpret_cast<uint32_t*>(h_rows + kMultiMaxItems);
    const float** h_base = reinterpret_cast<const float**>(h_mask + kMultiMaxItems);
    const int32_t** h_tags = reinterpret_cast<const int32_t**>(h_base + kMultiMaxItems);
    for (int done = 0; done < nq;) {
        const int b = std::min(RASS_MAX_QBATCH, nq - done);
        memcpy(sl->h_q, queries + (int64_t)done * dim, (size_t)b * dim * sizeof(float));
        if (q_filter) memcpy(sl->h_filter, q_filter + done, (size_t)b * sizeof(int32_t));
        if (q_filter_mask) memcpy(sl->h_mask, q_filter_mask + done, (size_t)b * sizeof(int32_t));
        {
            std::lock_guard<std::mutex> lk(eng->mu);  // slab pointers and row counts are stable under it
            hipStream_t st = eng->stream;
            if (!eng->d_mw_tile) {
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_mw_tile), (size_t)kMultiMaxItems * 4));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_mw_rows), (size_t)kMultiMaxItems * 4));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_mw_mask), (size_t)kMultiMaxItems * 4));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_mw_base), (size_t)kMultiMaxItems * 8));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_mw_tags), (size_t)kMultiMaxItems * 8));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_mw_n), 4));
            }
            // the work list: for every DISTINCT index of the batch its tiles, each with the mask of the batch's
            // queries that belong to that index (a tile is fetched once however many of them there are)
            int n_items = 0;
            for (int q = 0; q < b; ++q) {
                rass_index* idx = idxs[done + q];
                bool seen = false;
                for (int p = 0; p < q && !seen; ++p) seen = idxs[done + p] == idx;
                if (seen) continue;
                uint32_t mask = 0;
                for (int p = q; p < b; ++p)
                    if (idxs[done + p] == idx) mask |= 1u << p;
                const int64_t rows = idx->rows.load(std::memory_order_acquire);
                const bool need_tags = idx->deleted.load(std::memory_order_acquire) > 0 || q_filter != nullptr;
                const int64_t tiles = (rows + 31) / 32;
                if (n_items + tiles > kMultiMaxItems)
                    return fail(RASS_ERR_UNSUPPORTED, "cross-index batch exceeds 65536 tiles (2 M rows): search the large index on its own");
                for (int64_t t = 0; t < tiles; ++t) {
                    h_tile[n_items] = (int32_t)t;
                    h_rows[n_items] = (int32_t)std::min<int64_t>(32, rows - 32 * t);
                    h_mask[n_items] = mask;
                    h_base[n_items] = idx->d_rows;
                    h_tags[n_items] = need_tags ? idx->d_tags : nullptr;
                    ++n_items;
                }
            }
            sl->h_scanned[0] = n_items;  // reused as the pinned source of the item count
            if (n_items > 0) {
                HIP_TRY(hipMemcpyAsync(eng->d_mw_tile, h_tile, (size_t)n_items * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(eng->d_mw_rows, h_rows, (size_t)n_items * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(eng->d_mw_mask, h_mask, (size_t)n_items * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(eng->d_mw_base, h_base, (size_t)n_items * 8, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(eng->d_mw_tags, h_tags, (size_t)n_items * 8, hipMemcpyHostToDevice, st));
            }
            HIP_TRY(hipMemcpyAsync(eng->d_mw_n, sl->h_scanned, 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(eng->d_qraw, sl->h_q, (size_t)b * dim * sizeof(float), hipMemcpyHostToDevice, st));
            const int32_t* d_filter = nullptr;
            if (q_filter) {
                HIP_TRY(hipMemcpyAsync(eng->d_qfilter, sl->h_filter, (size_t)b * 4, hipMemcpyHostToDevice, st));
                d_filter = eng->d_qfilter;
            }
            if (q_filter_mask) HIP_TRY(hipMemcpyAsync(eng->d_qmask, sl->h_mask, (size_t)b * 4, hipMemcpyHostToDevice, st));
            // launch: normalise -> MULTI scan over the work list -> merge (ids are rows of each query's own index)
            const int64_t stride = pad_stride(dim);
            const ScratchLayout L = scratch_layout(b, k);
            unsigned char* ws = eng->d_scratch;
            float* q_padded = reinterpret_cast<float*>(ws + L.q_padded);
            float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
            int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
            const int nq_pad = b <= 16 ? 16 : 32;
            HIP_TRY(rass::launch_normalize_rows_f32(eng->d_qraw, dim, q_padded, stride, b, dim, st, nq_pad));
            int grid = std::min(std::max(n_items, 1), std::min(eng->n_cus, kMaxGrid));
            if ((int64_t)grid * k > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / k;
            rass::ScanArgs a;
            a.corpus = reinterpret_cast<const float*>(eng->d_scratch);
            a.row_tag = nullptr;
            a.q_padded = q_padded;
            a.q_filter = d_filter;
            a.q_filter_mask = q_filter_mask ? eng->d_qmask : nullptr;
            a.part_scores = part_scores;
            a.part_ids = part_ids;
            a.row_stride = stride;
            a.id_base = 0;
            a.n_rows = 0;
            a.nq = b;
            a.k = k;
            a.work_tile = eng->d_mw_tile;
            a.work_rows = eng->d_mw_rows;
            a.work_mask = eng->d_mw_mask;
            a.n_work = eng->d_mw_n;
            a.work_base = eng->d_mw_base;
            a.work_tags = eng->d_mw_tags;
            // counted by rass_engine_kernel_timing_* like every other scan launch
            const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
            if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
            HIP_TRY(rass::launch_scan_topk_f32(a, grid, st));
            if (timed) {
                HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
                eng->ev_used += 1;
            }
            HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, b, k, eng->d_out_scores, eng->d_out_ids, st));
            HIP_TRY(hipMemcpyAsync(sl->h_out_s, eng->d_out_scores, (size_t)b * k * sizeof(float), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sl->h_out_i, eng->d_out_ids, (size_t)b * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(sl->done, st));
        }
        HIP_TRY(hipEventSynchronize(sl->done));
        memcpy(out_scores + (int64_t)done * k, sl->h_out_s, (size_t)b * k * sizeof(float));
        memcpy(out_ids + (int64_t)done * k, sl->h_out_i, (size_t)b * k * sizeof(int64_t));
        done += b;
    }
    return RASS_OK;
}

int rass_index_search(rass_index_t* idx, const float* queries, int nq, int k, const int32_t* q_filter,
                      float* out_scores, int64_t* out_ids) {
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    return rass_index_search_ex(idx, queries, nq, k, q_filter, nullptr, out_scores, out_ids);
}

// ---- persistence: header + unpadded fp32 rows + tags
struct SaveHeader {
    char magic[8];
    int32_t version;
    int32_t dim;
    int32_t dtype;
    int32_t reserved;
    int64_t rows;
    int64_t deleted;
};

int rass_index_save(rass_index_t* idx, const char* path) {
    if (!idx || !path) return fail(RASS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = set_device(idx->eng);
    if (rc != RASS_OK) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RASS_ERR_IO, std::string("cannot open for write: ") + path);
    SaveHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "RASSIDX1", 8);
    h.version = 1;
    h.dim = idx->dim;
    h.dtype = (int32_t)idx->dtype;
    h.rows = idx->rows;
    h.deleted = idx->deleted;
    const bool save_gid = idx->has_gid.load();
    h.reserved = save_gid ? 1 : 0;  // 1: rows x int64 caller-assigned ids follow the tags
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    hipStream_t st = idx->eng->stream;
    std::vector<float> buf((size_t)kStageRows * idx->dim);
    for (int64_t r = 0; ok && r < idx->rows; r += kStageRows) {
        const int64_t m = std::min<int64_t>(kStageRows, idx->rows - r);
        std::lock_guard<std::mutex> elk(idx->eng->mu);  // d_stage is shared engine scratch
        hipError_t e = idx->dtype == RASS_BF16
                           ? rass::launch_unpack_rows_tile16b(idx->d_rows_bf16, idx->stride, r, m, idx->dim,
                                                              idx->eng->d_stage, idx->dim, st)
                           : rass::launch_unpack_rows_tile16(idx->d_rows, idx->stride, r, m, idx->dim, idx->eng->d_stage,
                                                             idx->dim, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(buf.data(), idx->eng->d_stage, (size_t)m * idx->dim * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            fclose(f);
            return fail(RASS_ERR_HIP, std::string("save: device read failed: ") + hipGetErrorString(e));
        }
        ok = fwrite(buf.data(), sizeof(float), (size_t)m * idx->dim, f) == (size_t)m * idx->dim;
    }
    if (ok && idx->rows > 0) {
        std::vector<int32_t> tags((size_t)idx->rows);
        hipError_t e = hipMemcpyAsync(tags.data(), idx->d_tags, (size_t)idx->rows * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            fclose(f);
            return fail(RASS_ERR_HIP, std::string("save: tag read failed: ") + hipGetErrorString(e));
        }
        ok = fwrite(tags.data(), 4, (size_t)idx->rows, f) == (size_t)idx->rows;
    }
    if (ok && idx->rows > 0 && save_gid) {
        std::vector<int64_t> gids((size_t)idx->rows);
        hipError_t e = hipMemcpyAsync(gids.data(), idx->d_gid, (size_t)idx->rows * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            fclose(f);
            return fail(RASS_ERR_HIP, std::string("save: id read failed: ") + hipGetErrorString(e));
        }
        ok = fwrite(gids.data(), 8, (size_t)idx->rows, f) == (size_t)idx->rows;
    }
    // durable before the caller renames it into place (docstore.py's manifest scheme)
    ok = ok && fflush(f) == 0 && fsync(fileno(f)) == 0;
    ok = (fclose(f) == 0) && ok;
    return ok ? RASS_OK : fail(RASS_ERR_IO, std::string("short write: ") + path);
}

int rass_index_load(rass_engine_t* eng, const char* name, const char* path, rass_index_t** out) {
    if (!eng || !name || !path || !out) return fail(RASS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RASS_ERR_IO, std::string("cannot open for read: ") + path);
    SaveHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "RASSIDX1", 8) != 0 || h.version != 1) {
        fclose(f);
        return fail(RASS_ERR_IO, "not a rass index file");
    }
    if (h.dim != eng->dim || (h.dtype != RASS_F32 && h.dtype != RASS_BF16) || h.rows < 0) {
        fclose(f);
        return fail(RASS_ERR_INVALID, "index file does not match the engine (dim / dtype)");
    }
    {   // the header's row count must agree with the file length before anything is allocated from it
        const long body = ftell(f);
        int64_t file_len = -1;
        if (body >= 0 && fseek(f, 0, SEEK_END) == 0) file_len = (int64_t)ftell(f);
        const int64_t need = (int64_t)sizeof(SaveHeader) + h.rows * ((int64_t)h.dim * 4 + 4 + (h.reserved == 1 ? 8 : 0));
        if (body < 0 || h.rows > ((int64_t)1 << 40) || file_len < need || fseek(f, body, SEEK_SET) != 0) {
            fclose(f);
            return fail(RASS_ERR_IO, "truncated index file (shorter than its header says)");
        }
    }
    {
        std::lock_guard<std::mutex> lk(eng->mu);
        if (eng->indices.count(name)) {
            fclose(f);
            return fail(RASS_ERR_INVALID, "an index of that name is already open");
        }
    }
    rass_index_t* idx = nullptr;
    int rc = rass_index_open(eng, name, (rass_dtype)h.dtype, h.rows, &idx);  // a bf16 corpus was saved as its exact
                                                                              // fp32 upcast: re-rounding is lossless
    if (rc != RASS_OK) {
        fclose(f);
        return rc;
    }
    std::vector<float> buf((size_t)kStageRows * h.dim);
    for (int64_t r = 0; r < h.rows; r += kStageRows) {
        const int64_t m = std::min<int64_t>(kStageRows, h.rows - r);
        if (fread(buf.data(), sizeof(float), (size_t)m * h.dim, f) != (size_t)m * h.dim) {
            fclose(f);
            (void)rass_index_drop(eng, name);
            return fail(RASS_ERR_IO, "truncated index file (rows)");
        }
        rc = rass_index_add(idx, buf.data(), nullptr, m, /*normalize*/ 0, nullptr);
        if (rc != RASS_OK) {
            fclose(f);
            (void)rass_index_drop(eng, name);
            return rc;
        }
    }
    if (h.rows > 0) {
        std::vector<int32_t> tags((size_t)h.rows);
        if (fread(tags.data(), 4, (size_t)h.rows, f) != (size_t)h.rows) {
            fclose(f);
            (void)rass_index_drop(eng, name);
            return fail(RASS_ERR_IO, "truncated index file (tags)");
        }
        hipError_t e;
        {
            std::lock_guard<std::mutex> lk(idx->mu);
            e = hipMemcpyAsync(idx->d_tags, tags.data(), (size_t)h.rows * 4, hipMemcpyHostToDevice, eng->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(eng->stream);
        }
        if (e != hipSuccess) {
            fclose(f);
            (void)rass_index_drop(eng, name);
            return fail(RASS_ERR_HIP, std::string("load: tag upload failed: ") + hipGetErrorString(e));
        }
        std::lock_guard<std::mutex> lk(idx->mu);
        for (int64_t r = 0; r < h.rows; ++r) {
            if (tags[(size_t)r] == RASS_ROW_TAG_DELETED) {
                idx->host_deleted[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
                idx->deleted += 1;
            } else if (tags[(size_t)r] != 0) {
                idx->has_tags = true;
            }
        }
    }
    if (h.rows > 0 && h.reserved == 1) {
        std::vector<int64_t> gids((size_t)h.rows);
        hipError_t e = hipSuccess;
        if (fread(gids.data(), 8, (size_t)h.rows, f) != (size_t)h.rows) {
            fclose(f);
            (void)rass_index_drop(eng, name);
            return fail(RASS_ERR_IO, "truncated index file (ids)");
        }
        {
            std::lock_guard<std::mutex> lk(idx->mu);
            e = hipMemcpyAsync(idx->d_gid, gids.data(), (size_t)h.rows * 8, hipMemcpyHostToDevice, eng->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(eng->stream);
            idx->has_gid = true;
        }
        if (e != hipSuccess) {
            fclose(f);
            (void)rass_index_drop(eng, name);
            return fail(RASS_ERR_HIP, std::string("load: id upload failed: ") + hipGetErrorString(e));
        }
    }
    fclose(f);
    *out = idx;
    return RASS_OK;
}

int rass_index_fill_synthetic(rass_index_t* idx, int64_t n, uint64_t seed, int64_t row_id_base) {
    if (!idx) return fail(RASS_ERR_INVALID, "index is NULL");
    if (n < 0) return fail(RASS_ERR_INVALID, "n < 0");
    rass_engine* eng = idx->eng;
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    if (n == 0) return RASS_OK;
    {
        std::lock_guard<std::mutex> elk(eng->mu);
        rc = index_reserve(idx, idx->rows + n);
        if (rc != RASS_OK) return rc;
    }
    hipStream_t st = eng->stream;
    if (idx->dtype == RASS_BF16) {
        std::lock_guard<std::mutex> elk(eng->mu);
        if (!eng->d_stage_t16)
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&eng->d_stage_t16), (size_t)(kStageRows + 32) * kMaxStride * sizeof(float)));
        for (int64_t done = 0; done < n; done += kStageRows) {
            const int64_t m = std::min<int64_t>(kStageRows, n - done);
            const int64_t at = idx->rows + done, phase = at & 15;
            // the generator keys a row by row_id_base + its slab row: shift the base so staging row `phase` is row `at`
            HIP_TRY(rass::launch_fill_synthetic_f32(eng->d_stage_t16, idx->stride, phase, m, idx->dim, seed,
                                                    row_id_base + at - phase, st));
            HIP_TRY(rass::launch_convert_tile16_bf16(eng->d_stage_t16, idx->d_rows_bf16, idx->stride, at >> 4,
                                                     (at + m + 15) >> 4, st, 0, at, at + m));
        }
        HIP_TRY(hipStreamSynchronize(st));
    } else
    HIP_TRY(rass::launch_fill_synthetic_f32(idx->d_rows, idx->stride, idx->rows, n, idx->dim, seed, row_id_base, st));
    HIP_TRY(rass::launch_fill_i32(idx->d_tags + idx->rows, n, 0, st));
    HIP_TRY(rass::launch_iota_i64(idx->d_gid + idx->rows, n, idx->rows.load(), st));
    if (idx->prefilter == 1 && idx->dtype == RASS_F32)
        HIP_TRY(rass::launch_convert_tile16_bf16(idx->d_rows, idx->d_rows_bf16, idx->stride, idx->rows >> 4,
                                                 (idx->rows + n + 15) >> 4, st));
    if (idx->prefilter == 2)   // whole blocks: the earlier rows of a partially filled block quantise to the same bytes again
        HIP_TRY(rass::launch_quantize_tile16_i8(idx->d_rows, idx->d_rows_i8, idx->d_row_scale, idx->stride, idx->stride_i8,
                                                idx->rows >> 4, (idx->rows + n + 15) >> 4, st));
    idx->rows += n;
    idx->host_deleted.resize((size_t)((idx->rows + 7) / 8), 0);
    return RASS_OK;
}

size_t rass_scan_workspace_bytes(int nq, int k) {
    if (nq < 1 || nq > RASS_MAX_QBATCH || k < 1 || k > RASS_MAX_K) return 0;
    return scratch_layout(nq, k).total;
}

int rass_scan_topk_f32(const float* d_corpus, int64_t n_rows, int dim, int64_t row_stride,
                       const int32_t* d_row_tag, const float* d_queries, int nq, const int32_t* d_q_filter, int k,
                       int64_t id_base, float* d_out_scores, int64_t* d_out_ids, void* d_workspace,
                       size_t workspace_bytes, void* stream) {
    if (!d_queries || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (n_rows > 0 && !d_corpus) return fail(RASS_ERR_INVALID, "corpus is NULL");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const float* corpus = d_corpus ? d_corpus : reinterpret_cast<const float*>(d_workspace);
    return scan_launch(corpus, n_rows, row_stride, d_row_tag, d_queries, dim, dim, nq, d_q_filter, k, id_base,
                       d_out_scores, d_out_ids, reinterpret_cast<unsigned char*>(d_workspace), workspace_bytes,
                       device_cus(dev), reinterpret_cast<hipStream_t>(stream));
}

int rass_pack_rows_f32(const float* d_in, int64_t in_stride, float* d_packed, int64_t row_stride, int64_t first_row,
                       int64_t n, int dim, int normalize, void* stream) {
    if (n < 0 || dim < 1 || in_stride < dim || row_stride < dim || row_stride % 128 != 0 || first_row < 0)
        return fail(RASS_ERR_INVALID, "bad shape");
    if (n > 0 && (!d_in || !d_packed)) return fail(RASS_ERR_INVALID, "NULL argument");
    HIP_TRY(rass::launch_pack_rows_tile16(d_in, in_stride, d_packed, row_stride, first_row, n, dim, normalize ? 1 : 0,
                                          reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

int rass_unpack_rows_f32(const float* d_packed, int64_t row_stride, int64_t first_row, int64_t n, int dim,
                         float* d_out, int64_t out_stride, void* stream) {
    if (n < 0 || dim < 1 || out_stride < dim || row_stride < dim || row_stride % 128 != 0 || first_row < 0)
        return fail(RASS_ERR_INVALID, "bad shape");
    if (n > 0 && (!d_out || !d_packed)) return fail(RASS_ERR_INVALID, "NULL argument");
    HIP_TRY(rass::launch_unpack_rows_tile16(d_packed, row_stride, first_row, n, dim, d_out, out_stride,
                                            reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

int rass_gather_rows_f32(const float* d_packed, int64_t row_stride, int64_t n_rows, const int64_t* d_row_ids, int64_t n,
                         int dim, float* d_out, int64_t out_stride, void* stream) {
    if (n < 0 || n_rows < 0 || dim < 1 || out_stride < dim || row_stride < dim || row_stride % 128 != 0)
        return fail(RASS_ERR_INVALID, "bad shape");
    if (n > 0 && (!d_out || !d_packed || !d_row_ids)) return fail(RASS_ERR_INVALID, "NULL argument");
    HIP_TRY(rass::launch_gather_rows_tile16(d_packed, row_stride, d_row_ids, n, n_rows, dim, d_out, out_stride,
                                            reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

int rass_topk_merge(const float* d_scores, const int64_t* d_ids, int n_lists, int nq, int k, float* d_out_scores,
                    int64_t* d_out_ids, void* stream) {
    if (!d_scores || !d_ids || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (n_lists < 1 || nq < 1 || k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "bad n_lists / nq / k");
    if ((int64_t)n_lists * k > rass::kMergeMaxCandidates)
        return fail(RASS_ERR_UNSUPPORTED, "n_lists * k exceeds 8192 candidates");
    HIP_TRY(rass::launch_merge_topk(d_scores, d_ids, n_lists, nq, k, d_out_scores, d_out_ids,
                                    reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

int rass_topk_merge_strided(const float* d_scores, const int64_t* d_ids, int64_t score_list_stride,
                            int64_t id_list_stride, int n_lists, int nq, int k, float* d_out_scores,
                            int64_t* d_out_ids, void* stream) {
    if (!d_scores || !d_ids || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (n_lists < 1 || nq < 1 || k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "bad n_lists / nq / k");
    if (score_list_stride < (int64_t)nq * k || id_list_stride < (int64_t)nq * k)
        return fail(RASS_ERR_INVALID, "list strides must be >= nq * k elements");
    if ((int64_t)n_lists * k > rass::kMergeMaxCandidates)
        return fail(RASS_ERR_UNSUPPORTED, "n_lists * k exceeds 8192 candidates");
    HIP_TRY(rass::launch_merge_topk(d_scores, d_ids, n_lists, nq, k, d_out_scores, d_out_ids,
                                    reinterpret_cast<hipStream_t>(stream), nullptr, score_list_stride,
                                    id_list_stride));
    return RASS_OK;
}

int rass_topk_merge_strided_batch(const float* d_scores, const int64_t* d_ids, int64_t score_list_stride,
                                  int64_t id_list_stride, int n_lists, int nq_total, int group_size,
                                  int64_t score_group_stride, int64_t id_group_stride, int k, float* d_out_scores,
                                  int64_t* d_out_ids, void* stream) {
    if (!d_scores || !d_ids || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (n_lists < 1 || nq_total < 1 || k < 1 || k > RASS_MAX_K || group_size < 1 || group_size > RASS_MAX_QBATCH)
        return fail(RASS_ERR_INVALID, "bad n_lists / nq_total / group_size / k");
    if (score_group_stride < (int64_t)group_size * k || id_group_stride < (int64_t)group_size * k ||
        score_list_stride < (int64_t)group_size * k || id_list_stride < (int64_t)group_size * k)
        return fail(RASS_ERR_INVALID, "list and group strides must be >= group_size * k elements");
    if ((int64_t)n_lists * k > rass::kMergeMaxCandidates)
        return fail(RASS_ERR_UNSUPPORTED, "n_lists * k exceeds 8192 candidates");
    rass::MergeGroups mg;
    mg.size = group_size;
    mg.nq_total = nq_total;
    mg.lists_are_dense = false;
    mg.score_stride = score_group_stride;
    mg.id_stride = id_group_stride;
    mg.out_score_stride = mg.out_id_stride = (int64_t)group_size * k;
    HIP_TRY(rass::launch_merge_topk(d_scores, d_ids, n_lists, nq_total, k, d_out_scores, d_out_ids,
                                    reinterpret_cast<hipStream_t>(stream), nullptr, score_list_stride, id_list_stride,
                                    &mg));
    return RASS_OK;
}

int rass_normalize_rows_f32(const float* d_in, int64_t in_stride, float* d_out, int64_t out_stride, int64_t n,
                            int dim, void* stream) {
    if (n < 0 || dim < 1 || in_stride < dim || out_stride < dim) return fail(RASS_ERR_INVALID, "bad shape");
    if (n > 0 && (!d_in || !d_out)) return fail(RASS_ERR_INVALID, "NULL argument");
    HIP_TRY(rass::launch_normalize_rows_f32(d_in, in_stride, d_out, out_stride, n, dim,
                                            reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

struct rass_timer {
    hipEvent_t start = nullptr, stop = nullptr;
};

int rass_timer_create(rass_timer_t** out) {
    if (!out) return fail(RASS_ERR_INVALID, "out is NULL");
    rass_timer* t = new (std::nothrow) rass_timer();
    if (!t) return fail(RASS_ERR_OOM, "host allocation failed");
    hipError_t e = hipEventCreate(&t->start);
    if (e == hipSuccess) e = hipEventCreate(&t->stop);
    if (e != hipSuccess) {
        rass_timer_destroy(t);
        return fail(RASS_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
    }
    *out = t;
    return RASS_OK;
}

void rass_timer_destroy(rass_timer_t* t) {
    if (!t) return;
    if (t->start) (void)hipEventDestroy(t->start);
    if (t->stop) (void)hipEventDestroy(t->stop);
    delete t;
}

int rass_timer_start(rass_timer_t* t, void* stream) {
    if (!t) return fail(RASS_ERR_INVALID, "timer is NULL");
    HIP_TRY(hipEventRecord(t->start, reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

int rass_timer_stop(rass_timer_t* t, void* stream) {
    if (!t) return fail(RASS_ERR_INVALID, "timer is NULL");
    HIP_TRY(hipEventRecord(t->stop, reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

int rass_timer_elapsed_ms(rass_timer_t* t, float* ms) {
    if (!t || !ms) return fail(RASS_ERR_INVALID, "NULL argument");
    HIP_TRY(hipEventSynchronize(t->stop));
    HIP_TRY(hipEventElapsedTime(ms, t->start, t->stop));
    return RASS_OK;
}

int rass_engine_kernel_timing_begin(rass_engine_t* eng, int max_launches) {
    if (!eng || max_launches < 1) return fail(RASS_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(eng->mu);
    HIP_TRY(hipSetDevice(eng->device));
    while (eng->ev_pool.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        eng->ev_pool.push_back(e);
    }
    eng->ev_used = 0;
    eng->ev_on = true;
    return RASS_OK;
}

int rass_engine_kernel_timing_end(rass_engine_t* eng, double* total_ms, int* launches) {
    if (!eng || !total_ms || !launches) return fail(RASS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(eng->mu);
    HIP_TRY(hipSetDevice(eng->device));
    eng->ev_on = false;
    double sum = 0.0;
    for (int i = 0; i < eng->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(eng->ev_pool[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, eng->ev_pool[2 * i], eng->ev_pool[2 * i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = eng->ev_used;
    eng->ev_used = 0;
    return RASS_OK;
}

void* rass_index_device_rows(rass_index_t* idx) { return idx ? reinterpret_cast<void*>(idx->d_rows) : nullptr; }
void* rass_index_device_tags(rass_index_t* idx) { return idx ? reinterpret_cast<void*>(idx->d_tags) : nullptr; }

static void ivf_free(rass_ivf* v) {
    if (!v) return;
    for (void* p : {(void*)v->d_slab, (void*)v->d_slab_b16, (void*)v->d_slab_i8, (void*)v->d_slab_scale, (void*)v->d_cand_scores, (void*)v->d_cand_rows, (void*)v->d_tags, (void*)v->d_ids, (void*)v->d_centroids, (void*)v->d_list_tile0,
                    (void*)v->d_list_len, (void*)v->d_work_tile, (void*)v->d_work_rows, (void*)v->d_n_work,
                    (void*)v->d_work_mask, (void*)v->d_scanned, (void*)v->d_probe_scores, (void*)v->d_probe_ids,
                    (void*)v->d_tau, (void*)v->d_list_mask, (void*)v->d_pair_scores, (void*)v->d_pair_ids, (void*)v->d_batch})
        if (p) (void)hipFree(p);
    delete v;
}

int rass_ivf_build(rass_index_t* src, const float* centroids, int nlist, const int32_t* assign, rass_ivf_t** out) {
    return rass_ivf_build_ex(src, centroids, nlist, assign, RASS_F32, out);
}

int rass_ivf_build_ex(rass_index_t* src, const float* centroids, int nlist, const int32_t* assign, rass_dtype slab_dtype,
                      rass_ivf_t** out) {
    return rass_ivf_build_prefix(src, centroids, nlist, assign, slab_dtype, -1, out);
}

int rass_ivf_build_prefix(rass_index_t* src, const float* centroids, int nlist, const int32_t* assign,
                          rass_dtype slab_dtype, int64_t n_rows, rass_ivf_t** out) {
    if (!src || !centroids || !assign || !out) return fail(RASS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (nlist < 1 || nlist > 32768) return fail(RASS_ERR_INVALID, "nlist must be in [1, 32768]");
    if (src->dtype != RASS_F32) return fail(RASS_ERR_UNSUPPORTED, "the IVF build needs an fp32 source index");
    if (src->stride > kNarrowStride) return fail(RASS_ERR_UNSUPPORTED, "IVF needs dim <= 1024 (wide rows: flat scan only)");
    if (slab_dtype != RASS_F32 && slab_dtype != RASS_BF16 && slab_dtype != RASS_I8) return fail(RASS_ERR_INVALID, "unknown slab dtype");
    if (slab_dtype == RASS_BF16 && src->stride % 256 != 0)
        return fail(RASS_ERR_UNSUPPORTED, "a bf16 slab needs dim padded to a multiple of 256 (the bf16 scan's K split)");
    const int tile_rows = slab_dtype == RASS_F32 ? 32 : 64;
    rass_engine* eng = src->eng;
    std::lock_guard<std::mutex> lk(src->mu);
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    if (n_rows > src->rows) return fail(RASS_ERR_INVALID, "n_rows exceeds the rows of the source index");
    const int64_t n = n_rows < 0 ? src->rows.load() : n_rows;
    // list lengths over live rows, tile-aligned offsets
    std::vector<int32_t> len((size_t)nlist, 0), tile0((size_t)nlist, 0);
    for (int64_t r = 0; r < n; ++r) {
        if (src->host_deleted[(size_t)(r >> 3)] & (1u << (r & 7))) continue;
        const int32_t l = assign[r];
        if (l < 0 || l >= nlist) return fail(RASS_ERR_INVALID, "assign[] holds a list id outside [0, nlist)");
        len[(size_t)l] += 1;
    }
    int64_t tiles = 0;
    for (int l = 0; l < nlist; ++l) {
        tile0[(size_t)l] = (int32_t)tiles;
        tiles += (len[(size_t)l] + tile_rows - 1) / tile_rows;
    }
    if (tiles * tile_rows > 0x7fffffc0LL) return fail(RASS_ERR_UNSUPPORTED, "slab too large for one IVF shard");
    const int64_t slab_rows = std::max<int64_t>(tiles, 1) * tile_rows;
    std::vector<int64_t> src_of((size_t)slab_rows, -1);
    std::vector<int32_t> fill((size_t)nlist, 0);
    for (int64_t r = 0; r < n; ++r) {  // ascending source id inside every list
        if (src->host_deleted[(size_t)(r >> 3)] & (1u << (r & 7))) continue;
        const int32_t l = assign[r];
        src_of[(size_t)((int64_t)tile0[(size_t)l] * tile_rows + fill[(size_t)l]++)] = r;
    }
    rass_ivf* v = new (std::nothrow) rass_ivf();
    if (!v) return fail(RASS_ERR_OOM, "host allocation failed");
    v->eng = eng;
    v->dtype = slab_dtype;
    v->tile_rows = tile_rows;
    v->dim = src->dim;
    v->stride = src->stride;
    v->nlist = nlist;
    v->rows = 0;
    for (int l = 0; l < nlist; ++l) v->rows += len[(size_t)l];
    v->src_rows = n;
    v->pos_of.assign((size_t)n, -1);
    for (int64_t d = 0; d < slab_rows; ++d)
        if (src_of[(size_t)d] >= 0) v->pos_of[(size_t)src_of[(size_t)d]] = (int32_t)d;
    v->slab_rows = slab_rows;
    v->total_tiles = std::max<int64_t>(tiles, 1);
    v->any_tags = src->has_tags;
    hipStream_t st = eng->stream;
    const int64_t cent_rows = ((int64_t)nlist + 15) / 16 * 16;
#define IVF_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            ivf_free(v);                                                                                    \
            return fail(_e == hipErrorOutOfMemory ? RASS_ERR_OOM : RASS_ERR_HIP,                            \
                        std::string("ivf build: ") + #expr + ": " + hipGetErrorString(_e));                 \
        }                                                                                                   \
    } while (0)
    if (slab_dtype == RASS_BF16)
        IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_slab_b16), (size_t)slab_rows * v->stride * 2));
    else
        IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_slab), (size_t)slab_rows * v->stride * 4));
    if (slab_dtype == RASS_I8) {
        v->stride_i8 = (v->stride + 511) / 512 * 512;
        IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_slab_i8), (size_t)slab_rows * v->stride_i8));
        IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_slab_scale), (size_t)slab_rows * 4));
        IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_cand_scores), RASS_MAX_QBATCH * RASS_MAX_K * 4));
        IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_cand_rows), RASS_MAX_QBATCH * RASS_MAX_K * 8));
        IVF_TRY(hipMemsetAsync(v->d_slab_i8, 0, (size_t)slab_rows * v->stride_i8, st));
    }
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_tags), (size_t)slab_rows * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_ids), (size_t)slab_rows * 8));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_centroids), (size_t)cent_rows * v->stride * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_list_tile0), (size_t)nlist * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_list_len), (size_t)nlist * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_work_tile), (size_t)v->total_tiles * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_work_rows), (size_t)v->total_tiles * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_work_mask), (size_t)v->total_tiles * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_n_work), 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_scanned), 8));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_probe_scores), RASS_MAX_QBATCH * RASS_MAX_K * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_probe_ids), RASS_MAX_QBATCH * RASS_MAX_K * 8));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_tau), RASS_MAX_QBATCH * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_list_mask), (size_t)nlist * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_pair_scores), 2 * RASS_MAX_QBATCH * RASS_MAX_K * 4));
    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_pair_ids), 2 * RASS_MAX_QBATCH * RASS_MAX_K * 8));
    IVF_TRY(hipMemcpyAsync(v->d_ids, src_of.data(), (size_t)slab_rows * 8, hipMemcpyHostToDevice, st));
    IVF_TRY(hipMemcpyAsync(v->d_list_tile0, tile0.data(), (size_t)nlist * 4, hipMemcpyHostToDevice, st));
    IVF_TRY(hipMemcpyAsync(v->d_list_len, len.data(), (size_t)nlist * 4, hipMemcpyHostToDevice, st));
    if (slab_dtype == RASS_BF16)
        IVF_TRY(rass::launch_permute_rows_tile16_bf16(src->d_rows, v->d_slab_b16, v->stride, v->d_ids, slab_rows, st));
    else
        IVF_TRY(rass::launch_permute_rows_tile16(src->d_rows, v->d_slab, v->stride, v->d_ids, slab_rows, st));
    if (slab_dtype == RASS_I8)
        IVF_TRY(rass::launch_quantize_tile16_i8(v->d_slab, v->d_slab_i8, v->d_slab_scale, v->stride, v->stride_i8, 0, slab_rows / 16, st));
    // tags: permuted on the host (small), padding rows get 0
    {
        std::vector<int32_t> tags((size_t)std::max<int64_t>(n, 1), 0), ptags((size_t)slab_rows, 0);
        if (n > 0) {
            IVF_TRY(hipMemcpyAsync(tags.data(), src->d_tags, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            IVF_TRY(hipStreamSynchronize(st));
        }
        for (int64_t d = 0; d < slab_rows; ++d)
            if (src_of[(size_t)d] >= 0) ptags[(size_t)d] = tags[(size_t)src_of[(size_t)d]];
        IVF_TRY(hipMemcpyAsync(v->d_tags, ptags.data(), (size_t)slab_rows * 4, hipMemcpyHostToDevice, st));
        IVF_TRY(hipStreamSynchronize(st));
    }
    // centroids: normalise + pack through the engine's staging buffer
    IVF_TRY(hipMemsetAsync(v->d_centroids, 0, (size_t)cent_rows * v->stride * 4, st));
    {
        std::lock_guard<std::mutex> elk(eng->mu);
        for (int64_t done = 0; done < nlist; done += kStageRows) {
            const int64_t m = std::min<int64_t>(kStageRows, nlist - done);
            IVF_TRY(hipMemcpyAsync(eng->d_stage, centroids + done * v->dim, (size_t)m * v->dim * 4,
                                   hipMemcpyHostToDevice, st));
            IVF_TRY(rass::launch_pack_rows_tile16(eng->d_stage, v->dim, v->d_centroids, v->stride, done, m, v->dim, 1, st));
            IVF_TRY(hipStreamSynchronize(st));
        }
    }
#undef IVF_TRY
    *out = v;
    return RASS_OK;
}

void rass_ivf_destroy(rass_ivf_t* v) {
    if (!v) return;
    (void)hipSetDevice(v->eng->device);
    (void)hipStreamSynchronize(v->eng->stream);
    ivf_free(v);
}

// ---- IVF persistence: header + list table + slab ids + tags + centroid slab + row slab (raw tile16)
struct IvfSaveHeader {
    char magic[8];
    int32_t version, dim, nlist, any_tags;
    int64_t stride, rows, slab_rows, total_tiles, cent_rows;
};

static bool dev_to_file(FILE* f, const void* d_src, size_t bytes, hipStream_t st, std::vector<unsigned char>& buf) {
    const unsigned char* p = static_cast<const unsigned char*>(d_src);
    for (size_t done = 0; done < bytes;) {
        const size_t m = std::min(buf.size(), bytes - done);
        if (hipMemcpyAsync(buf.data(), p + done, m, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        if (fwrite(buf.data(), 1, m, f) != m) return false;
        done += m;
    }
    return true;
}

static bool file_to_dev(FILE* f, void* d_dst, size_t bytes, hipStream_t st, std::vector<unsigned char>& buf) {
    unsigned char* p = static_cast<unsigned char*>(d_dst);
    for (size_t done = 0; done < bytes;) {
        const size_t m = std::min(buf.size(), bytes - done);
        if (fread(buf.data(), 1, m, f) != m) return false;
        if (hipMemcpyAsync(p + done, buf.data(), m, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        done += m;
    }
    return true;
}

int rass_ivf_save(rass_ivf_t* v, const char* path) {
    if (!v || !path) return fail(RASS_ERR_INVALID, "NULL argument");
    rass_engine* eng = v->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    std::lock_guard<std::mutex> lk(eng->mu);
    hipStream_t st = eng->stream;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RASS_ERR_IO, std::string("cannot open for write: ") + path);
    IvfSaveHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "RASSIVF1", 8);
    // 3 / 4 (since round 4) = 1 / 2 followed by one int64: the source rows the IVF covers (rass_ivf_covered_rows).
    // 2, 4: the row slab is bf16 (tile16b) with lists on 64-row tiles
    // 5: the fp32 slab with lists on 64-row tiles of an int8 IVF (the int8 copy and its scales are rebuilt by the load)
    h.version = v->dtype == RASS_BF16 ? 4 : v->dtype == RASS_I8 ? 5 : 3;
    h.dim = v->dim;
    h.nlist = v->nlist;
    h.any_tags = v->any_tags ? 1 : 0;
    h.stride = v->stride;
    h.rows = v->rows;
    h.slab_rows = v->slab_rows;
    h.total_tiles = v->total_tiles;
    h.cent_rows = ((int64_t)v->nlist + 15) / 16 * 16;
    std::vector<unsigned char> buf((size_t)32 << 20);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && fwrite(&v->src_rows, sizeof(int64_t), 1, f) == 1;
    ok = ok && dev_to_file(f, v->d_list_tile0, (size_t)v->nlist * 4, st, buf);
    ok = ok && dev_to_file(f, v->d_list_len, (size_t)v->nlist * 4, st, buf);
    ok = ok && dev_to_file(f, v->d_ids, (size_t)v->slab_rows * 8, st, buf);
    ok = ok && dev_to_file(f, v->d_tags, (size_t)v->slab_rows * 4, st, buf);
    ok = ok && dev_to_file(f, v->d_centroids, (size_t)h.cent_rows * v->stride * 4, st, buf);
    ok = ok && (v->dtype == RASS_BF16 ? dev_to_file(f, v->d_slab_b16, (size_t)v->slab_rows * v->stride * 2, st, buf)
                                       : dev_to_file(f, v->d_slab, (size_t)v->slab_rows * v->stride * 4, st, buf));
    ok = ok && fflush(f) == 0 && fsync(fileno(f)) == 0;
    ok = (fclose(f) == 0) && ok;
    return ok ? RASS_OK : fail(RASS_ERR_IO, std::string("ivf save failed (short write or device read): ") + path);
}

int rass_ivf_load(rass_engine_t* eng, const char* path, rass_ivf_t** out) {
    if (!eng || !path || !out) return fail(RASS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RASS_ERR_IO, std::string("cannot open for read: ") + path);
    IvfSaveHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "RASSIVF1", 8) != 0 || h.version < 1 || h.version > 5) {
        fclose(f);
        return fail(RASS_ERR_IO, "not a rass IVF file");
    }
    int64_t src_rows = -1;   // versions 1 / 2 do not carry it: taken from the slab's ids below
    const int64_t extra = h.version >= 3 ? (int64_t)sizeof(int64_t) : 0;
    if (extra && (fread(&src_rows, sizeof(int64_t), 1, f) != 1 || src_rows < 0)) {
        fclose(f);
        return fail(RASS_ERR_IO, "IVF file is truncated / corrupt");
    }
    const bool b16 = h.version == 2 || h.version == 4;
    const bool i8 = h.version == 5;
    const int tile_rows = (b16 || i8) ? 64 : 32;
    const int64_t esize = b16 ? 2 : 4;
    const int64_t cent_rows = ((int64_t)h.nlist + 15) / 16 * 16;
    bool sane = h.dim == eng->dim && h.stride == pad128(h.dim) && h.nlist >= 1 && h.nlist <= 32768 && h.rows >= 0 &&
                h.slab_rows >= tile_rows && h.slab_rows % tile_rows == 0 && h.slab_rows <= 0x7fffffc0LL &&
                h.total_tiles == h.slab_rows / tile_rows && h.cent_rows == cent_rows && h.rows <= h.slab_rows &&
                (!b16 || h.stride % 256 == 0) && h.stride <= kNarrowStride;
    if (sane) {  // the header must agree with the file length before anything is allocated from it
        const long body = ftell(f);
        int64_t len = -1;
        if (body >= 0 && fseek(f, 0, SEEK_END) == 0) len = (int64_t)ftell(f);
        const int64_t need = (int64_t)sizeof(h) + extra + (int64_t)h.nlist * 8 + h.slab_rows * 12 + cent_rows * h.stride * 4 +
                             h.slab_rows * h.stride * esize;
        sane = body >= 0 && len == need && fseek(f, body, SEEK_SET) == 0;
    }
    if (!sane) {
        fclose(f);
        return fail(RASS_ERR_IO, "IVF file does not match the engine (dim) or is truncated / corrupt");
    }
    rass_ivf* v = new (std::nothrow) rass_ivf();
    if (!v) {
        fclose(f);
        return fail(RASS_ERR_OOM, "host allocation failed");
    }
    v->eng = eng;
    v->dim = h.dim;
    v->stride = h.stride;
    v->nlist = h.nlist;
    v->rows = h.rows;
    v->slab_rows = h.slab_rows;
    v->total_tiles = h.total_tiles;
    v->any_tags = h.any_tags != 0;
    v->dtype = b16 ? RASS_BF16 : i8 ? RASS_I8 : RASS_F32;
    v->tile_rows = tile_rows;
    v->stride_i8 = (h.stride + 511) / 512 * 512;
    std::lock_guard<std::mutex> lk(eng->mu);
    hipStream_t st = eng->stream;
    auto alloc = [&](void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; };
    bool ok = (b16 ? alloc((void**)&v->d_slab_b16, (size_t)h.slab_rows * h.stride * 2)
                   : alloc((void**)&v->d_slab, (size_t)h.slab_rows * h.stride * 4)) &&
              alloc((void**)&v->d_tags, (size_t)h.slab_rows * 4) &&
              alloc((void**)&v->d_ids, (size_t)h.slab_rows * 8) && alloc((void**)&v->d_centroids, (size_t)cent_rows * h.stride * 4) &&
              alloc((void**)&v->d_list_tile0, (size_t)h.nlist * 4) && alloc((void**)&v->d_list_len, (size_t)h.nlist * 4) &&
              alloc((void**)&v->d_work_tile, (size_t)h.total_tiles * 4) && alloc((void**)&v->d_work_rows, (size_t)h.total_tiles * 4) &&
              alloc((void**)&v->d_work_mask, (size_t)h.total_tiles * 4) && alloc((void**)&v->d_n_work, 4) &&
              alloc((void**)&v->d_scanned, 8) && alloc((void**)&v->d_probe_scores, RASS_MAX_QBATCH * RASS_MAX_K * 4) &&
              alloc((void**)&v->d_probe_ids, RASS_MAX_QBATCH * RASS_MAX_K * 8) && alloc((void**)&v->d_tau, RASS_MAX_QBATCH * 4) &&
              alloc((void**)&v->d_list_mask, (size_t)h.nlist * 4) &&
              alloc((void**)&v->d_pair_scores, 2 * RASS_MAX_QBATCH * RASS_MAX_K * 4) &&
              alloc((void**)&v->d_pair_ids, 2 * RASS_MAX_QBATCH * RASS_MAX_K * 8);
    if (ok && i8)
        ok = alloc((void**)&v->d_slab_i8, (size_t)h.slab_rows * v->stride_i8) && alloc((void**)&v->d_slab_scale, (size_t)h.slab_rows * 4) &&
             alloc((void**)&v->d_cand_scores, RASS_MAX_QBATCH * RASS_MAX_K * 4) && alloc((void**)&v->d_cand_rows, RASS_MAX_QBATCH * RASS_MAX_K * 8);
    if (!ok) {
        fclose(f);
        ivf_free(v);
        return fail(RASS_ERR_OOM, "ivf load: device allocation failed");
    }
    std::vector<unsigned char> buf((size_t)32 << 20);
    ok = file_to_dev(f, v->d_list_tile0, (size_t)h.nlist * 4, st, buf) && file_to_dev(f, v->d_list_len, (size_t)h.nlist * 4, st, buf) &&
         file_to_dev(f, v->d_ids, (size_t)h.slab_rows * 8, st, buf) && file_to_dev(f, v->d_tags, (size_t)h.slab_rows * 4, st, buf) &&
         file_to_dev(f, v->d_centroids, (size_t)cent_rows * h.stride * 4, st, buf) &&
         (b16 ? file_to_dev(f, v->d_slab_b16, (size_t)h.slab_rows * h.stride * 2, st, buf)
              : file_to_dev(f, v->d_slab, (size_t)h.slab_rows * h.stride * 4, st, buf));
    fclose(f);
    if (ok && i8)   // the int8 copy is a function of the fp32 slab: rebuilt, not stored
        ok = hipMemsetAsync(v->d_slab_i8, 0, (size_t)h.slab_rows * v->stride_i8, st) == hipSuccess &&
             rass::launch_quantize_tile16_i8(v->d_slab, v->d_slab_i8, v->d_slab_scale, v->stride, v->stride_i8, 0, h.slab_rows / 16, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    if (!ok) {
        ivf_free(v);
        return fail(RASS_ERR_IO, "ivf load: short read or upload failure");
    }
    // the list table must index inside the slab: a corrupt table would send the probe out of bounds
    {
        std::vector<int32_t> t0((size_t)h.nlist), len((size_t)h.nlist);
        bool good = hipMemcpy(t0.data(), v->d_list_tile0, (size_t)h.nlist * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(len.data(), v->d_list_len, (size_t)h.nlist * 4, hipMemcpyDeviceToHost) == hipSuccess;
        int64_t tiles = 0;
        for (int l = 0; good && l < h.nlist; ++l) {
            good = len[(size_t)l] >= 0 && t0[(size_t)l] == tiles;
            tiles += (len[(size_t)l] + tile_rows - 1) / tile_rows;
        }
        if (!good || std::max<int64_t>(tiles, 1) != h.total_tiles) {
            ivf_free(v);
            return fail(RASS_ERR_IO, "ivf load: inconsistent list table");
        }
    }
    // source row -> slab position (rass_ivf_delete), from the slab's ids and tags (-1 tag = tombstoned after the build)
    {
        std::vector<int64_t> ids((size_t)h.slab_rows);
        std::vector<int32_t> tags((size_t)h.slab_rows);
        if (hipMemcpy(ids.data(), v->d_ids, (size_t)h.slab_rows * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(tags.data(), v->d_tags, (size_t)h.slab_rows * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            ivf_free(v);
            return fail(RASS_ERR_HIP, "ivf load: reading back the slab ids failed");
        }
        int64_t max_id = -1;
        for (int64_t d = 0; d < h.slab_rows; ++d) max_id = std::max(max_id, ids[(size_t)d]);
        if (src_rows < 0) src_rows = max_id + 1;
        if (max_id >= src_rows) {
            ivf_free(v);
            return fail(RASS_ERR_IO, "ivf load: a slab id lies outside the covered source rows");
        }
        v->src_rows = src_rows;
        v->pos_of.assign((size_t)src_rows, -1);
        for (int64_t d = 0; d < h.slab_rows; ++d)
            if (ids[(size_t)d] >= 0 && tags[(size_t)d] != -1) v->pos_of[(size_t)ids[(size_t)d]] = (int32_t)d;
    }
    *out = v;
    return RASS_OK;
}

int64_t rass_ivf_rows(const rass_ivf_t* v) { return v ? v->rows : 0; }
int rass_ivf_nlist(const rass_ivf_t* v) { return v ? v->nlist : 0; }
int rass_ivf_dtype(const rass_ivf_t* v) { return v ? v->dtype : -1; }
int64_t rass_ivf_covered_rows(const rass_ivf_t* v) { return v ? v->src_rows : 0; }

int rass_ivf_delete(rass_ivf_t* v, int64_t src_row) {
    if (!v) return fail(RASS_ERR_INVALID, "NULL argument");
    rass_engine* eng = v->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    // the engine mutex: a search holds it for its whole enqueue sequence, so the fill cannot land between a probe's
    // plan and its fine scan (as rass_index_delete)
    std::lock_guard<std::mutex> lk(eng->mu);
    if (src_row < 0 || src_row >= v->src_rows) return RASS_OK;   // not covered: the row lives in the flat delta only
    const int32_t pos = v->pos_of[(size_t)src_row];
    if (pos < 0) return RASS_OK;                                  // already gone
    const int32_t dead = -1;
    HIP_TRY(hipMemcpyAsync(v->d_tags + pos, &dead, 4, hipMemcpyHostToDevice, eng->stream));
    HIP_TRY(hipStreamSynchronize(eng->stream));                   // `dead` is a stack variable
    v->pos_of[(size_t)src_row] = -1;
    v->any_tags = true;
    v->rows -= 1;
    return RASS_OK;
}

// Caller holds eng->mu (the probe scratch of the IVF object and the engine scratch are shared).
static int ivf_search_locked(rass_ivf_t* v, const float* d_queries, int nq, int k, int nprobe,
                             const int32_t* d_q_filter, float* d_out_scores, int64_t* d_out_ids,
                             const int32_t* d_q_filter_mask = nullptr) {
    if (!v || !d_queries || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nprobe < 1) return fail(RASS_ERR_INVALID, "nprobe must be >= 1");
    if (nq < 1 || nq > RASS_MAX_QBATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, RASS_MAX_QBATCH]");
    rass_engine* eng = v->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    hipStream_t st = eng->stream;
    const int np = std::min(nprobe, v->nlist);
    const int n_ctiles = (v->nlist + 31) / 32;
    if (np <= RASS_MAX_K || n_ctiles > kMaxGrid) {
        // (i) coarse: top-nprobe centroids per query with the flat fused scan
        rc = scan_launch(v->d_centroids, v->nlist, v->stride, nullptr, d_queries, v->dim, v->dim, nq, nullptr,
                         std::min(np, RASS_MAX_K), 0, v->d_probe_scores, v->d_probe_ids, eng->d_scratch,
                         eng->scratch_bytes, eng->n_cus, st);
        if (rc != RASS_OK) return rc;
        // (ii) plan: union of probed lists -> work tiles with per-tile query masks
        HIP_TRY(rass::launch_plan_probe(v->d_probe_ids, nq, std::min(np, RASS_MAX_K), v->nlist, v->d_list_tile0,
                                        v->d_list_len, v->d_work_tile, v->d_work_rows, v->d_work_mask, v->d_n_work,
                                        v->d_scanned, st, nullptr, v->tile_rows));
    } else {
        // nprobe > 32: one workgroup per 32-centroid tile with k = 32 leaves EVERY centroid score in
        // the per-workgroup lists; radix-select the nprobe-th best per query, mask by threshold
        const ScratchLayout L = scratch_layout(nq, RASS_MAX_K);
        unsigned char* ws = eng->d_scratch;
        float* q_padded = reinterpret_cast<float*>(ws + L.q_padded);
        float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
        int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
        const int nq_pad = nq <= 16 ? 16 : 32;
        HIP_TRY(rass::launch_normalize_rows_f32(d_queries, v->dim, q_padded, v->stride, nq, v->dim, st, nq_pad));
        rass::ScanArgs a;
        a.corpus = v->d_centroids;
        a.row_tag = nullptr;
        a.q_padded = q_padded;
        a.q_filter = nullptr;
        a.part_scores = part_scores;
        a.part_ids = part_ids;
        a.row_stride = v->stride;
        a.id_base = 0;
        a.n_rows = v->nlist;
        a.nq = nq;
        a.k = RASS_MAX_K;
        HIP_TRY(rass::launch_scan_topk_f32(a, n_ctiles, st));
        HIP_TRY(rass::launch_ivf_threshold(part_scores, part_ids, n_ctiles, nq, np, v->d_tau, st));
        HIP_TRY(rass::launch_ivf_mask_from_scores(part_scores, part_ids, n_ctiles, nq, v->nlist, v->d_tau,
                                                  v->d_list_mask, st));
        HIP_TRY(rass::launch_plan_probe(v->d_probe_ids, nq, 1, v->nlist, v->d_list_tile0, v->d_list_len, v->d_work_tile,
                                        v->d_work_rows, v->d_work_mask, v->d_n_work, v->d_scanned, st, v->d_list_mask,
                                        v->tile_rows));
    }
    // (iii) fine: the same fused scan over the planned tiles; slab positions -> source ids in the merge
    IvfPlan plan{v->d_work_tile, v->d_work_rows, v->d_work_mask, v->d_n_work, v->total_tiles};
    const bool need_tags = v->any_tags || d_q_filter != nullptr;
    // (both branches above left the batch's normalised queries at the head of the engine scratch, at this stride)
    if (v->dtype == RASS_BF16) {
        // the bf16 scan over the planned 64-row tiles: queries rounded to bf16, fp32 accumulation, slab positions -> source
        // ids in the merge.  Scores are those of a flat bf16 index holding the same rows.
        const ScratchLayout L = scratch_layout(RASS_MAX_QBATCH, RASS_MAX_K);
        unsigned char* ws = eng->d_scratch;
        const float* q_padded = reinterpret_cast<const float*>(ws + L.q_padded);
        float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
        int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
        unsigned short* q_bf16 = reinterpret_cast<unsigned short*>(ws + L.q_bf16);
        const int nq_pad = nq <= 16 ? 16 : 32;
        HIP_TRY(rass::launch_queries_to_bf16(q_padded, q_bf16, (int64_t)nq_pad * v->stride, st));
        int grid = (int)std::min<int64_t>(std::max<int64_t>(v->total_tiles, 1), std::min(eng->n_cus, kMaxGrid));
        if ((int64_t)grid * k > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / k;
        rass::ScanBf16Args a;
        a.corpus = v->d_slab_b16;
        a.row_tag = need_tags ? v->d_tags : nullptr;
        a.q_bf16 = q_bf16;
        a.q_filter = d_q_filter;
        a.part_scores = part_scores;
        a.part_ids = part_ids;
        a.row_stride = v->stride;
        a.n_rows = (int)v->slab_rows;
        a.nq = nq;
        a.k = k;
        a.id_base = 0;
        a.work_tile = v->d_work_tile;
        a.work_rows = v->d_work_rows;
        a.work_mask = v->d_work_mask;
        a.n_work = v->d_n_work;
        a.q_filter_mask = d_q_filter_mask;
        const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        HIP_TRY(rass::launch_scan_bf16_topk(a, grid, st));
        if (timed) {
            HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
            eng->ev_used += 1;
        }
        HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, nq, k, d_out_scores, d_out_ids, st, v->d_ids));
        return RASS_OK;
    }
    if (v->dtype == RASS_I8) {
        // the int8 scan over the planned 64-row tiles keeps 32 candidates per query (slab positions); the re-rank rescores
        // them exactly from the fp32 slab in the flat kernel's order and returns the best k under (score desc, source id asc)
        if (k > kPrefilterMaxK) return fail(RASS_ERR_UNSUPPORTED, "an int8 IVF slab serves k <= 16 (32 candidates per query)");
        const ScratchLayout L = scratch_layout(RASS_MAX_QBATCH, RASS_MAX_K);
        unsigned char* ws = eng->d_scratch;
        const float* q_padded = reinterpret_cast<const float*>(ws + L.q_padded);
        float* part_scores = reinterpret_cast<float*>(ws + L.part_scores);
        int64_t* part_ids = reinterpret_cast<int64_t*>(ws + L.part_ids);
        signed char* q_i8 = reinterpret_cast<signed char*>(ws + L.q_bf16);
        const int nq_pad = nq <= 16 ? 16 : 32;
        const int kc = RASS_MAX_K;
        HIP_TRY(rass::launch_queries_to_i8(q_padded, q_i8, nq_pad, v->stride, v->stride_i8, st));
        int grid = (int)std::min<int64_t>(std::max<int64_t>(v->total_tiles, 1), std::min(eng->n_cus, kMaxGrid));
        if ((int64_t)grid * kc > rass::kMergeMaxCandidates) grid = rass::kMergeMaxCandidates / kc;
        rass::ScanI8Args a;
        a.corpus = v->d_slab_i8;
        a.row_scale = v->d_slab_scale;
        a.row_tag = need_tags ? v->d_tags : nullptr;
        a.q_i8 = q_i8;
        a.q_filter = d_q_filter;
        a.q_filter_mask = d_q_filter_mask;
        a.part_scores = part_scores;
        a.part_ids = part_ids;
        a.row_stride = v->stride_i8;
        a.n_rows = (int)v->slab_rows;
        a.nq = nq;
        a.k = kc;
        a.work_tile = v->d_work_tile;
        a.work_rows = v->d_work_rows;
        a.work_mask = v->d_work_mask;
        a.n_work = v->d_n_work;
        const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        HIP_TRY(rass::launch_scan_i8_topk(a, grid, st));
        if (timed) {
            HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
            eng->ev_used += 1;
        }
        HIP_TRY(rass::launch_merge_topk(part_scores, part_ids, grid, nq, kc, v->d_cand_scores, v->d_cand_rows, st));
        HIP_TRY(rass::launch_rerank_f32(v->d_slab, v->stride, q_padded, v->d_cand_rows, nq, kc, k, 0, d_out_scores, d_out_ids, st,
                                        0, 0, v->d_ids));
        return RASS_OK;
    }
    ScanExt ext;
    ext.d_q_mask = d_q_filter_mask;
    return scan_launch(v->d_slab, v->slab_rows, v->stride, need_tags ? v->d_tags : nullptr, d_queries, v->dim, v->dim,
                       nq, d_q_filter, k, 0, d_out_scores, d_out_ids, eng->d_scratch, eng->scratch_bytes, eng->n_cus,
                       st, eng, &plan, v->d_ids, d_q_filter_mask ? &ext : nullptr, /*queries_prepared=*/true);
}

// One launch group of an IVF + delta search; the caller holds eng->mu.  List 0 = the probe (source ordinals through the
// slab's id map), list 1 = the exact scan of the source rows the IVF does not cover (ordinals through id_base); the
// final merge orders them by (score desc, ordinal asc) and maps ordinals to the source's caller-assigned ids, if any.
static int ivf_delta_group_locked(rass_ivf_t* v, rass_index* flat, const float* d_queries, int nq, int k, int nprobe,
                                  const int32_t* d_q_filter, const int32_t* d_q_filter_mask, float* d_out_scores,
                                  int64_t* d_out_ids) {
    rass_engine* eng = v->eng;
    if (!flat || flat->eng != eng) return fail(RASS_ERR_INVALID, "the delta index must live on the IVF's engine");
    if (flat->dtype != RASS_F32 || flat->stride != v->stride || flat->dim != v->dim)
        return fail(RASS_ERR_UNSUPPORTED, "the delta index must be the fp32 index the IVF was built from");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    if (d_q_filter_mask && !d_q_filter) return fail(RASS_ERR_INVALID, "q_filter_mask without q_filter");
    const int64_t rows = flat->rows.load(std::memory_order_acquire);
    const int64_t covered = v->src_rows;
    if (covered > rows) return fail(RASS_ERR_INVALID, "the IVF covers more rows than the delta index holds");
    const int64_t delta = rows - covered;
    if (delta > 0 && covered % 32 != 0)
        return fail(RASS_ERR_UNSUPPORTED, "an IVF with a delta must cover a multiple of 32 source rows (rass_ivf_build_prefix)");
    const bool gid = flat->has_gid.load(std::memory_order_acquire);
    hipStream_t st = eng->stream;
    float* ps = v->d_pair_scores;
    int64_t* pi = v->d_pair_ids;
    int rc = ivf_search_locked(v, d_queries, nq, k, nprobe, d_q_filter, ps, pi, d_q_filter_mask);
    if (rc != RASS_OK) return rc;
    int n_lists = 1;
    if (delta > 0) {
        const bool need_tags = (flat->deleted.load(std::memory_order_acquire) > 0) || (d_q_filter != nullptr);
        ScanExt ext;
        ext.d_q_mask = d_q_filter_mask;
        rc = scan_launch(flat->d_rows + covered * flat->stride, delta, flat->stride,
                         need_tags ? flat->d_tags + covered : nullptr, d_queries, flat->dim, flat->dim, nq, d_q_filter, k,
                         covered, ps + (int64_t)nq * k, pi + (int64_t)nq * k, eng->d_scratch, eng->scratch_bytes,
                         eng->n_cus, st, eng, nullptr, nullptr, d_q_filter_mask ? &ext : nullptr);
        if (rc != RASS_OK) return rc;
        n_lists = 2;
    }
    HIP_TRY(rass::launch_merge_topk(ps, pi, n_lists, nq, k, d_out_scores, d_out_ids, st, gid ? flat->d_gid : nullptr));
    return RASS_OK;
}

int rass_ivf_search_delta_device(rass_ivf_t* v, rass_index_t* flat, const float* d_queries, int nq, int k, int nprobe,
                                 const int32_t* d_q_filter, const int32_t* d_q_filter_mask, float* d_out_scores,
                                 int64_t* d_out_ids) {
    if (!v || !flat || !d_queries || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 1 || nq > RASS_MAX_QBATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, RASS_MAX_QBATCH]");
    int rc = set_device(v->eng);
    if (rc != RASS_OK) return rc;
    std::lock_guard<std::mutex> lk(v->eng->mu);
    return ivf_delta_group_locked(v, flat, d_queries, nq, k, nprobe, d_q_filter, d_q_filter_mask, d_out_scores, d_out_ids);
}

int rass_ivf_search_delta(rass_ivf_t* v, rass_index_t* flat, const float* queries, int nq, int k, int nprobe,
                          const int32_t* q_filter, const int32_t* q_filter_mask, float* out_scores, int64_t* out_ids,
                          int64_t* scanned_rows) {
    if (!v || !flat || !out_scores || !out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 0 || (nq > 0 && !queries)) return fail(RASS_ERR_INVALID, "bad queries / nq");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    if (q_filter_mask && !q_filter) return fail(RASS_ERR_INVALID, "q_filter_mask without q_filter");
    rass_engine* eng = v->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    int64_t scanned_total = 0;
    SlotGuard guard(eng);
    HostSlot* sl = guard.sl;
    for (int done = 0; done < nq;) {
        const int b = std::min(RASS_MAX_QBATCH, nq - done);
        memcpy(sl->h_q, queries + (int64_t)done * v->dim, (size_t)b * v->dim * 4);
        if (q_filter) memcpy(sl->h_filter, q_filter + done, (size_t)b * 4);
        if (q_filter_mask) memcpy(sl->h_mask, q_filter_mask + done, (size_t)b * 4);
        {
            std::lock_guard<std::mutex> lk(eng->mu);
            hipStream_t st = eng->stream;
            HIP_TRY(hipMemcpyAsync(eng->d_qraw, sl->h_q, (size_t)b * v->dim * 4, hipMemcpyHostToDevice, st));
            if (q_filter) HIP_TRY(hipMemcpyAsync(eng->d_qfilter, sl->h_filter, (size_t)b * 4, hipMemcpyHostToDevice, st));
            if (q_filter_mask) HIP_TRY(hipMemcpyAsync(eng->d_qmask, sl->h_mask, (size_t)b * 4, hipMemcpyHostToDevice, st));
            rc = ivf_delta_group_locked(v, flat, eng->d_qraw, b, k, nprobe, q_filter ? eng->d_qfilter : nullptr,
                                        q_filter_mask ? eng->d_qmask : nullptr, eng->d_out_scores, eng->d_out_ids);
            if (rc != RASS_OK) return rc;
            HIP_TRY(hipMemcpyAsync(sl->h_out_s, eng->d_out_scores, (size_t)b * k * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sl->h_out_i, eng->d_out_ids, (size_t)b * k * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sl->h_scanned, v->d_scanned, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(sl->done, st));
        }
        HIP_TRY(hipEventSynchronize(sl->done));
        memcpy(out_scores + (int64_t)done * k, sl->h_out_s, (size_t)b * k * 4);
        memcpy(out_ids + (int64_t)done * k, sl->h_out_i, (size_t)b * k * 8);
        scanned_total += *sl->h_scanned + std::max<int64_t>(0, flat->rows.load() - v->src_rows);
        done += b;
    }
    if (scanned_rows) *scanned_rows = scanned_total;
    return RASS_OK;
}

// RASS_IVF_BATCH_FINE=groups: one fine-scan launch per group (the A/B of kIvfGroups); default: one launch for all groups
static bool ivf_batch_one_launch() {
    const char* e = getenv("RASS_IVF_BATCH_FINE");
    return !(e && e[0] == 'g');
}

// A whole batch of launch groups (nq <= 1 024 queries) of an IVF probe with 4 + G launches instead of 5 G: ONE normalise,
// ONE grouped coarse scan (kFlatGroups: every group's 32 queries over the centroid slab, 8 workgroups per group), ONE plan
// launch (a workgroup per group; the coarse lists are merged inside it), the G fine scans over their groups' work lists,
// ONE grouped merge.  Same lists probed, same scores, same (score desc, id asc) order as rass_ivf_search_device group by
// group (tests/test_gpu_ivf.py).  nprobe <= 32 (deeper probes go group by group through the threshold path).
static int ivf_search_batch_locked(rass_ivf_t* v, const float* d_queries, int nq, int k, int nprobe,
                                   const int32_t* d_q_filter, float* d_out_scores, int64_t* d_out_ids,
                                   int64_t* d_scanned_per_group) {
    rass_engine* eng = v->eng;
    hipStream_t st = eng->stream;
    const int np = std::min(nprobe, v->nlist);
    const int G = (nq + RASS_MAX_QBATCH - 1) / RASS_MAX_QBATCH;
    const int n_ctiles = (v->nlist + 31) / 32;
    const int wpg = std::max(1, std::min(8, std::min(n_ctiles, 256 / np)));      // coarse workgroups per group
    const int64_t stride = v->stride;
    // fine-scan workgroups per group.  fp32 slab: ALL groups' fine scans are one launch (kIvfGroups) — the more groups, the
    // fewer workgroups each (32 at 32 groups: 1 024 in all, dispatched in group order, no launch boundary between groups);
    // bf16 slab: one launch per group over the whole chip.
    const bool one_fine_launch = v->dtype == RASS_F32 && ivf_batch_one_launch();
    const bool i8 = v->dtype == RASS_I8;
    if (i8 && k > kPrefilterMaxK) return fail(RASS_ERR_UNSUPPORTED, "an int8 IVF slab serves k <= 16 (32 candidates per query)");
    const int kf = i8 ? RASS_MAX_K : k;   // entries per fine list: the int8 scan keeps 32 candidates whatever k is
    int fgrid = (int)std::min<int64_t>(std::max<int64_t>(v->total_tiles, 1), std::min(eng->n_cus, kMaxGrid));
    if (one_fine_launch) fgrid = std::max(1, std::min(fgrid, std::max(32, 1024 / G)));
    if ((int64_t)fgrid * kf > rass::kMergeMaxCandidates) fgrid = rass::kMergeMaxCandidates / kf;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    const size_t o_q = off;        off = up(off + (size_t)G * 32 * stride * 4);
    const size_t o_qb = off;       off = up(off + (v->dtype != RASS_F32 ? (size_t)G * 32 * kMaxStride * 2 : 0));   // bf16 / int8 queries
    const int64_t cper = (int64_t)wpg * 32 * np;                                   // coarse list elements per group
    const size_t o_cs = off;       off = up(off + (size_t)G * cper * 4);
    const size_t o_ci = off;       off = up(off + (size_t)G * cper * 8);
    const int64_t fper = (int64_t)fgrid * 32 * kf;                                 // fine list elements per group
    const size_t o_fs = off;       off = up(off + (size_t)G * fper * 4);
    const size_t o_fi = off;       off = up(off + (size_t)G * fper * 8);
    const int64_t cap = v->total_tiles;
    const size_t o_wt = off;       off = up(off + (size_t)G * cap * 4);
    const size_t o_wr = off;       off = up(off + (size_t)G * cap * 4);
    const size_t o_wm = off;       off = up(off + (size_t)G * cap * 4);
    const size_t o_nw = off;       off = up(off + (size_t)G * 4);
    const size_t o_sc = off;       off = up(off + (size_t)G * 8);
    const size_t o_cds = off;      off = up(off + (i8 ? (size_t)G * 32 * RASS_MAX_K * 4 : 0));   // int8: the merged candidates
    const size_t o_cdr = off;      off = up(off + (i8 ? (size_t)G * 32 * RASS_MAX_K * 8 : 0));
    if (v->batch_bytes < off) {
        if (v->d_batch) HIP_TRY(hipFree(v->d_batch));    // waits for earlier batches that may still read the old block
        v->d_batch = nullptr;
        v->batch_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&v->d_batch), off));
        v->batch_bytes = off;
    }
    unsigned char* ws = v->d_batch;
    float* q_all = reinterpret_cast<float*>(ws + o_q);
    unsigned short* qb_all = reinterpret_cast<unsigned short*>(ws + o_qb);
    float* cs = reinterpret_cast<float*>(ws + o_cs);
    int64_t* ci = reinterpret_cast<int64_t*>(ws + o_ci);
    float* fs = reinterpret_cast<float*>(ws + o_fs);
    int64_t* fi = reinterpret_cast<int64_t*>(ws + o_fi);
    int32_t* wt = reinterpret_cast<int32_t*>(ws + o_wt);
    int32_t* wr = reinterpret_cast<int32_t*>(ws + o_wr);
    uint32_t* wm = reinterpret_cast<uint32_t*>(ws + o_wm);
    int32_t* nw = reinterpret_cast<int32_t*>(ws + o_nw);
    int64_t* sc = reinterpret_cast<int64_t*>(ws + o_sc);

    // (1) every query normalised and zero-padded, the groups' 32-row blocks back to back
    HIP_TRY(rass::launch_normalize_rows_f32(d_queries, v->dim, q_all, stride, nq, v->dim, st, (int64_t)G * 32));
    // (2) coarse: all groups in one launch
    {
        rass::ScanArgs a;
        a.corpus = v->d_centroids;
        a.row_tag = nullptr;
        a.q_padded = q_all;
        a.q_filter = nullptr;
        a.part_scores = cs;
        a.part_ids = ci;
        a.row_stride = stride;
        a.id_base = 0;
        a.n_rows = v->nlist;
        a.nq = 32;
        a.k = np;
        a.wgs_per_group = wpg;
        a.q_group_stride = 32 * stride;
        a.part_group_stride = cper;
        HIP_TRY(rass::launch_scan_topk_f32(a, G * wpg, st));
    }
    // (3) plan: one workgroup per group, the coarse lists merged inside
    HIP_TRY(rass::launch_plan_probe_groups(cs, ci, wpg, np, G, nq, cper, v->nlist, v->d_list_tile0, v->d_list_len, wt, wr, wm,
                                           cap, nw, sc, st, v->tile_rows));
    // (4) the fine scans, one per group, over the group's work list
    const bool need_tags = v->any_tags || d_q_filter != nullptr;
    if (v->dtype == RASS_BF16) HIP_TRY(rass::launch_queries_to_bf16(q_all, qb_all, (int64_t)G * 32 * stride, st));
    if (i8) HIP_TRY(rass::launch_queries_to_i8(q_all, qb_all, G * 32, stride, v->stride_i8, st));
    if (one_fine_launch) {
        rass::ScanArgs a;
        a.corpus = v->d_slab;
        a.row_tag = need_tags ? v->d_tags : nullptr;
        a.q_padded = q_all;
        a.q_filter = d_q_filter;
        a.part_scores = fs;
        a.part_ids = fi;
        a.row_stride = stride;
        a.id_base = 0;
        a.n_rows = (int)v->slab_rows;
        a.nq = 32;
        a.k = k;
        a.work_tile = wt;
        a.work_rows = wr;
        a.work_mask = wm;
        a.n_work = nw;
        a.wgs_per_group = fgrid;
        a.q_group_stride = 32 * stride;
        a.part_group_stride = fper;
        a.work_group_stride = cap;
        a.nq_total = nq;
        const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        HIP_TRY(rass::launch_scan_topk_f32(a, G * fgrid, st));
        if (timed) {
            HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
            eng->ev_used += 1;
        }
    }
    for (int g = 0; g < G && !one_fine_launch; ++g) {
        const int b = std::min(RASS_MAX_QBATCH, nq - g * 32);
        const bool timed = eng->ev_on && (size_t)(2 * eng->ev_used + 1) < eng->ev_pool.size();
        if (timed) HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used], st));
        if (v->dtype == RASS_BF16) {
            rass::ScanBf16Args a;
            a.corpus = v->d_slab_b16;
            a.row_tag = need_tags ? v->d_tags : nullptr;
            a.q_bf16 = qb_all + (int64_t)g * 32 * stride;
            a.q_filter = d_q_filter ? d_q_filter + g * 32 : nullptr;
            a.part_scores = fs + g * fper;
            a.part_ids = fi + g * fper;
            a.row_stride = stride;
            a.n_rows = (int)v->slab_rows;
            a.nq = b;
            a.k = k;
            a.id_base = 0;
            a.work_tile = wt + g * cap;
            a.work_rows = wr + g * cap;
            a.work_mask = wm + g * cap;
            a.n_work = nw + g;
            HIP_TRY(rass::launch_scan_bf16_topk(a, fgrid, st));
        } else if (i8) {
            rass::ScanI8Args a;
            a.corpus = v->d_slab_i8;
            a.row_scale = v->d_slab_scale;
            a.row_tag = need_tags ? v->d_tags : nullptr;
            a.q_i8 = reinterpret_cast<const signed char*>(qb_all) + (int64_t)g * 32 * v->stride_i8;
            a.q_filter = d_q_filter ? d_q_filter + g * 32 : nullptr;
            a.part_scores = fs + g * fper;
            a.part_ids = fi + g * fper;
            a.row_stride = v->stride_i8;
            a.n_rows = (int)v->slab_rows;
            a.nq = b;
            a.k = kf;
            a.work_tile = wt + g * cap;
            a.work_rows = wr + g * cap;
            a.work_mask = wm + g * cap;
            a.n_work = nw + g;
            HIP_TRY(rass::launch_scan_i8_topk(a, fgrid, st));
        } else {
            rass::ScanArgs a;
            a.corpus = v->d_slab;
            a.row_tag = need_tags ? v->d_tags : nullptr;
            a.q_padded = q_all + (int64_t)g * 32 * stride;
            a.q_filter = d_q_filter ? d_q_filter + g * 32 : nullptr;
            a.part_scores = fs + g * fper;
            a.part_ids = fi + g * fper;
            a.row_stride = stride;
            a.id_base = 0;
            a.n_rows = (int)v->slab_rows;
            a.nq = b;
            a.k = k;
            a.work_tile = wt + g * cap;
            a.work_rows = wr + g * cap;
            a.work_mask = wm + g * cap;
            a.n_work = nw + g;
            HIP_TRY(rass::launch_scan_topk_f32(a, fgrid, st));
        }
        if (timed) {
            HIP_TRY(hipEventRecord(eng->ev_pool[2 * eng->ev_used + 1], st));
            eng->ev_used += 1;
        }
    }
    // (5) one grouped merge: slab positions -> source row ids
    rass::MergeGroups mg;
    mg.size = RASS_MAX_QBATCH;
    mg.nq_total = nq;
    mg.lists_are_dense = true;
    mg.score_stride = mg.id_stride = fper;
    mg.out_score_stride = mg.out_id_stride = (int64_t)RASS_MAX_QBATCH * kf;
    if (i8) {   // candidates (slab positions) of every group, then ONE exact re-rank over all queries
        float* cds = reinterpret_cast<float*>(ws + o_cds);
        int64_t* cdr = reinterpret_cast<int64_t*>(ws + o_cdr);
        HIP_TRY(rass::launch_merge_topk(fs, fi, fgrid, nq, kf, cds, cdr, st, nullptr, 0, 0, &mg));
        HIP_TRY(rass::launch_rerank_f32(v->d_slab, stride, q_all, cdr, nq, kf, k, 0, d_out_scores, d_out_ids, st, 0, 0, v->d_ids));
    } else
    HIP_TRY(rass::launch_merge_topk(fs, fi, fgrid, nq, k, d_out_scores, d_out_ids, st, v->d_ids, 0, 0, &mg));
    if (d_scanned_per_group) HIP_TRY(hipMemcpyAsync(d_scanned_per_group, sc, (size_t)G * 8, hipMemcpyDeviceToDevice, st));
    return RASS_OK;
}

int rass_ivf_search_device_batch(rass_ivf_t* v, const float* d_queries, int nq, int k, int nprobe,
                                 const int32_t* d_q_filter, float* d_out_scores, int64_t* d_out_ids,
                                 int64_t* d_scanned_per_group) {
    if (!v || !d_queries || !d_out_scores || !d_out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 1 || nq > 32 * RASS_MAX_QBATCH) return fail(RASS_ERR_INVALID, "nq must be in [1, 1024]");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    if (nprobe < 1) return fail(RASS_ERR_INVALID, "nprobe must be >= 1");
    int rc = set_device(v->eng);
    if (rc != RASS_OK) return rc;
    std::lock_guard<std::mutex> lk(v->eng->mu);
    if (std::min(nprobe, v->nlist) > RASS_MAX_K) {
        // deep probes: the threshold path, group by group
        for (int g = 0; g * RASS_MAX_QBATCH < nq; ++g) {
            const int b = std::min(RASS_MAX_QBATCH, nq - g * RASS_MAX_QBATCH);
            rc = ivf_search_locked(v, d_queries + (int64_t)g * RASS_MAX_QBATCH * v->dim, b, k, nprobe,
                                   d_q_filter ? d_q_filter + g * RASS_MAX_QBATCH : nullptr,
                                   d_out_scores + (int64_t)g * RASS_MAX_QBATCH * k, d_out_ids + (int64_t)g * RASS_MAX_QBATCH * k);
            if (rc != RASS_OK) return rc;
            if (d_scanned_per_group)
                HIP_TRY(hipMemcpyAsync(d_scanned_per_group + g, v->d_scanned, 8, hipMemcpyDeviceToDevice, v->eng->stream));
        }
        return RASS_OK;
    }
    return ivf_search_batch_locked(v, d_queries, nq, k, nprobe, d_q_filter, d_out_scores, d_out_ids, d_scanned_per_group);
}

int rass_ivf_search_device(rass_ivf_t* v, const float* d_queries, int nq, int k, int nprobe,
                           const int32_t* d_q_filter, float* d_out_scores, int64_t* d_out_ids) {
    if (!v) return fail(RASS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(v->eng->mu);
    return ivf_search_locked(v, d_queries, nq, k, nprobe, d_q_filter, d_out_scores, d_out_ids);
}

int rass_ivf_search(rass_ivf_t* v, const float* queries, int nq, int k, int nprobe, const int32_t* q_filter,
                    float* out_scores, int64_t* out_ids, int64_t* scanned_rows) {
    if (!v || !out_scores || !out_ids) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nq < 0 || (nq > 0 && !queries)) return fail(RASS_ERR_INVALID, "bad queries / nq");
    if (k < 1 || k > RASS_MAX_K) return fail(RASS_ERR_INVALID, "k must be in [1, RASS_MAX_K]");
    rass_engine* eng = v->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    int64_t scanned_total = 0;
    SlotGuard guard(eng);
    HostSlot* sl = guard.sl;
    for (int done = 0; done < nq;) {
        const int b = std::min(RASS_MAX_QBATCH, nq - done);
        memcpy(sl->h_q, queries + (int64_t)done * v->dim, (size_t)b * v->dim * 4);
        if (q_filter) memcpy(sl->h_filter, q_filter + done, (size_t)b * 4);
        {
            // engine lock while enqueuing only (shared device staging is safe by stream order)
            std::lock_guard<std::mutex> lk(eng->mu);
            hipStream_t st = eng->stream;
            HIP_TRY(hipMemcpyAsync(eng->d_qraw, sl->h_q, (size_t)b * v->dim * 4, hipMemcpyHostToDevice, st));
            if (q_filter) HIP_TRY(hipMemcpyAsync(eng->d_qfilter, sl->h_filter, (size_t)b * 4, hipMemcpyHostToDevice, st));
            rc = ivf_search_locked(v, eng->d_qraw, b, k, nprobe, q_filter ? eng->d_qfilter : nullptr, eng->d_out_scores,
                                   eng->d_out_ids);
            if (rc != RASS_OK) return rc;
            HIP_TRY(hipMemcpyAsync(sl->h_out_s, eng->d_out_scores, (size_t)b * k * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sl->h_out_i, eng->d_out_ids, (size_t)b * k * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sl->h_scanned, v->d_scanned, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(sl->done, st));
        }
        HIP_TRY(hipEventSynchronize(sl->done));
        memcpy(out_scores + (int64_t)done * k, sl->h_out_s, (size_t)b * k * 4);
        memcpy(out_ids + (int64_t)done * k, sl->h_out_i, (size_t)b * k * 8);
        scanned_total += *sl->h_scanned;
        done += b;
    }
    if (scanned_rows) *scanned_rows = scanned_total;
    return RASS_OK;
}

// ---- K9(i): k-means over rows resident in an index's slab
static int kmeans_range_ok(const rass_index* idx, int64_t first_block, int64_t block_step, int64_t n_blocks) {
    if (first_block < 0 || block_step < 1 || n_blocks < 0) return 0;
    if (n_blocks == 0) return 1;
    const int64_t last = first_block + (n_blocks - 1) * block_step;
    return last * 32 < idx->rows.load();  // the last processed block must hold at least one appended row
}

int rass_kmeans_assign(rass_index_t* idx, int64_t first_block, int64_t block_step, int64_t n_blocks,
                       const float* d_centroids_tile16, int nlist, int32_t* d_assign, float* d_best) {
    if (!idx || !d_centroids_tile16 || !d_assign) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nlist < 1 || nlist > 65536) return fail(RASS_ERR_INVALID, "nlist must be in [1, 65536]");
    if (idx->dtype != RASS_F32) return fail(RASS_ERR_UNSUPPORTED, "k-means needs an fp32 index");
    if (idx->stride > kNarrowStride) return fail(RASS_ERR_UNSUPPORTED, "k-means needs dim <= 1024 (wide rows: flat scan only)");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!kmeans_range_ok(idx, first_block, block_step, n_blocks) || n_blocks > 0x7fffffff)
        return fail(RASS_ERR_INVALID, "block range outside the index");
    rass_engine* eng = idx->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    std::lock_guard<std::mutex> elk(eng->mu);
    rass::AssignArgs a;
    a.rows = idx->d_rows;
    a.centroids = d_centroids_tile16;
    a.assign = d_assign;
    a.best = d_best;
    a.row_stride = idx->stride;
    a.slab_rows = idx->capacity;
    a.first_block = first_block;
    a.block_step = block_step;
    a.n_blocks = (int)n_blocks;
    a.nlist = nlist;
    HIP_TRY(rass::launch_kmeans_assign_f32(a, eng->n_cus, eng->stream));
    return RASS_OK;
}

int rass_kmeans_accumulate(rass_index_t* idx, int64_t first_block, int64_t block_step, int64_t n_blocks,
                           const int32_t* d_assign, float* d_sums, float* d_counts, int nlist) {
    if (!idx || !d_assign || !d_sums || !d_counts) return fail(RASS_ERR_INVALID, "NULL argument");
    if (nlist < 1) return fail(RASS_ERR_INVALID, "nlist < 1");
    if (idx->dtype != RASS_F32) return fail(RASS_ERR_UNSUPPORTED, "k-means needs an fp32 index");
    if (idx->stride > kNarrowStride) return fail(RASS_ERR_UNSUPPORTED, "k-means needs dim <= 1024 (wide rows: flat scan only)");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!kmeans_range_ok(idx, first_block, block_step, n_blocks) || n_blocks > 0x7fffffff)
        return fail(RASS_ERR_INVALID, "block range outside the index");
    rass_engine* eng = idx->eng;
    int rc = set_device(eng);
    if (rc != RASS_OK) return rc;
    std::lock_guard<std::mutex> elk(eng->mu);
    HIP_TRY(rass::launch_kmeans_accumulate(idx->d_rows, idx->stride, first_block, block_step, (int)n_blocks,
                                           idx->rows.load(), d_assign, d_sums, d_counts, idx->dim, nlist, eng->stream));
    return RASS_OK;
}

// ---- §8f-4: peer-store exchange (no collective on the search path)
int rass_peer_buffer_create(int device, size_t bytes, void** d_ptr, unsigned char* handle64) {
    if (!d_ptr || !handle64 || bytes == 0) return fail(RASS_ERR_INVALID, "bad argument");
    *d_ptr = nullptr;
    HIP_TRY(hipSetDevice(device));
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // the flags must BE zero before any peer (or stream) touches them
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(RASS_ERR_HIP, std::string("peer buffer: ") + hipGetErrorString(e) +
                                      " (multi-process GPU memory sharing needs HSA_ENABLE_IPC_MODE_LEGACY=0 here)");
    }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI carries the IPC handle as 64 bytes");
    memcpy(handle64, &h, 64);
    *d_ptr = p;
    return RASS_OK;
}

int rass_peer_buffer_open(int device, const unsigned char* handle64, void** d_ptr) {
    if (!d_ptr || !handle64) return fail(RASS_ERR_INVALID, "NULL argument");
    *d_ptr = nullptr;
    HIP_TRY(hipSetDevice(device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    HIP_TRY(hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return RASS_OK;
}

int rass_peer_buffer_close(void* d_ptr, int opened_from_handle) {
    if (!d_ptr) return RASS_OK;
    if (opened_from_handle)
        HIP_TRY(hipIpcCloseMemHandle(d_ptr));
    else
        HIP_TRY(hipFree(d_ptr));
    return RASS_OK;
}

int rass_peer_post(const void* d_record, size_t bytes, void* d_remote_slot, void* d_remote_flag, uint64_t seq,
                   void* stream) {
    if (!d_record || !d_remote_slot || !d_remote_flag) return fail(RASS_ERR_INVALID, "NULL argument");
    HIP_TRY(rass::launch_peer_post(d_record, bytes, d_remote_slot, d_remote_flag, seq, reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

int rass_peer_wait(const void* d_flags, int n, int flag_stride_bytes, uint64_t seq, int* d_status, int64_t max_spins,
                   void* stream) {
    if (!d_flags || !d_status || max_spins < 1) return fail(RASS_ERR_INVALID, "bad argument");
    HIP_TRY(rass::launch_peer_wait(d_flags, n, flag_stride_bytes, seq, d_status, max_spins,
                                   reinterpret_cast<hipStream_t>(stream)));
    return RASS_OK;
}

const char* rass_scan_kernel_name(int dim, int nq) {
    static thread_local char buf[64];
    const int64_t stride = pad_stride(dim);
    if (dim < 1 || !rass::scan_supported_stride(stride) || stride > kMaxStride || nq < 1 || nq > RASS_MAX_QBATCH)
        return "";
    if (stride > kNarrowStride) {
        const int ch = (int)(stride / 128);
        snprintf(buf, sizeof(buf), "scan_topk_f32_wide_kernel<%d, %d, false>", ch == 16 ? 4 : ch / 2, ch == 16 ? 4 : 2);
        return buf;
    }
    snprintf(buf, sizeof(buf), "scan_topk_f32_kernel<%d, %d, 0, false>", (int)(stride / 128), nq <= 16 ? 1 : 2);
    return buf;
}

}  // extern "C"
