// encoder_api.hip — C ABI of the sentence encoder (include/rass_engine.h, "encoder" section):
// a BERT-class post-LN transformer (mxbai-embed-large class: 24 x 1024 x 16 heads x 4096,
// reference OLLAMA_EMBED_MODEL, app/main.py:67) whose forward pass is the K4-K8 kernels.
// Replaces the arithmetic behind ollama_embed_text (app/main.py:225-237): one batched,
// varlen-packed forward instead of one HTTP request per text.
//
// Weights arrive by their Hugging Face BERT names as fp32 host arrays (the Python loader
// reads a local model.safetensors); matrices are stored bf16 in HBM ([out][in], nn.Linear
// layout; q/k/v fused to one [3H][H]), biases and LayerNorm parameters fp32.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rass_engine.h"
#include "encoder_kernels.h"

extern "C" void rassint_set_last_error(const char* msg);  // api.hip (internal, not part of the ABI)

namespace {

int efail(int code, const std::string& msg) {
    rassint_set_last_error(msg.c_str());
    // A failed HIP call (e.g. an out-of-memory hipMalloc) stays behind as the runtime's "last error" and the
    // NEXT kernel launch's hipGetLastError() would report it as its own: every failure path ends here, clear it.
    (void)hipGetLastError();
    return code;
}

#define EHIP_TRY(expr)                                                                                  \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            char _b[512];                                                                               \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                     __LINE__);                                                                         \
            return efail(_e == hipErrorOutOfMemory ? RASS_ERR_OOM : RASS_ERR_HIP, _b);                  \
        }                                                                                               \
    } while (0)

struct Layer {
    void* w_qkv = nullptr;   // bf16 [3H][H]
    float* b_qkv = nullptr;  // [3H]
    void* w_o = nullptr;     // bf16 [H][H]
    float* b_o = nullptr;
    float *ln1_g = nullptr, *ln1_b = nullptr;
    void* w_up = nullptr;    // bf16 [I][H]
    float* b_up = nullptr;
    void* w_down = nullptr;  // bf16 [H][I]
    float* b_down = nullptr;
    float *ln2_g = nullptr, *ln2_b = nullptr;
    // LayerNorm folded into its consumer GEMMs (big batches; encoder_gemm.hip, LnFold): W' = W diag(gamma) in bf16, the
    // fp32 column sums of W' and bias' = b + beta W^T.  QKV folds the PREVIOUS layer's output LayerNorm (layer 0: identity,
    // its input arrives normalised from the embedding LayerNorm), FFN-up this layer's attention-output LayerNorm.
    void *w_qkv_f = nullptr, *w_up_f = nullptr;
    float *cs_qkv = nullptr, *bf_qkv = nullptr, *cs_up = nullptr, *bf_up = nullptr;
};

}  // namespace

struct rass_encoder {
    int device = 0;
    rass_encoder_config cfg;
    hipStream_t own_stream = nullptr;
    std::mutex mu;
    std::vector<void*> allocs;
    void *word = nullptr, *pos = nullptr, *type0 = nullptr;  // bf16
    float *emb_g = nullptr, *emb_b = nullptr;
    std::vector<Layer> layers;
    std::map<std::string, bool> seen;
    bool finalized = false;
    // workspace (grown on demand)
    int cap_tokens = 0, cap_seqs = 0;
    void *x = nullptr, *qkv = nullptr, *ctx = nullptr, *y = nullptr, *h = nullptr;  // bf16 activations
    int32_t *d_ids = nullptr, *d_cu = nullptr;
    float* d_out = nullptr;
    float* d_stage = nullptr;  // fp32 staging for weight upload
    size_t stage_elems = 0;
    float* d_splitk = nullptr;  // fp32 partial tiles of the split-K GEMMs of small batches (query-time embedding)
    size_t splitk_bytes = 0;
    // LN fold (big batches): folded weights are prepared on the first forward that takes the path (any later
    // rass_encoder_set_weight invalidates them); per-row statistics of the two raw residual streams
    bool fold_ready = false;
    float *ones_h = nullptr, *zeros_h = nullptr;        // [H] the identity LayerNorm in front of layer 0
    float *ln_stats = nullptr;                          // [cap_tokens][H / 128][2] partial (sum, sum of squares)
    float *mr_a = nullptr, *mr_b = nullptr, *mr_id = nullptr;   // [cap_tokens][2] (mean, rstd); mr_id = (0, 1)
    // The activation workspace is ONE set per encoder: a forward enqueued on stream A must finish
    // before a forward on stream B (or a reallocation) touches it.  `done` is recorded at the end of
    // every forward on `last_stream`.
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;
    bool done_recorded = false;
    int64_t n_forwards = 0, n_seqs = 0, n_tokens = 0;  // rass_encoder_stats (under mu)
};

namespace {

int dev_alloc(rass_encoder* e, void** p, size_t bytes) {
    EHIP_TRY(hipMalloc(p, bytes));
    e->allocs.push_back(*p);
    return RASS_OK;
}

// upload fp32 host -> device, optionally converting to bf16 at dst (element offset dst_off)
int upload(rass_encoder* e, const float* host, int64_t n, void* dst, int64_t dst_off, bool to_bf16) {
    hipStream_t st = e->own_stream;
    if (!to_bf16) {
        EHIP_TRY(hipMemcpyAsync(static_cast<float*>(dst) + dst_off, host, (size_t)n * 4, hipMemcpyHostToDevice, st));
        EHIP_TRY(hipStreamSynchronize(st));
        return RASS_OK;
    }
    const int64_t chunk = (int64_t)e->stage_elems;
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t m = std::min(chunk, n - done);
        EHIP_TRY(hipMemcpyAsync(e->d_stage, host + done, (size_t)m * 4, hipMemcpyHostToDevice, st));
        EHIP_TRY(rass::launch_f32_to_bf16(e->d_stage, static_cast<unsigned short*>(dst) + dst_off + done, m, st));
        EHIP_TRY(hipStreamSynchronize(st));
    }
    return RASS_OK;
}

int ensure_workspace(rass_encoder* e, int tokens_pad, int nseq) {
    const int H = e->cfg.hidden, I = e->cfg.intermediate;
    if ((tokens_pad > e->cap_tokens || nseq > e->cap_seqs) && e->done_recorded)
        EHIP_TRY(hipEventSynchronize(e->done));  // a forward still in flight reads what is freed below
    if (tokens_pad > e->cap_tokens) {
        // the capacity is withdrawn BEFORE anything is freed and published only after every
        // allocation succeeded: a failed grow leaves cap_tokens = 0 and the next call re-allocates
        e->cap_tokens = 0;
        for (void** p : {&e->x, &e->qkv, &e->ctx, &e->y, &e->h, (void**)&e->ln_stats, (void**)&e->mr_a, (void**)&e->mr_b,
                         (void**)&e->mr_id})
            if (*p) {
                (void)hipFree(*p);
                *p = nullptr;
            }
        if (e->d_ids) (void)hipFree(e->d_ids);
        e->d_ids = nullptr;
        const size_t T = (size_t)tokens_pad;
        EHIP_TRY(hipMalloc(&e->x, T * H * 2));
        EHIP_TRY(hipMalloc(&e->qkv, T * 3 * H * 2));
        EHIP_TRY(hipMalloc(&e->ctx, T * H * 2));
        EHIP_TRY(hipMalloc(&e->y, T * H * 2));
        EHIP_TRY(hipMalloc(&e->h, T * I * 2));
        EHIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_ids), T * 4));
        if (H % 128 == 0 && T >= 1024) {   // LN-fold statistics (only batches of >= 12 288 tokens take that path)
            EHIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->ln_stats), T * (H / 128) * 2 * 4));
            EHIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->mr_a), T * 2 * 4));
            EHIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->mr_b), T * 2 * 4));
            EHIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->mr_id), T * 2 * 4));
            std::vector<float> id(T * 2);
            for (size_t i = 0; i < T; ++i) id[2 * i] = 0.f, id[2 * i + 1] = 1.f;
            EHIP_TRY(hipMemcpy(e->mr_id, id.data(), T * 2 * 4, hipMemcpyHostToDevice));
        }
        // padding rows are read as GEMM operands: keep them finite
        EHIP_TRY(hipMemset(e->x, 0, T * H * 2));
        EHIP_TRY(hipMemset(e->qkv, 0, T * 3 * H * 2));
        EHIP_TRY(hipMemset(e->ctx, 0, T * H * 2));
        EHIP_TRY(hipMemset(e->y, 0, T * H * 2));
        EHIP_TRY(hipMemset(e->h, 0, T * I * 2));
        // hipMemset runs on the NULL stream and may return before it has executed; the forward runs on a NON-BLOCKING
        // stream, which the NULL stream does not order: without this wait a late memset could zero activations the first
        // forward after a (re)allocation had already written (seen once as a wrong first embedding of a new encoder)
        EHIP_TRY(hipDeviceSynchronize());
        e->cap_tokens = tokens_pad;
    }
    if (nseq > e->cap_seqs) {
        e->cap_seqs = 0;
        if (e->d_cu) (void)hipFree(e->d_cu);
        if (e->d_out) (void)hipFree(e->d_out);
        e->d_cu = nullptr;
        e->d_out = nullptr;
        EHIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_cu), ((size_t)nseq + 1) * 4));
        EHIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_out), (size_t)nseq * H * 4));
        e->cap_seqs = nseq;
    }
    return RASS_OK;
}

// W' / colsum / bias' of every layer's QKV and FFN-up (LnFold); on the encoder's own stream, synchronously, once
int prepare_fold(rass_encoder* e) {
    const rass_encoder_config& c = e->cfg;
    const int H = c.hidden, I = c.intermediate;
    hipStream_t st = e->own_stream;
    if (!e->ones_h) {
        int rc;
        if ((rc = dev_alloc(e, reinterpret_cast<void**>(&e->ones_h), (size_t)H * 4)) != RASS_OK) return rc;
        if ((rc = dev_alloc(e, reinterpret_cast<void**>(&e->zeros_h), (size_t)H * 4)) != RASS_OK) return rc;
        std::vector<float> one((size_t)H, 1.f), zero((size_t)H, 0.f);
        EHIP_TRY(hipMemcpy(e->ones_h, one.data(), (size_t)H * 4, hipMemcpyHostToDevice));
        EHIP_TRY(hipMemcpy(e->zeros_h, zero.data(), (size_t)H * 4, hipMemcpyHostToDevice));
    }
    for (int l = 0; l < c.layers; ++l) {
        Layer& L = e->layers[(size_t)l];
        if (!L.w_qkv_f) {
            int rc;
            if ((rc = dev_alloc(e, &L.w_qkv_f, (size_t)3 * H * H * 2)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.cs_qkv), (size_t)3 * H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.bf_qkv), (size_t)3 * H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, &L.w_up_f, (size_t)I * H * 2)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.cs_up), (size_t)I * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.bf_up), (size_t)I * 4)) != RASS_OK) return rc;
        }
        const float* g_prev = l == 0 ? e->ones_h : e->layers[(size_t)l - 1].ln2_g;
        const float* b_prev = l == 0 ? e->zeros_h : e->layers[(size_t)l - 1].ln2_b;
        EHIP_TRY(rass::launch_fold_gamma(L.w_qkv, g_prev, b_prev, L.b_qkv, 3 * H, H, L.w_qkv_f, L.cs_qkv, L.bf_qkv, st));
        EHIP_TRY(rass::launch_fold_gamma(L.w_up, L.ln1_g, L.ln1_b, L.b_up, I, H, L.w_up_f, L.cs_up, L.bf_up, st));
    }
    EHIP_TRY(hipStreamSynchronize(st));
    e->fold_ready = true;
    return RASS_OK;
}

// The big-batch forward with both LayerNorms of every layer folded into the GEMMs around them (encoder_gemm.hip, LnFold):
// e->x holds the RAW layer output r2 (layer 0: the normalised embeddings, statistics = identity), e->y the raw r1.
int forward_fold(rass_encoder* e, const int32_t* d_cu, int nseq, int total, int Tp, int max_seqlen, float* d_out, hipStream_t st) {
    const rass_encoder_config& c = e->cfg;
    const int H = c.hidden, I = c.intermediate;
    const float eps = c.layer_norm_eps;
    const float* mr_prev = e->mr_id;
    for (int l = 0; l < c.layers; ++l) {
        const Layer& L = e->layers[(size_t)l];
        const float* g_prev = l == 0 ? e->ones_h : e->layers[(size_t)l - 1].ln2_g;
        const float* b_prev = l == 0 ? e->zeros_h : e->layers[(size_t)l - 1].ln2_b;
        // qkv = LN_prev(x) Wqkv^T + b, on the raw x
        EHIP_TRY(rass::launch_gemm_bf16_fold(e->x, L.w_qkv_f, L.bf_qkv, nullptr, e->qkv, total, Tp, 3 * H, H, 4, mr_prev, nullptr,
                                             nullptr, nullptr, L.cs_qkv, st));
        EHIP_TRY(rass::launch_attention(e->qkv, d_cu, nseq, total, max_seqlen, H, c.heads, e->ctx, st));
        // r1 = ctx Wo^T + b + LN_prev(x)  (raw, + row statistics)
        EHIP_TRY(rass::launch_gemm_bf16_fold(e->ctx, L.w_o, L.b_o, e->x, e->y, total, Tp, H, H, 3, mr_prev, g_prev, b_prev,
                                             e->ln_stats, nullptr, st));
        EHIP_TRY(rass::launch_ln_stats_finalize(e->ln_stats, total, H, eps, e->mr_b, st));
        // h = gelu(LN1(r1) Wup^T + b), on the raw r1
        EHIP_TRY(rass::launch_gemm_bf16_fold(e->y, L.w_up_f, L.bf_up, nullptr, e->h, total, Tp, I, H, 5, e->mr_b, nullptr, nullptr,
                                             nullptr, L.cs_up, st));
        // r2 = h Wdown^T + b + LN1(r1)  (raw, + row statistics) -> e->x
        EHIP_TRY(rass::launch_gemm_bf16_fold(e->h, L.w_down, L.b_down, e->y, e->x, total, Tp, H, I, 3, e->mr_b, L.ln1_g, L.ln1_b,
                                             e->ln_stats, nullptr, st));
        EHIP_TRY(rass::launch_ln_stats_finalize(e->ln_stats, total, H, eps, e->mr_a, st));
        mr_prev = e->mr_a;
    }
    // the last layer's output LayerNorm, once, in front of the pooling
    const Layer& last = e->layers[(size_t)c.layers - 1];
    EHIP_TRY(rass::launch_layernorm(e->x, last.ln2_g, last.ln2_b, eps, total, H, e->y, st));
    EHIP_TRY(rass::launch_pool(e->y, d_cu, nseq, H, c.pooling == 1 ? 1 : 0, c.normalize ? 1 : 0, d_out, st));
    return RASS_OK;
}

int forward(rass_encoder* e, const int32_t* d_ids, const int32_t* d_cu, int nseq, int total, int max_seqlen,
            float* d_out, hipStream_t st) {
    rass::rass_env_new_scope();   // the kernels' A/B switches are read once per forward
    const rass_encoder_config& c = e->cfg;
    const int H = c.hidden, I = c.intermediate;
    const int Tp = (total + 255) / 256 * 256;  // whole 256-token GEMM tiles
    if (!e->x || !e->qkv || !e->ctx || !e->y || !e->h || Tp > e->cap_tokens)
        return efail(RASS_ERR_INVALID, "encoder workspace is not allocated for this batch");
    // one workspace per encoder: order this forward after the previous one when the stream differs
    if (e->done_recorded && e->last_stream != st) EHIP_TRY(hipStreamWaitEvent(st, e->done, 0));
    const bool fold = e->ln_stats != nullptr && rass::gemm_bf16_fold_ok(total, Tp, H, I);
    if (fold && !e->fold_ready) {
        const int rc = prepare_fold(e);
        if (rc != RASS_OK) return rc;
    }
    EHIP_TRY(rass::launch_embed_layernorm(d_ids, d_cu, nseq, total, e->word, e->pos, e->type0, e->emb_g, e->emb_b,
                                          c.layer_norm_eps, H, c.vocab_size, c.max_positions, e->x, st));
    if (fold) {
        const int rc = forward_fold(e, d_cu, nseq, total, Tp, max_seqlen, d_out, st);
        if (rc != RASS_OK) return rc;
    }
    for (int l = 0; l < c.layers && !fold; ++l) {
        const Layer& L = e->layers[(size_t)l];
        EHIP_TRY(rass::launch_gemm_bf16(e->x, L.w_qkv, L.b_qkv, nullptr, e->qkv, total, Tp, 3 * H, H, 0, st, e->d_splitk, e->splitk_bytes));
        const bool query_rows = rass::gemm_bf16_ln_input_ok(total, I, H);
        const bool attn_fused = query_rows && e->d_splitk != nullptr && rass::attn_out_fused_ok(total, nseq, H, c.heads, H) &&
                                rass::attn_out_fused_pays(total, nseq);
        if (!attn_fused) EHIP_TRY(rass::launch_attention(e->qkv, d_cu, nseq, total, max_seqlen, H, c.heads, e->ctx, st));
        // attn-out + residual + LayerNorm; the residual (e->x) is also the output: every row is read before it is written
        // (a wave owns a row), and the big-batch form goes through e->y
        if (query_rows) {
            // a query: y = ctx W_o^T + b_o + x (the attention recomputed inside that GEMM's workgroups: one launch less), then
            // FFN-up normalises y itself (and stores x = LayerNorm(y) once)
            if (attn_fused)
                EHIP_TRY(rass::launch_attn_out_fused(e->qkv, d_cu, nseq, total, H, c.heads, L.w_o, L.b_o, e->x, e->y, H, st));
            else
                EHIP_TRY(rass::launch_gemm_bf16(e->ctx, L.w_o, L.b_o, e->x, e->y, total, Tp, H, H, 1, st, e->d_splitk, e->splitk_bytes));
            EHIP_TRY(rass::launch_gemm_bf16_ln_input(e->y, L.ln1_g, L.ln1_b, c.layer_norm_eps, e->x, L.w_up, L.b_up, e->h,
                                                     total, I, H, 2, st));
        } else {
            EHIP_TRY(rass::launch_gemm_bf16_residual_layernorm(e->ctx, L.w_o, L.b_o, e->x, e->y, L.ln1_g, L.ln1_b,
                                                               c.layer_norm_eps, e->x, total, Tp, H, H, st, e->d_splitk,
                                                               e->splitk_bytes));
            EHIP_TRY(rass::launch_gemm_bf16(e->x, L.w_up, L.b_up, nullptr, e->h, total, Tp, I, H, 2, st, e->d_splitk, e->splitk_bytes));
        }
        EHIP_TRY(rass::launch_gemm_bf16_residual_layernorm(e->h, L.w_down, L.b_down, e->x, e->y, L.ln2_g, L.ln2_b,
                                                           c.layer_norm_eps, e->x, total, Tp, H, I, st, e->d_splitk,
                                                           e->splitk_bytes));
    }
    if (!fold) EHIP_TRY(rass::launch_pool(e->x, d_cu, nseq, H, c.pooling == 1 ? 1 : 0, c.normalize ? 1 : 0, d_out, st));
    EHIP_TRY(hipEventRecord(e->done, st));
    e->last_stream = st;
    e->done_recorded = true;
    e->n_forwards += 1;
    e->n_seqs += nseq;
    e->n_tokens += total;
    return RASS_OK;
}

}  // namespace

extern "C" {

int rass_encoder_create(int device, const rass_encoder_config* cfg, rass_encoder_t** out) {
    if (!cfg || !out) return efail(RASS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (cfg->hidden < 64 || cfg->hidden % 128 != 0 || cfg->hidden > 2048 || cfg->heads < 1 ||
        cfg->hidden != cfg->heads * 64)
        return efail(RASS_ERR_UNSUPPORTED, "hidden must be heads*64, a multiple of 128, <= 2048");
    if (cfg->intermediate < 128 || cfg->intermediate % 128 != 0)
        return efail(RASS_ERR_UNSUPPORTED, "intermediate must be a multiple of 128");
    if (cfg->layers < 1 || cfg->vocab_size < 1 || cfg->max_positions < 1 || cfg->max_positions > 512)
        return efail(RASS_ERR_UNSUPPORTED, "layers/vocab/max_positions out of range (max_positions <= 512)");
    int n = 0;
    EHIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return efail(RASS_ERR_INVALID, "no such HIP device");
    rass_encoder* e = new (std::nothrow) rass_encoder();
    if (!e) return efail(RASS_ERR_OOM, "host allocation failed");
    e->device = device;
    e->cfg = *cfg;
    auto init = [&]() -> int {
        EHIP_TRY(hipSetDevice(device));
        EHIP_TRY(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
        EHIP_TRY(hipEventCreateWithFlags(&e->done, hipEventDisableTiming));
        const size_t H = (size_t)cfg->hidden, I = (size_t)cfg->intermediate;
        e->stage_elems = std::max<size_t>(I * H, 1 << 20);
        int rc;
        if ((rc = dev_alloc(e, reinterpret_cast<void**>(&e->d_stage), e->stage_elems * 4)) != RASS_OK) return rc;
        // split-K scratch: 16 slices x 256 rows x the widest GEMM output that is split that far (H), or fewer
        // slices of wider outputs: S * M_pad * N floats, the launcher picks S to fit
        // (the mid-size kernel splits K for batches up to ~2 000 tokens: 4 slices x 1 024 rows x the widest output)
        e->splitk_bytes = std::max((size_t)16 * 256 * std::max(H, I / 4), (size_t)4 * 1024 * std::max(3 * H, I)) * sizeof(float);
        if ((rc = dev_alloc(e, reinterpret_cast<void**>(&e->d_splitk), e->splitk_bytes)) != RASS_OK) return rc;
        if ((rc = dev_alloc(e, &e->word, (size_t)cfg->vocab_size * H * 2)) != RASS_OK) return rc;
        if ((rc = dev_alloc(e, &e->pos, (size_t)cfg->max_positions * H * 2)) != RASS_OK) return rc;
        if ((rc = dev_alloc(e, &e->type0, H * 2)) != RASS_OK) return rc;
        if ((rc = dev_alloc(e, reinterpret_cast<void**>(&e->emb_g), H * 4)) != RASS_OK) return rc;
        if ((rc = dev_alloc(e, reinterpret_cast<void**>(&e->emb_b), H * 4)) != RASS_OK) return rc;
        e->layers.resize((size_t)cfg->layers);
        for (Layer& L : e->layers) {
            if ((rc = dev_alloc(e, &L.w_qkv, 3 * H * H * 2)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.b_qkv), 3 * H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, &L.w_o, H * H * 2)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.b_o), H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.ln1_g), H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.ln1_b), H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, &L.w_up, I * H * 2)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.b_up), I * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, &L.w_down, H * I * 2)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.b_down), H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.ln2_g), H * 4)) != RASS_OK) return rc;
            if ((rc = dev_alloc(e, reinterpret_cast<void**>(&L.ln2_b), H * 4)) != RASS_OK) return rc;
        }
        return RASS_OK;
    };
    const int rc = init();
    if (rc != RASS_OK) {  // release the stream, the event and every allocation made so far
        const std::string why = rass_last_error();
        rass_encoder_destroy(e);
        return efail(rc, why);
    }
    *out = e;
    return RASS_OK;
}

void rass_encoder_destroy(rass_encoder_t* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    for (void* p : {e->x, e->qkv, e->ctx, e->y, e->h, (void*)e->d_ids, (void*)e->d_cu, (void*)e->d_out, (void*)e->ln_stats,
                    (void*)e->mr_a, (void*)e->mr_b, (void*)e->mr_id})
        if (p) (void)hipFree(p);
    if (e->done) (void)hipEventDestroy(e->done);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int rass_encoder_set_weight(rass_encoder_t* e, const char* name, const float* data, int64_t numel) {
    if (!e || !name || !data) return efail(RASS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(e->mu);
    EHIP_TRY(hipSetDevice(e->device));
    const int64_t H = e->cfg.hidden, I = e->cfg.intermediate;
    const std::string n(name);
    auto want = [&](int64_t expect) -> int {
        if (numel != expect) {
            char b[256];
            snprintf(b, sizeof(b), "%s: expected %lld elements, got %lld", name, (long long)expect, (long long)numel);
            return efail(RASS_ERR_INVALID, b);
        }
        return RASS_OK;
    };
    int rc = RASS_ERR_NOT_FOUND;
    if (n == "embeddings.word_embeddings.weight") {
        if ((rc = want((int64_t)e->cfg.vocab_size * H)) == RASS_OK) rc = upload(e, data, numel, e->word, 0, true);
    } else if (n == "embeddings.position_embeddings.weight") {
        // a checkpoint may carry more positions than the engine serves: take the first max_positions rows
        if (numel >= (int64_t)e->cfg.max_positions * H && numel % H == 0)
            rc = upload(e, data, (int64_t)e->cfg.max_positions * H, e->pos, 0, true);
        else
            rc = want((int64_t)e->cfg.max_positions * H);
    } else if (n == "embeddings.token_type_embeddings.weight") {
        if (numel >= H && numel % H == 0) rc = upload(e, data, H, e->type0, 0, true);  // row 0: single-segment input
        else rc = want(H);
    } else if (n == "embeddings.LayerNorm.weight") {
        if ((rc = want(H)) == RASS_OK) rc = upload(e, data, H, e->emb_g, 0, false);
    } else if (n == "embeddings.LayerNorm.bias") {
        if ((rc = want(H)) == RASS_OK) rc = upload(e, data, H, e->emb_b, 0, false);
    } else if (n.rfind("encoder.layer.", 0) == 0) {
        const size_t p0 = strlen("encoder.layer.");
        const size_t dot = n.find('.', p0);
        if (dot == std::string::npos) return efail(RASS_ERR_NOT_FOUND, std::string("unknown weight ") + name);
        const int l = atoi(n.substr(p0, dot - p0).c_str());
        if (l < 0 || l >= e->cfg.layers) return efail(RASS_ERR_NOT_FOUND, std::string("layer out of range: ") + name);
        Layer& L = e->layers[(size_t)l];
        const std::string s = n.substr(dot + 1);
        if (s == "attention.self.query.weight") { if ((rc = want(H * H)) == RASS_OK) rc = upload(e, data, numel, L.w_qkv, 0, true); }
        else if (s == "attention.self.key.weight") { if ((rc = want(H * H)) == RASS_OK) rc = upload(e, data, numel, L.w_qkv, H * H, true); }
        else if (s == "attention.self.value.weight") { if ((rc = want(H * H)) == RASS_OK) rc = upload(e, data, numel, L.w_qkv, 2 * H * H, true); }
        else if (s == "attention.self.query.bias") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.b_qkv, 0, false); }
        else if (s == "attention.self.key.bias") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.b_qkv, H, false); }
        else if (s == "attention.self.value.bias") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.b_qkv, 2 * H, false); }
        else if (s == "attention.output.dense.weight") { if ((rc = want(H * H)) == RASS_OK) rc = upload(e, data, numel, L.w_o, 0, true); }
        else if (s == "attention.output.dense.bias") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.b_o, 0, false); }
        else if (s == "attention.output.LayerNorm.weight") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.ln1_g, 0, false); }
        else if (s == "attention.output.LayerNorm.bias") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.ln1_b, 0, false); }
        else if (s == "intermediate.dense.weight") { if ((rc = want(I * H)) == RASS_OK) rc = upload(e, data, numel, L.w_up, 0, true); }
        else if (s == "intermediate.dense.bias") { if ((rc = want(I)) == RASS_OK) rc = upload(e, data, numel, L.b_up, 0, false); }
        else if (s == "output.dense.weight") { if ((rc = want(H * I)) == RASS_OK) rc = upload(e, data, numel, L.w_down, 0, true); }
        else if (s == "output.dense.bias") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.b_down, 0, false); }
        else if (s == "output.LayerNorm.weight") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.ln2_g, 0, false); }
        else if (s == "output.LayerNorm.bias") { if ((rc = want(H)) == RASS_OK) rc = upload(e, data, numel, L.ln2_b, 0, false); }
    }
    if (rc == RASS_ERR_NOT_FOUND) return efail(RASS_ERR_NOT_FOUND, std::string("unknown weight ") + name);
    if (rc == RASS_OK) {
        e->seen[n] = true;
        e->fold_ready = false;   // the folded copies (LnFold) are rebuilt on the next big-batch forward
    }
    return rc;
}

int rass_encoder_finalize(rass_encoder_t* e) {
    if (!e) return efail(RASS_ERR_INVALID, "encoder is NULL");
    std::lock_guard<std::mutex> lk(e->mu);
    std::vector<std::string> need = {"embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
                                     "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight",
                                     "embeddings.LayerNorm.bias"};
    const char* per_layer[] = {"attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
                               "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias",
                               "attention.output.dense.weight", "attention.output.dense.bias",
                               "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
                               "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight",
                               "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"};
    for (int l = 0; l < e->cfg.layers; ++l)
        for (const char* s : per_layer) need.push_back("encoder.layer." + std::to_string(l) + "." + s);
    for (const std::string& n : need)
        if (!e->seen.count(n)) return efail(RASS_ERR_NOT_FOUND, "missing weight: " + n);
    e->finalized = true;
    return RASS_OK;
}

int rass_encode_device(rass_encoder_t* e, const int32_t* d_token_ids, const int32_t* d_cu_seqlens, int nseq,
                       int total_tokens, int max_seqlen, float* d_out, void* stream) {
    if (!e || !d_token_ids || !d_cu_seqlens || !d_out) return efail(RASS_ERR_INVALID, "NULL argument");
    if (!e->finalized) return efail(RASS_ERR_INVALID, "encoder weights not finalized");
    if (nseq < 1 || total_tokens < 1 || max_seqlen < 1 || max_seqlen > e->cfg.max_positions)
        return efail(RASS_ERR_INVALID, "bad nseq / total_tokens / max_seqlen");
    std::lock_guard<std::mutex> lk(e->mu);
    EHIP_TRY(hipSetDevice(e->device));
    int rc = ensure_workspace(e, (total_tokens + 255) / 256 * 256, nseq);
    if (rc != RASS_OK) return rc;
    return forward(e, d_token_ids, d_cu_seqlens, nseq, total_tokens, max_seqlen, d_out,
                   stream ? reinterpret_cast<hipStream_t>(stream) : e->own_stream);
}

int rass_encode(rass_encoder_t* e, const int32_t* token_ids, const int32_t* cu_seqlens, int nseq, float* out) {
    if (!e || !token_ids || !cu_seqlens || !out) return efail(RASS_ERR_INVALID, "NULL argument");
    if (!e->finalized) return efail(RASS_ERR_INVALID, "encoder weights not finalized");
    if (nseq < 1) return efail(RASS_ERR_INVALID, "nseq < 1");
    if (cu_seqlens[0] != 0) return efail(RASS_ERR_INVALID, "cu_seqlens[0] must be 0");
    int max_len = 0;
    for (int s = 0; s < nseq; ++s) {
        const int len = cu_seqlens[s + 1] - cu_seqlens[s];
        if (len < 1) return efail(RASS_ERR_INVALID, "empty sequence (blank texts are handled by the caller)");
        max_len = std::max(max_len, len);
    }
    if (max_len > e->cfg.max_positions) return efail(RASS_ERR_INVALID, "sequence longer than max_positions");
    const int total = cu_seqlens[nseq];
    std::lock_guard<std::mutex> lk(e->mu);
    EHIP_TRY(hipSetDevice(e->device));
    int rc = ensure_workspace(e, (total + 255) / 256 * 256, nseq);
    if (rc != RASS_OK) return rc;
    hipStream_t st = e->own_stream;
    EHIP_TRY(hipMemcpyAsync(e->d_ids, token_ids, (size_t)total * 4, hipMemcpyHostToDevice, st));
    EHIP_TRY(hipMemcpyAsync(e->d_cu, cu_seqlens, ((size_t)nseq + 1) * 4, hipMemcpyHostToDevice, st));
    rc = forward(e, e->d_ids, e->d_cu, nseq, total, max_len, e->d_out, st);
    if (rc != RASS_OK) return rc;
    EHIP_TRY(hipMemcpyAsync(out, e->d_out, (size_t)nseq * e->cfg.hidden * 4, hipMemcpyDeviceToHost, st));
    EHIP_TRY(hipStreamSynchronize(st));
    return RASS_OK;
}

void* rass_encoder_get_stream(rass_encoder_t* e) { return e ? reinterpret_cast<void*>(e->own_stream) : nullptr; }

int rass_encoder_stats(rass_encoder_t* e, int64_t out[3]) {
    if (!e || !out) return efail(RASS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(e->mu);
    out[0] = e->n_forwards;
    out[1] = e->n_seqs;
    out[2] = e->n_tokens;
    return RASS_OK;
}

int rass_encoder_hidden(const rass_encoder_t* e) { return e ? e->cfg.hidden : efail(RASS_ERR_INVALID, "encoder is NULL"); }

/* Stand-alone launcher of the encoder GEMM (tests, micro-benchmarks). */
int rass_gemm_bf16(const void* d_x, const void* d_w, const float* d_bias, const void* d_residual, void* d_y, int m,
                   int m_pad, int n, int k, int epilogue, void* stream) {
    rass::rass_env_new_scope();
    if (!d_x || !d_w || !d_bias || !d_y) return efail(RASS_ERR_INVALID, "NULL argument");
    hipError_t err = rass::launch_gemm_bf16(d_x, d_w, d_bias, d_residual, d_y, m, m_pad, n, k, epilogue,
                                            reinterpret_cast<hipStream_t>(stream));
    if (err != hipSuccess) return efail(RASS_ERR_INVALID, std::string("gemm launch: ") + hipGetErrorString(err));
    return RASS_OK;
}

/* Same with a caller-owned fp32 scratch: lets a GEMM over few rows (m_pad <= 256) take the split-K path the encoder
 * uses for query-time embedding (tests). */
int rass_gemm_bf16_ws(const void* d_x, const void* d_w, const float* d_bias, const void* d_residual, void* d_y, int m,
                      int m_pad, int n, int k, int epilogue, void* d_ws, size_t ws_bytes, void* stream) {
    rass::rass_env_new_scope();
    if (!d_x || !d_w || !d_bias || !d_y) return efail(RASS_ERR_INVALID, "NULL argument");
    hipError_t err = rass::launch_gemm_bf16(d_x, d_w, d_bias, d_residual, d_y, m, m_pad, n, k, epilogue,
                                            reinterpret_cast<hipStream_t>(stream), static_cast<float*>(d_ws), ws_bytes);
    if (err != hipSuccess) return efail(RASS_ERR_INVALID, std::string("gemm launch: ") + hipGetErrorString(err));
    return RASS_OK;
}

/* Stand-alone launcher of the encoder's attention (tests, micro-benchmarks). */
int rass_attention_bf16(const void* d_qkv, const int32_t* d_cu_seqlens, int nseq, int total_tokens, int max_seqlen,
                        int hidden, int heads, void* d_ctx, void* stream) {
    rass::rass_env_new_scope();
    if (!d_qkv || !d_cu_seqlens || !d_ctx) return efail(RASS_ERR_INVALID, "NULL argument");
    hipError_t err = rass::launch_attention(d_qkv, d_cu_seqlens, nseq, total_tokens, max_seqlen, hidden, heads, d_ctx,
                                            reinterpret_cast<hipStream_t>(stream));
    if (err != hipSuccess) return efail(RASS_ERR_INVALID, std::string("attention launch: ") + hipGetErrorString(err));
    return RASS_OK;
}

/* Stand-alone launcher of the query-time fusion of the two (tests, micro-benchmarks): d_y = attention(d_qkv) d_w^T + d_bias +
 * d_residual in one launch; RASS_ERR_UNSUPPORTED outside its range (or with RASS_ATTN_FUSE=0). */
int rass_attention_out_bf16(const void* d_qkv, const int32_t* d_cu_seqlens, int nseq, int total_tokens, int hidden, int heads,
                            const void* d_w, const float* d_bias, const void* d_residual, void* d_y, int n, void* stream) {
    rass::rass_env_new_scope();
    if (!d_qkv || !d_cu_seqlens || !d_w || !d_bias || !d_residual || !d_y) return efail(RASS_ERR_INVALID, "NULL argument");
    if (!rass::attn_out_fused_ok(total_tokens, nseq, hidden, heads, n))
        return efail(RASS_ERR_UNSUPPORTED, "fused attention + output GEMM: 1..32 tokens in all, hidden 1024, 16 heads, n % 16 == 0");
    hipError_t err = rass::launch_attn_out_fused(d_qkv, d_cu_seqlens, nseq, total_tokens, hidden, heads, d_w, d_bias, d_residual,
                                                 d_y, n, reinterpret_cast<hipStream_t>(stream));
    if (err != hipSuccess) return efail(RASS_ERR_INVALID, std::string("fused attention launch: ") + hipGetErrorString(err));
    return RASS_OK;
}

}  // extern "C"
