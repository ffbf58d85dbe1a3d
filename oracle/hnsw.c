/* hnsw.c — CPU restatement of the reference's approximate k-NN structure.  TEST INFRASTRUCTURE
 * ONLY (same rules as rass_oracle.c: tests/, smoke() and the cpu_baseline legs of the bench
 * scripts may use it; the product never does).
 *
 * What it restates.  The reference creates its vector field as
 *     knn_vector, dimension 1024, method {name "hnsw", space_type "cosinesimil",
 *     engine "nmslib", parameters {m 48, ef_construction 400}}        (app/main.py:563-572)
 * and queries it with {"knn": {"embedding": {"vector": q, "k": k}}}    (app/main.py:1093-1107).
 * The graph code itself lives in OpenSearch's k-NN plugin / nmslib, absent from the reference
 * tree, so this file restates the PUBLISHED algorithm those engines implement —
 * Malkov & Yashunin, "Efficient and robust approximate nearest neighbor search using
 * Hierarchical Navigable Small World graphs" (Alg. 1 insert, Alg. 2 search-layer, Alg. 4
 * heuristic neighbour selection, Alg. 5 k-NN search) — with the reference's parameters:
 * M = 48 (2M on layer 0), ef_construction = 400, distance 1 - cos on unit vectors.
 * ef_search is a parameter of the search call (the plugin's index-level default for this
 * engine generation is 512; the reference does not override it).
 *
 * "Parity unpinned": node levels come from this file's own RNG and ties are broken by id, so
 * the graph is *an* HNSW with the reference's parameters, not OpenSearch's byte-identical graph.
 * It is used for two things only: (i) a CPU baseline that has the reference's algorithmic
 * complexity (cfg 1), (ii) measuring how far HNSW's answers are from the exact top-k that the
 * HIP path returns (recall of the reference's algorithm vs ours = 1.0 by construction).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct {
    float d;
    int32_t id;
} cand_t;

typedef struct {
    int64_t n;
    int dim, M, M0, efc;
    int64_t stride;
    const float* x;    /* unit rows, not owned */
    int32_t* level;    /* [n] */
    int32_t* link0;    /* [n][M0+1]: count, neighbours */
    int32_t** linkup;  /* [n] -> [level][M+1] or NULL */
    int32_t entry;
    int32_t max_level;
    int64_t dist_evals;
} hnsw_t;

static inline float dist_rows(const hnsw_t* h, const float* q, int32_t b) {
    const float* y = h->x + (int64_t)b * h->stride;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= h->dim; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += q[i + j] * y[i + j];
    float s = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    for (; i < h->dim; ++i) s += q[i] * y[i];
    return 1.0f - s;
}

static inline int32_t* links_of(const hnsw_t* h, int32_t v, int lev) {
    return lev == 0 ? h->link0 + (int64_t)v * (h->M0 + 1) : h->linkup[v] + (int64_t)(lev - 1) * (h->M + 1);
}

/* total order on candidates: distance, then id (deterministic ties) */
static inline int cand_less(cand_t a, cand_t b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }

/* binary heaps over cand_t; `maxheap` selects the comparison */
typedef struct {
    cand_t* a;
    int n, cap;
} heap_t;

static void heap_push(heap_t* hp, cand_t c, int maxheap) {
    if (hp->n == hp->cap) {
        hp->cap = hp->cap ? hp->cap * 2 : 64;
        hp->a = (cand_t*)realloc(hp->a, (size_t)hp->cap * sizeof(cand_t));
    }
    int i = hp->n++;
    while (i > 0) {
        const int p = (i - 1) >> 1;
        const int up = maxheap ? cand_less(hp->a[p], c) : cand_less(c, hp->a[p]);
        if (!up) break;
        hp->a[i] = hp->a[p];
        i = p;
    }
    hp->a[i] = c;
}

static cand_t heap_pop(heap_t* hp, int maxheap) {
    const cand_t top = hp->a[0];
    const cand_t last = hp->a[--hp->n];
    int i = 0;
    for (;;) {
        int c = 2 * i + 1;
        if (c >= hp->n) break;
        if (c + 1 < hp->n && (maxheap ? cand_less(hp->a[c], hp->a[c + 1]) : cand_less(hp->a[c + 1], hp->a[c]))) ++c;
        const int down = maxheap ? cand_less(last, hp->a[c]) : cand_less(hp->a[c], last);
        if (!down) break;
        hp->a[i] = hp->a[c];
        i = c;
    }
    if (hp->n > 0) hp->a[i] = last;
    return top;
}

typedef struct {
    uint32_t* stamp;
    uint32_t epoch;
    heap_t cand, res;
    cand_t* tmp;
    int tmp_cap;
    int64_t evals;
} scratch_t;

static scratch_t* scratch_new(int64_t n) {
    scratch_t* s = (scratch_t*)calloc(1, sizeof(scratch_t));
    s->stamp = (uint32_t*)calloc((size_t)n, sizeof(uint32_t));
    return s;
}

static void scratch_free(scratch_t* s) {
    if (!s) return;
    free(s->stamp);
    free(s->cand.a);
    free(s->res.a);
    free(s->tmp);
    free(s);
}

/* Alg. 2: ef closest to q on layer `lev`, starting from ep.  Result left in s->res (max-heap). */
static void search_layer(const hnsw_t* h, scratch_t* s, const float* q, cand_t ep, int ef, int lev) {
    if (++s->epoch == 0) {
        memset(s->stamp, 0, (size_t)h->n * sizeof(uint32_t));
        s->epoch = 1;
    }
    s->cand.n = 0;
    s->res.n = 0;
    s->stamp[ep.id] = s->epoch;
    heap_push(&s->cand, ep, 0);
    heap_push(&s->res, ep, 1);
    while (s->cand.n > 0) {
        const cand_t c = heap_pop(&s->cand, 0);
        if (s->res.n >= ef && cand_less(s->res.a[0], c)) break;
        const int32_t* lk = links_of(h, c.id, lev);
        const int cnt = lk[0];
        for (int j = 1; j <= cnt; ++j) {
            const int32_t e = lk[j];
            if (s->stamp[e] == s->epoch) continue;
            s->stamp[e] = s->epoch;
            const cand_t ce = {dist_rows(h, q, e), e};
            ++s->evals;
            if (s->res.n < ef || cand_less(ce, s->res.a[0])) {
                heap_push(&s->cand, ce, 0);
                heap_push(&s->res, ce, 1);
                if (s->res.n > ef) heap_pop(&s->res, 1);
            }
        }
    }
}

static int cmp_cand(const void* a, const void* b) {
    const cand_t x = *(const cand_t*)a, y = *(const cand_t*)b;
    return cand_less(x, y) ? -1 : cand_less(y, x) ? 1 : 0;
}

/* Alg. 4 (no extend, no keep-pruned): from `c` (sorted ascending), keep a candidate only if
 * it is closer to the base point than to every neighbour kept so far.  Returns the count. */
static int select_heuristic(const hnsw_t* h, scratch_t* s, cand_t* c, int nc, int M, int32_t* out) {
    int kept = 0;
    for (int i = 0; i < nc && kept < M; ++i) {
        const float* xi = h->x + (int64_t)c[i].id * h->stride;
        int ok = 1;
        for (int j = 0; j < kept; ++j) {
            ++s->evals;
            if (dist_rows(h, xi, out[j]) < c[i].d) {
                ok = 0;
                break;
            }
        }
        if (ok) out[kept++] = c[i].id;
    }
    return kept;
}

static cand_t* tmp_reserve(scratch_t* s, int n) {
    if (n > s->tmp_cap) {
        s->tmp_cap = n * 2;
        s->tmp = (cand_t*)realloc(s->tmp, (size_t)s->tmp_cap * sizeof(cand_t));
    }
    return s->tmp;
}

static uint64_t splitmix64(uint64_t* st) {
    uint64_t z = (*st += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static void insert(hnsw_t* h, scratch_t* s, int32_t v) {
    const float* q = h->x + (int64_t)v * h->stride;
    const int lv = h->level[v];
    if (h->entry < 0) {
        h->entry = v;
        h->max_level = lv;
        return;
    }
    cand_t ep = {dist_rows(h, q, h->entry), h->entry};
    ++s->evals;
    for (int lev = h->max_level; lev > lv; --lev) { /* greedy descent, ef = 1 */
        search_layer(h, s, q, ep, 1, lev);
        ep = s->res.a[0];
    }
    int32_t* sel = (int32_t*)malloc((size_t)(h->M0 + 1) * sizeof(int32_t));
    for (int lev = lv < h->max_level ? lv : h->max_level; lev >= 0; --lev) {
        search_layer(h, s, q, ep, h->efc, lev);
        const int nc = s->res.n;
        cand_t* c = tmp_reserve(s, nc + h->M0 + 2);
        memcpy(c, s->res.a, (size_t)nc * sizeof(cand_t));
        qsort(c, (size_t)nc, sizeof(cand_t), cmp_cand);
        ep = c[0];
        const int Mmax = lev == 0 ? h->M0 : h->M;
        const int ns = select_heuristic(h, s, c, nc, h->M, sel);
        int32_t* mine = links_of(h, v, lev);
        mine[0] = ns;
        memcpy(mine + 1, sel, (size_t)ns * sizeof(int32_t));
        for (int j = 0; j < ns; ++j) { /* back links, shrinking with the same heuristic */
            const int32_t u = sel[j];
            int32_t* lk = links_of(h, u, lev);
            if (lk[0] < Mmax) {
                lk[++lk[0]] = v;
                continue;
            }
            const float* xu = h->x + (int64_t)u * h->stride;
            cand_t* cu = tmp_reserve(s, Mmax + 2);
            int m = 0;
            cu[m++] = (cand_t){dist_rows(h, xu, v), v};
            for (int t = 1; t <= lk[0]; ++t) cu[m++] = (cand_t){dist_rows(h, xu, lk[t]), lk[t]};
            s->evals += m;
            qsort(cu, (size_t)m, sizeof(cand_t), cmp_cand);
            int32_t keep[512];
            const int nk = select_heuristic(h, s, cu, m, Mmax, keep);
            lk[0] = nk;
            memcpy(lk + 1, keep, (size_t)nk * sizeof(int32_t));
        }
    }
    free(sel);
    if (lv > h->max_level) {
        h->max_level = lv;
        h->entry = v;
    }
}

void rass_oracle_hnsw_free(void* p) {
    hnsw_t* h = (hnsw_t*)p;
    if (!h) return;
    if (h->linkup)
        for (int64_t i = 0; i < h->n; ++i) free(h->linkup[i]);
    free(h->linkup);
    free(h->link0);
    free(h->level);
    free(h);
}

/* Build over n unit rows (row-major, `stride` floats apart; the caller keeps them alive).
 * Sequential inserts in row order, as a single-shard bulk index would. */
void* rass_oracle_hnsw_build(const float* x, int64_t n, int dim, int64_t stride, int M, int efc, uint64_t seed) {
    if (!x || n <= 0 || n > INT32_MAX || dim <= 0 || M < 2 || 2 * M > 511 || efc < 1) return NULL;
    hnsw_t* h = (hnsw_t*)calloc(1, sizeof(hnsw_t));
    h->n = n; h->dim = dim; h->stride = stride; h->M = M; h->M0 = 2 * M; h->efc = efc; h->x = x;
    h->entry = -1;
    h->level = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    h->link0 = (int32_t*)calloc((size_t)n * (size_t)(h->M0 + 1), sizeof(int32_t));
    h->linkup = (int32_t**)calloc((size_t)n, sizeof(int32_t*));
    const double mL = 1.0 / log((double)M);
    uint64_t st = seed ? seed : 1;
    for (int64_t i = 0; i < n; ++i) {
        const double u = ((double)(splitmix64(&st) >> 11) + 1.0) / 9007199254740993.0; /* (0,1) */
        int lv = (int)floor(-log(u) * mL);
        if (lv > 30) lv = 30;
        h->level[i] = lv;
        if (lv > 0) h->linkup[i] = (int32_t*)calloc((size_t)lv * (size_t)(M + 1), sizeof(int32_t));
    }
    scratch_t* s = scratch_new(n);
    for (int64_t i = 0; i < n; ++i) insert(h, s, (int32_t)i);
    h->dist_evals = s->evals;
    scratch_free(s);
    return h;
}

int64_t rass_oracle_hnsw_build_evals(const void* p) { return p ? ((const hnsw_t*)p)->dist_evals : -1; }
int rass_oracle_hnsw_max_level(const void* p) { return p ? ((const hnsw_t*)p)->max_level : -1; }

/* Alg. 5 for nq unit queries; scores are cosines (1 - distance), ids -1 / scores -inf padded.
 * Returns the total number of distance evaluations, < 0 on bad arguments. */
int64_t rass_oracle_hnsw_search(const void* p, const float* q, int nq, int64_t q_stride, int k, int ef, int threads,
                                float* out_scores, int64_t* out_ids) {
    const hnsw_t* h = (const hnsw_t*)p;
    if (!h || !q || nq < 0 || k < 1 || !out_scores || !out_ids) return -1;
    if (ef < k) ef = k;
    if (threads < 1) threads = omp_get_max_threads();
    int64_t total = 0;
#pragma omp parallel num_threads(threads) reduction(+ : total)
    {
        scratch_t* s = scratch_new(h->n);
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < nq; ++r) {
            const float* qr = q + (int64_t)r * q_stride;
            cand_t ep = {dist_rows(h, qr, h->entry), h->entry};
            ++s->evals;
            for (int lev = h->max_level; lev > 0; --lev) {
                search_layer(h, s, qr, ep, 1, lev);
                ep = s->res.a[0];
            }
            search_layer(h, s, qr, ep, ef, 0);
            const int nc = s->res.n;
            cand_t* c = tmp_reserve(s, nc + 1);
            memcpy(c, s->res.a, (size_t)nc * sizeof(cand_t));
            qsort(c, (size_t)nc, sizeof(cand_t), cmp_cand);
            for (int j = 0; j < k; ++j) {
                out_scores[(int64_t)r * k + j] = j < nc ? 1.0f - c[j].d : -INFINITY;
                out_ids[(int64_t)r * k + j] = j < nc ? c[j].id : -1;
            }
        }
        total += s->evals;
        scratch_free(s);
    }
    return total;
}
