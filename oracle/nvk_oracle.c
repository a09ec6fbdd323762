/* oracle/nvk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's paged-attention decode path, used
 * only as the checker for the HIP path (tests/, __graft_entry__.smoke(), and
 * the cpu_baseline leg of bench.py).  Nothing under nano_vllm_amd/ links,
 * loads or calls it; the product path has no CPU fallback.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors
 * (SURVEY.md section 4), so the pin is (a) tests/test_oracle_vs_ref.py, which
 * compares every function here bit-for-bit with the reference's own headers
 * compiled into oracle/_ref/libnvk_ref.so (in the build container, where
 * /root/reference exists), and (b) the committed fixtures tests/golden/ *.npz
 * that oracle/make_golden.py generated from that same library.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  All pool offsets are 64-bit here; the reference computes
 * them in int (attention.hpp:114-115), which limits it to pools < 2^31 floats.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ RNG --
 * Counter-based value generator of SURVEY.md section 8d (ours, not the
 * reference's): value(seed, i) = 2*u - 1, u = top 24 bits of
 * splitmix64_finalise((seed ^ i) + golden) scaled to [0,1). */
static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

uint64_t oracle_splitmix64(uint64_t seed, uint64_t index) { return mix64(seed ^ index); }

static inline float unit_from_bits(uint64_t z)
{
    return (float)(z >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
}

void oracle_fill_uniform(float *dst, uint64_t seed, uint64_t first_index, uint64_t count)
{
    for (uint64_t i = 0; i < count; i++)
        dst[i] = unit_from_bits(mix64(seed ^ (first_index + i)));
}

/* ------------------------------------------------------------ attention --
 * Restates Attention::paged_attention, include/core/attention.hpp:82-159.
 *   kv_mul/scale                      :94-95
 *   out zeroed first                  :98
 *   per q-head loop (outermost)       :102-105
 *   QK^T through the block table      :108-123  (address :109-115)
 *   3-pass softmax, max init -1e10    :126-140
 *   PV through the block table        :144-157
 * Same loop order and same serial fp32 accumulation, so with equal compiler
 * flags the result is bit-identical to the reference. */
void oracle_paged_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                            const int *block_table, float *att_scores, int num_tokens, int block_size,
                            int head_dim, int n_heads, int n_kv_heads)
{
    const int     group   = n_heads / n_kv_heads;
    const float   inv_rt  = 1.0f / sqrtf((float)head_dim);
    const int64_t row_len = (int64_t)n_kv_heads * head_dim;    /* floats per token row     */
    const int64_t blk_len = (int64_t)block_size * row_len;     /* floats per physical block */

    memset(out, 0, (size_t)n_heads * head_dim * sizeof(float));

    for (int head = 0; head < n_heads; head++) {
        const float  *qv      = q + (int64_t)head * head_dim;
        float        *w       = att_scores + (int64_t)head * num_tokens;
        const int64_t head_at = (int64_t)(head / group) * head_dim;

        for (int t = 0; t < num_tokens; t++) {
            const int64_t at = (int64_t)block_table[t / block_size] * blk_len + (int64_t)(t % block_size) * row_len
                             + head_at;
            const float *kv  = key_cache + at;
            float        acc = 0.0f;
            for (int d = 0; d < head_dim; d++)
                acc += qv[d] * kv[d];
            acc *= inv_rt;
            w[t] = acc;
        }

        float top = -1e10;
        for (int t = 0; t < num_tokens; t++)
            if (w[t] > top)
                top = w[t];

        float denom = 0.0f;
        for (int t = 0; t < num_tokens; t++) {
            w[t] = expf(w[t] - top);
            denom += w[t];
        }
        for (int t = 0; t < num_tokens; t++)
            w[t] /= denom;

        float *ov = out + (int64_t)head * head_dim;
        for (int t = 0; t < num_tokens; t++) {
            const int64_t at = (int64_t)block_table[t / block_size] * blk_len + (int64_t)(t % block_size) * row_len
                             + head_at;
            const float *vv = value_cache + at;
            const float  p  = w[t];
            for (int d = 0; d < head_dim; d++)
                ov[d] += p * vv[d];
        }
    }
}

/* Restates Attention::standard_attention, include/core/attention.hpp:16-76:
 * identical maths over a contiguous [max_seq_len, n_kv_heads, head_dim] cache,
 * tokens 0..pos inclusive (:39), scratch stride max_seq_len (:35). */
void oracle_standard_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                               float *att_scores, int pos, int head_dim, int n_heads, int n_kv_heads,
                               int max_seq_len)
{
    const int     group   = n_heads / n_kv_heads;
    const float   inv_rt  = 1.0f / sqrtf((float)head_dim);
    const int64_t row_len = (int64_t)n_kv_heads * head_dim;

    memset(out, 0, (size_t)n_heads * head_dim * sizeof(float));

    for (int head = 0; head < n_heads; head++) {
        const float  *qv      = q + (int64_t)head * head_dim;
        float        *w       = att_scores + (int64_t)head * max_seq_len;
        const int64_t head_at = (int64_t)(head / group) * head_dim;

        for (int t = 0; t <= pos; t++) {
            const float *kv  = key_cache + (int64_t)t * row_len + head_at;
            float        acc = 0.0f;
            for (int d = 0; d < head_dim; d++)
                acc += qv[d] * kv[d];
            acc *= inv_rt;
            w[t] = acc;
        }
        float top = -1e10;
        for (int t = 0; t <= pos; t++)
            if (w[t] > top)
                top = w[t];
        float denom = 0.0f;
        for (int t = 0; t <= pos; t++) {
            w[t] = expf(w[t] - top);
            denom += w[t];
        }
        for (int t = 0; t <= pos; t++)
            w[t] /= denom;
        float *ov = out + (int64_t)head * head_dim;
        for (int t = 0; t <= pos; t++) {
            const float *vv = value_cache + (int64_t)t * row_len + head_at;
            const float  p  = w[t];
            for (int d = 0; d < head_dim; d++)
                ov[d] += p * vv[d];
        }
    }
}

/* Restates LlamaModelChunked::chunked_attention,
 * experimental/chunked_prefill/model_chunked.hpp:167-224: causal attention of a chunk of
 * chunk_size query tokens at positions start_pos + b over one layer's contiguous
 * [max_seq_len, n_kv_heads, head_dim] cache; query b attends to t <= start_pos + b (:186).
 * Scratch att is [chunk_size, n_heads, max_seq_len] (:183).  Same loops as the decode
 * attention per query, so row b equals oracle_standard_attention(pos = start_pos + b). */
void oracle_chunked_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                              float *att, int chunk_size, int start_pos, int head_dim, int n_heads, int n_kv_heads,
                              int max_seq_len)
{
    const int     kv_mul  = n_heads / n_kv_heads;
    const float   scale   = 1.0f / sqrtf((float)head_dim);
    const int64_t row_len = (int64_t)n_kv_heads * head_dim;
    memset(out, 0, (size_t)chunk_size * n_heads * head_dim * sizeof(float)); /* :176 */
    for (int b = 0; b < chunk_size; b++) {
        const int curr = start_pos + b;
        for (int h = 0; h < n_heads; h++) {
            const float  *qh   = q + ((int64_t)b * n_heads + h) * head_dim;
            float        *w    = att + ((int64_t)b * n_heads + h) * max_seq_len;
            const int64_t at_h = (int64_t)(h / kv_mul) * head_dim;
            for (int t = 0; t <= curr; t++) {
                const float *kv  = key_cache + (int64_t)t * row_len + at_h;
                float        acc = 0.0f;
                for (int i = 0; i < head_dim; i++)
                    acc += qh[i] * kv[i];
                acc *= scale;
                w[t] = acc;
            }
            float top = -1e10f;
            for (int t = 0; t <= curr; t++)
                if (w[t] > top)
                    top = w[t];
            float denom = 0.0f;
            for (int t = 0; t <= curr; t++) {
                w[t] = expf(w[t] - top);
                denom += w[t];
            }
            for (int t = 0; t <= curr; t++)
                w[t] /= denom;
            float *oh = out + ((int64_t)b * n_heads + h) * head_dim;
            for (int t = 0; t <= curr; t++) {
                const float *vv = value_cache + (int64_t)t * row_len + at_h;
                const float  p  = w[t];
                for (int i = 0; i < head_dim; i++)
                    oh[i] += p * vv[i];
            }
        }
    }
}

/* The paged form of the same thing (ours: the reference's chunked prefill only knows the
 * contiguous cache): query b of the chunk is Attention::paged_attention with
 * num_tokens = start_pos + b + 1 over the sequence's block table. */
void oracle_paged_prefill_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                                    const int *block_table, float *att_scores, int chunk_size, int start_pos,
                                    int block_size, int head_dim, int n_heads, int n_kv_heads)
{
    for (int b = 0; b < chunk_size; b++)
        oracle_paged_attention(out + (int64_t)b * n_heads * head_dim, q + (int64_t)b * n_heads * head_dim, key_cache,
                               value_cache, block_table, att_scores, start_pos + b + 1, block_size, head_dim, n_heads,
                               n_kv_heads);
}

/* Batched decode = the reference function once per sequence (SURVEY.md TL;DR:
 * the reference has no batched decode; parity for a batch is defined per
 * sequence).  block_tables is row-padded [num_seqs, max_blocks]. */
typedef struct
{
    float       *out;
    const float *q;
    const float *key_cache;
    const float *value_cache;
    const int   *block_tables;
    const int   *ctx_lens;
    int          max_blocks, block_size, head_dim, n_heads, n_kv_heads;
    int          seq_begin, seq_end, seq_stride;
    int          max_ctx;
} batch_job;

static void *batch_worker(void *arg)
{
    const batch_job *j     = (const batch_job *)arg;
    const size_t     qlen  = (size_t)j->n_heads * j->head_dim;
    float           *score = (float *)malloc((size_t)j->n_heads * (size_t)(j->max_ctx > 0 ? j->max_ctx : 1)
                                             * sizeof(float));
    if (!score)
        return NULL;
    for (int s = j->seq_begin; s < j->seq_end; s += j->seq_stride)
        oracle_paged_attention(j->out + s * qlen, j->q + s * qlen, j->key_cache, j->value_cache,
                               j->block_tables + (size_t)s * j->max_blocks, score, j->ctx_lens[s], j->block_size,
                               j->head_dim, j->n_heads, j->n_kv_heads);
    free(score);
    return NULL;
}

/* n_threads <= 1: serial on the calling thread (faithful to the reference's
 * single thread); otherwise sequences are dealt round-robin to n_threads
 * pthreads, one score scratch each ("all host cores"). */
int oracle_decode_batch(float *out, const float *q, const float *key_cache, const float *value_cache,
                        const int *block_tables, int max_blocks, const int *ctx_lens, int num_seqs,
                        int block_size, int head_dim, int n_heads, int n_kv_heads, int n_threads)
{
    int max_ctx = 0;
    for (int s = 0; s < num_seqs; s++)
        if (ctx_lens[s] > max_ctx)
            max_ctx = ctx_lens[s];
    batch_job proto = {out,        q,          key_cache, value_cache, block_tables, ctx_lens, max_blocks,
                       block_size, head_dim,   n_heads,   n_kv_heads,  0,            num_seqs, 1,
                       max_ctx};
    if (n_threads <= 1) {
        batch_worker(&proto);
        return 0;
    }
    if (n_threads > 256)
        n_threads = 256;
    pthread_t tid[256];
    batch_job jobs[256];
    for (int i = 0; i < n_threads; i++) {
        jobs[i]            = proto;
        jobs[i].seq_begin  = i;
        jobs[i].seq_stride = n_threads;
        if (pthread_create(&tid[i], NULL, batch_worker, &jobs[i]) != 0)
            return -1;
    }
    for (int i = 0; i < n_threads; i++)
        pthread_join(tid[i], NULL);
    return 0;
}

/* KV append: restates the two memcpys of LlamaModel::forward,
 * include/core/model.hpp:190-207 (layer slab :196-197, block :198, row :199). */
void oracle_kv_append(float *key_pool, float *value_pool, const float *k, const float *v, int layer, int num_blocks,
                      int block_size, int n_kv_heads, int head_dim, int physical_block, int block_offset)
{
    const int64_t row_len = (int64_t)n_kv_heads * head_dim;
    const int64_t at = ((int64_t)layer * num_blocks + physical_block) * block_size * row_len + block_offset * row_len;
    memcpy(key_pool + at, k, (size_t)row_len * sizeof(float));
    memcpy(value_pool + at, v, (size_t)row_len * sizeof(float));
}

/* --------------------------------------------------------- BlockManager --
 * Restates include/scheduler/block_manager.hpp:14-283 with the same data
 * structure (one flag per block, linear first-fit scan), so IDs come out in
 * the reference's order.  C has no exceptions: where the reference throws we
 * return ORACLE_BM_THROW and leave the state as the reference leaves it. */
#define ORACLE_BM_THROW (-2)

typedef struct
{
    int  id;
    int *blocks;
    int  n, cap;
} bm_request;

typedef struct
{
    int            num_blocks, block_size, num_free;
    unsigned char *is_free;
    bm_request    *req;
    int            n_req, cap_req;
} oracle_bm;

oracle_bm *oracle_bm_create(int num_blocks, int block_size) /* :17-24 */
{
    oracle_bm *b  = (oracle_bm *)calloc(1, sizeof *b);
    b->num_blocks = num_blocks;
    b->block_size = block_size;
    b->num_free   = num_blocks;
    b->is_free    = (unsigned char *)malloc((size_t)(num_blocks > 0 ? num_blocks : 1));
    memset(b->is_free, 1, (size_t)(num_blocks > 0 ? num_blocks : 1));
    return b;
}

void oracle_bm_destroy(oracle_bm *b)
{
    if (!b)
        return;
    for (int i = 0; i < b->n_req; i++)
        free(b->req[i].blocks);
    free(b->req);
    free(b->is_free);
    free(b);
}

int oracle_bm_allocate_block(oracle_bm *b) /* :28-44 and :231-244 */
{
    if (b->num_free == 0)
        return -1;
    for (int i = 0; i < b->num_blocks; i++)
        if (b->is_free[i]) {
            b->is_free[i] = 0;
            b->num_free--;
            return i;
        }
    return -1;
}

int oracle_bm_free_block(oracle_bm *b, int id) /* :48-62 */
{
    if (id < 0 || id >= b->num_blocks)
        return ORACLE_BM_THROW; /* "Invalid block_id" :52 */
    if (b->is_free[id])
        return 0; /* double free: warning + no-op :55-58 */
    b->is_free[id] = 1;
    b->num_free++;
    return 0;
}

static void free_block_quiet(oracle_bm *b, int id) /* :247-256 */
{
    if (id < 0 || id >= b->num_blocks)
        return;
    if (!b->is_free[id]) {
        b->is_free[id] = 1;
        b->num_free++;
    }
}

/* :67-92.  Returns the count written to ids, or ORACLE_BM_THROW ("Out of
 * memory" :73; the rollback branch :81-86 cannot fire single-threaded). */
int oracle_bm_allocate_sequence(oracle_bm *b, int num_tokens, int *ids, int cap)
{
    const int need = (num_tokens + b->block_size - 1) / b->block_size;
    if (need > b->num_free)
        return ORACLE_BM_THROW;
    int got = 0;
    for (int i = 0; i < need; i++) {
        const int id = oracle_bm_allocate_block(b);
        if (id == -1) {
            for (int k = 0; k < got && k < cap; k++)
                oracle_bm_free_block(b, ids[k]);
            return ORACLE_BM_THROW;
        }
        if (got < cap)
            ids[got] = id;
        got++;
    }
    return got;
}

int oracle_bm_free_sequence(oracle_bm *b, const int *ids, int n) /* :96-101 */
{
    for (int i = 0; i < n; i++)
        if (oracle_bm_free_block(b, ids[i]) == ORACLE_BM_THROW)
            return ORACLE_BM_THROW; /* reference stops at the first throw */
    return 0;
}

int   oracle_bm_num_free(const oracle_bm *b) { return b->num_free; }     /* :107 */
int   oracle_bm_num_blocks(const oracle_bm *b) { return b->num_blocks; } /* :110 */
int   oracle_bm_block_size(const oracle_bm *b) { return b->block_size; } /* :113 */
int   oracle_bm_is_free(const oracle_bm *b, int id)                      /* :116-121 */
{
    if (id < 0 || id >= b->num_blocks)
        return 0;
    return b->is_free[id];
}
float oracle_bm_utilization(const oracle_bm *b) /* :124 */
{
    return 1.0f - ((float)b->num_free / b->num_blocks);
}

static bm_request *find_request(oracle_bm *b, int request_id, int create)
{
    for (int i = 0; i < b->n_req; i++)
        if (b->req[i].id == request_id)
            return &b->req[i];
    if (!create)
        return NULL;
    if (b->n_req == b->cap_req) {
        b->cap_req = b->cap_req ? 2 * b->cap_req : 8;
        b->req     = (bm_request *)realloc(b->req, (size_t)b->cap_req * sizeof *b->req);
    }
    bm_request *r = &b->req[b->n_req++];
    r->id         = request_id;
    r->blocks     = NULL;
    r->n = r->cap = 0;
    return r;
}

static void request_push(bm_request *r, int id)
{
    if (r->n == r->cap) {
        r->cap    = r->cap ? 2 * r->cap : 8;
        r->blocks = (int *)realloc(r->blocks, (size_t)r->cap * sizeof(int));
    }
    r->blocks[r->n++] = id;
}

/* :132-141 over :259-282: non-throwing; returns 0 ids when it cannot fit. */
int oracle_bm_allocate_for_request(oracle_bm *b, int request_id, int num_tokens, int *ids, int cap)
{
    const int need = (num_tokens + b->block_size - 1) / b->block_size;
    if (need > b->num_free)
        return 0;
    int got = 0;
    for (int i = 0; i < need; i++) {
        const int id = oracle_bm_allocate_block(b);
        if (id == -1) {
            for (int k = 0; k < got && k < cap; k++)
                free_block_quiet(b, ids[k]);
            return 0;
        }
        if (got < cap)
            ids[got] = id;
        got++;
    }
    if (got > 0) { /* :137-139: the map entry is created only for a non-empty result */
        bm_request *r = find_request(b, request_id, 1);
        for (int k = 0; k < got && k < cap; k++)
            request_push(r, ids[k]);
    }
    return got;
}

int oracle_bm_allocate_block_for_request(oracle_bm *b, int request_id) /* :145-154 */
{
    const int id = oracle_bm_allocate_block(b);
    if (id >= 0)
        request_push(find_request(b, request_id, 1), id);
    return id;
}

void oracle_bm_free_request(oracle_bm *b, int request_id) /* :157-171 */
{
    bm_request *r = find_request(b, request_id, 0);
    if (!r)
        return;
    for (int i = 0; i < r->n; i++)
        free_block_quiet(b, r->blocks[i]);
    free(r->blocks);
    *r = b->req[--b->n_req];
}

int oracle_bm_get_request_blocks(oracle_bm *b, int request_id, int *ids, int cap) /* :174-183 */
{
    bm_request *r = find_request(b, request_id, 0);
    if (!r)
        return 0;
    for (int i = 0; i < r->n && i < cap; i++)
        ids[i] = r->blocks[i];
    return r->n;
}

int oracle_bm_get_request_block_count(oracle_bm *b, int request_id) /* :186-195 */
{
    bm_request *r = find_request(b, request_id, 0);
    return r ? r->n : 0;
}

int oracle_bm_num_active_requests(const oracle_bm *b) { return b->n_req; } /* :198-202 */

/* ---------------------------------------------------------------- model --
 * Restates the llama2.c-style single-token forward of LlamaModel,
 * include/core/model.hpp:118-247,316-365,393-416, with the paged path always
 * on (src/main.cpp:74), for the token-stream fixture only.  Ops follow
 * include/ops/linear.hpp:12-22, normalization.hpp:13-23,
 * positional.hpp:13-39, activation.hpp:34-40; greedy sampling follows
 * include/core/sampler.hpp:27-29 (first maximum wins). */
typedef struct
{
    int dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, max_seq_len, head_dim;
    int block_size, num_blocks;
    /* weights, in file order (model.hpp:322-347) */
    float *tok_emb, *rms_att, *wq, *wk, *wv, *wo, *rms_ffn, *w_gate, *w_down, *w_up, *rms_final, *lm_head;
    int    shared_head;
    /* run state */
    float *x, *xb, *xb2, *hb, *hb2, *q, *k, *v, *att, *logits;
    float *key_pool, *value_pool; /* [n_layers, num_blocks, block_size, n_kv_heads, head_dim] */
    oracle_bm *bm;
    int      **tables; /* [n_layers][...] */
    int       *table_len;
} oracle_model;

static float *read_floats(FILE *f, size_t n)
{
    float *p = (float *)malloc((n ? n : 1) * sizeof(float));
    if (fread(p, sizeof(float), n, f) != n) {
        free(p);
        return NULL;
    }
    return p;
}

void oracle_model_destroy(oracle_model *m);

oracle_model *oracle_model_create(const char *path, int block_size, int num_blocks)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return NULL;
    oracle_model *m = (oracle_model *)calloc(1, sizeof *m);
    int           hdr[7];
    if (fread(hdr, sizeof(int), 7, f) != 7) { /* model.hpp:127 */
        fclose(f);
        free(m);
        return NULL;
    }
    m->dim = hdr[0], m->hidden_dim = hdr[1], m->n_layers = hdr[2], m->n_heads = hdr[3];
    m->n_kv_heads = hdr[4], m->vocab_size = hdr[5], m->max_seq_len = hdr[6];
    m->head_dim   = m->dim / m->n_heads; /* :129 */
    m->block_size = block_size > 0 ? block_size : 16;   /* :35 */
    m->num_blocks = num_blocks > 0 ? num_blocks : 256;  /* :36 */

    const size_t L = (size_t)m->n_layers, D = (size_t)m->dim, H = (size_t)m->hidden_dim;
    const size_t QD = (size_t)m->n_heads * m->head_dim, KD = (size_t)m->n_kv_heads * m->head_dim;
    m->tok_emb   = read_floats(f, (size_t)m->vocab_size * D);
    m->rms_att   = read_floats(f, L * D);
    m->wq        = read_floats(f, L * D * QD);
    m->wk        = read_floats(f, L * D * KD);
    m->wv        = read_floats(f, L * D * KD);
    m->wo        = read_floats(f, L * QD * D);
    m->rms_ffn   = read_floats(f, L * D);
    m->w_gate    = read_floats(f, L * D * H);
    m->w_down    = read_floats(f, L * H * D);
    m->w_up      = read_floats(f, L * D * H);
    m->rms_final = read_floats(f, D);
    /* :350-364: a separate classifier only if the file still holds one */
    long here = ftell(f);
    fseek(f, 0, SEEK_END);
    long end = ftell(f);
    fseek(f, here, SEEK_SET);
    if ((size_t)(end - here) >= (size_t)m->vocab_size * D * sizeof(float))
        m->lm_head = read_floats(f, (size_t)m->vocab_size * D);
    else {
        m->shared_head = 1;
        m->lm_head     = m->tok_emb;
    }
    fclose(f);
    if (!m->tok_emb || !m->rms_att || !m->wq || !m->wk || !m->wv || !m->wo || !m->rms_ffn || !m->w_gate
        || !m->w_down || !m->w_up || !m->rms_final || !m->lm_head) {
        oracle_model_destroy(m);
        return NULL;
    }

    m->x      = (float *)calloc(D, sizeof(float));
    m->xb     = (float *)calloc(D, sizeof(float));
    m->xb2    = (float *)calloc(D, sizeof(float));
    m->hb     = (float *)calloc(H, sizeof(float));
    m->hb2    = (float *)calloc(H, sizeof(float));
    m->q      = (float *)calloc(D, sizeof(float));
    m->k      = (float *)calloc(D, sizeof(float));
    m->v      = (float *)calloc(D, sizeof(float));
    m->att    = (float *)calloc((size_t)m->n_heads * m->max_seq_len, sizeof(float));
    m->logits = (float *)calloc((size_t)m->vocab_size, sizeof(float));

    /* initialize_paged_attention, :249-277: one manager shared by all layers */
    const size_t pool = L * (size_t)m->num_blocks * m->block_size * KD;
    m->key_pool       = (float *)calloc(pool, sizeof(float));
    m->value_pool     = (float *)calloc(pool, sizeof(float));
    m->bm             = oracle_bm_create(m->num_blocks, m->block_size);
    m->tables         = (int **)calloc(L, sizeof(int *));
    m->table_len      = (int *)calloc(L, sizeof(int));
    for (size_t l = 0; l < L; l++)
        m->tables[l] = (int *)calloc((size_t)m->num_blocks + 1, sizeof(int));
    return m;
}

void oracle_model_destroy(oracle_model *m)
{
    if (!m)
        return;
    if (!m->shared_head)
        free(m->lm_head);
    free(m->tok_emb), free(m->rms_att), free(m->wq), free(m->wk), free(m->wv), free(m->wo);
    free(m->rms_ffn), free(m->w_gate), free(m->w_down), free(m->w_up), free(m->rms_final);
    free(m->x), free(m->xb), free(m->xb2), free(m->hb), free(m->hb2), free(m->q), free(m->k), free(m->v);
    free(m->att), free(m->logits), free(m->key_pool), free(m->value_pool);
    if (m->tables)
        for (int l = 0; l < m->n_layers; l++)
            free(m->tables[l]);
    free(m->tables), free(m->table_len);
    oracle_bm_destroy(m->bm);
    free(m);
}

static void gemv(float *y, const float *x, const float *w, int n_in, int n_out) /* ops/linear.hpp:12-22 */
{
    for (int r = 0; r < n_out; r++) {
        const float *row = w + (size_t)r * n_in;
        float        acc = 0.0f;
        for (int c = 0; c < n_in; c++)
            acc += x[c] * row[c];
        y[r] = acc;
    }
}

static void rmsnorm(float *y, const float *x, const float *g, int n) /* ops/normalization.hpp:13-23 */
{
    float ss = 0.0f;
    for (int i = 0; i < n; i++)
        ss += x[i] * x[i];
    const float r = 1.0f / sqrtf(ss / n + 1e-5f);
    for (int i = 0; i < n; i++)
        y[i] = x[i] * r * g[i];
}

/* The batched forms, BatchOps::batch_rms_norm / batch_matmul (experimental/chunked_prefill/batch_ops.hpp:58-76,
 * :8-23): the per-row operators above applied to every row, in row order.  Checker of nvk_batch_rms_norm_device,
 * nvk_batch_matmul_device and nvk_qkv_project_device. */
void oracle_batch_rms_norm(float *out, const float *in, const float *weight, int batch_size, int dim)
{
    for (int b = 0; b < batch_size; b++)
        rmsnorm(out + (size_t)b * dim, in + (size_t)b * dim, weight, dim);
}

void oracle_batch_matmul(float *out, const float *in, const float *weight, int batch_size, int in_dim, int out_dim)
{
    for (int b = 0; b < batch_size; b++)
        gemv(out + (size_t)b * out_dim, in + (size_t)b * in_dim, weight, in_dim, out_dim);
}

/* Restates Ops::apply_rope, include/ops/positional.hpp:13-39: pairs (i, i+1) of every
 * q-head and kv-head rotated in place by angle pos / theta^(i/head_dim). */
void oracle_apply_rope(float *q, float *k, int pos, int head_dim, int n_heads, int n_kv_heads, float theta)
{
    for (int i = 0; i < head_dim; i += 2) {
        const float freq = 1.0f / powf(theta, (float)i / head_dim);
        const float ang  = pos * freq;
        const float c = cosf(ang), s = sinf(ang);
        for (int h = 0; h < n_heads; h++) {
            float *p = q + h * head_dim + i;
            const float a = p[0], b = p[1];
            p[0] = a * c - b * s;
            p[1] = a * s + b * c;
        }
        for (int h = 0; h < n_kv_heads; h++) {
            float *p = k + h * head_dim + i;
            const float a = p[0], b = p[1];
            p[0] = a * c - b * s;
            p[1] = a * s + b * c;
        }
    }
}

/* One token through every layer (model.hpp:150-247).  Returns 0, or -1 where
 * the reference throws "Out of memory: no free blocks" (:183-185). */
int oracle_model_forward(oracle_model *m, int token, int pos, float *logits_out)
{
    const int    D = m->dim, H = m->hidden_dim, hd = m->head_dim;
    const int    QD = m->n_heads * hd, KD = m->n_kv_heads * hd;
    const size_t slab = (size_t)m->num_blocks * m->block_size * KD; /* one layer of the pool */

    memcpy(m->x, m->tok_emb + (size_t)token * D, (size_t)D * sizeof(float));
    for (int l = 0; l < m->n_layers; l++) {
        rmsnorm(m->xb, m->x, m->rms_att + (size_t)l * D, D);
        gemv(m->q, m->xb, m->wq + (size_t)l * D * QD, D, QD);
        gemv(m->k, m->xb, m->wk + (size_t)l * D * KD, D, KD);
        gemv(m->v, m->xb, m->wv + (size_t)l * D * KD, D, KD);
        oracle_apply_rope(m->q, m->k, pos, hd, m->n_heads, m->n_kv_heads, 10000.0f); /* config.rope_theta, model.hpp:40 */

        if (pos % m->block_size == 0) { /* :181-187 */
            const int fresh = oracle_bm_allocate_block(m->bm);
            if (fresh == -1)
                return -1;
            m->tables[l][m->table_len[l]++] = fresh;
        }
        oracle_kv_append(m->key_pool, m->value_pool, m->k, m->v, l, m->num_blocks, m->block_size, m->n_kv_heads, hd,
                         m->tables[l][pos / m->block_size], pos % m->block_size); /* :190-207 */

        oracle_paged_attention(m->xb2, m->q, m->key_pool + l * slab, m->value_pool + l * slab, m->tables[l], m->att,
                               pos + 1, m->block_size, hd, m->n_heads, m->n_kv_heads); /* :393-416 */

        gemv(m->xb, m->xb2, m->wo + (size_t)l * QD * D, QD, D);
        for (int i = 0; i < D; i++)
            m->x[i] += m->xb[i];
        rmsnorm(m->xb, m->x, m->rms_ffn + (size_t)l * D, D);
        gemv(m->hb, m->xb, m->w_gate + (size_t)l * D * H, D, H);
        gemv(m->hb2, m->xb, m->w_up + (size_t)l * D * H, D, H);
        for (int i = 0; i < H; i++) { /* ops/activation.hpp:34-40 */
            const float g = m->hb[i] / (1.0f + expf(-m->hb[i]));
            m->hb[i]      = g * m->hb2[i];
        }
        gemv(m->xb, m->hb, m->w_down + (size_t)l * H * D, H, D);
        for (int i = 0; i < D; i++)
            m->x[i] += m->xb[i];
    }
    rmsnorm(m->x, m->x, m->rms_final, D);
    gemv(m->logits, m->x, m->lm_head, D, m->vocab_size);
    if (logits_out)
        memcpy(logits_out, m->logits, (size_t)m->vocab_size * sizeof(float));
    return 0;
}

void oracle_model_config(const oracle_model *m, int *out8)
{
    out8[0] = m->dim, out8[1] = m->hidden_dim, out8[2] = m->n_layers, out8[3] = m->n_heads;
    out8[4] = m->n_kv_heads, out8[5] = m->vocab_size, out8[6] = m->max_seq_len, out8[7] = m->head_dim;
}

int oracle_model_block_table(const oracle_model *m, int layer, int *ids, int cap)
{
    if (layer < 0 || layer >= m->n_layers)
        return -1;
    for (int i = 0; i < m->table_len[layer] && i < cap; i++)
        ids[i] = m->tables[layer][i];
    return m->table_len[layer];
}

void oracle_model_last_attn_out(const oracle_model *m, float *out, int n)
{
    memcpy(out, m->xb2, (size_t)n * sizeof(float));
}

int oracle_greedy_sample(const float *logits, int vocab_size) /* sampler.hpp:27-29 */
{
    int best = 0;
    for (int i = 1; i < vocab_size; i++)
        if (logits[i] > logits[best])
            best = i;
    return best;
}
