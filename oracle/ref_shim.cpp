// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" wrappers (our own code) around the *reference's own headers*,
// which are compiled from where they lie under /root/reference/include and are
// never copied into this repository.  The result, oracle/_ref/libnvk_ref.so, is
// git-ignored; it exists only
//   (1) to pin our C restatement (oracle/nvk_oracle.c) against the real
//       reference code (tests/test_oracle_vs_ref.py, oracle/make_golden.py), and
//   (2) as the "reference" CPU baseline leg of bench.py.
// Nothing under nano_vllm_amd/ may load it.
//
// Wrapped reference symbols:
//   Attention::paged_attention      include/core/attention.hpp:82-159
//   Attention::standard_attention   include/core/attention.hpp:16-76
//   BlockManager                    include/scheduler/block_manager.hpp:14-283
//   LlamaModel                      include/core/model.hpp:104-433
//   Sampler (greedy branch)         include/core/sampler.hpp:24-29
//   Ops::apply_rope                 include/ops/positional.hpp:13-39
//   LlamaModelChunked::chunked_attention  experimental/chunked_prefill/model_chunked.hpp:167-224
//   Tokenizer                       include/core/tokenizer.hpp:15-149
//   Ops::rms_norm, Ops::matmul      include/ops/normalization.hpp:13-23, include/ops/linear.hpp:12-22
//   BatchOps::batch_rms_norm, batch_matmul  experimental/chunked_prefill/batch_ops.hpp:58-76, :8-23

#include <cstdio>
#include <cstring>
#include <exception>
#include <iostream>
#include <streambuf>
#include <string>
#include <vector>

#include <algorithm>
#include <chrono>
#include <cmath>

#include "core/attention.hpp"
#include "core/model.hpp"
#include "core/sampler.hpp"
#include "core/tokenizer.hpp"
#include "scheduler/block_manager.hpp"

// LlamaModelChunked::chunked_attention is a private member (model_chunked.hpp:166-167); the shim
// needs to call it on its own, so the access specifier is widened for this one include only.
#define private public
#include "experimental/chunked_prefill/model_chunked.hpp"
#undef private

namespace {

// The reference logs every BlockManager construction to std::cout; keep the
// test output readable by swallowing it while a wrapper runs.
struct NullBuf : std::streambuf
{
    int overflow(int c) override { return c; }
};

struct Quiet
{
    NullBuf         nb;
    std::streambuf *old_out;
    std::streambuf *old_err;
    Quiet()
        : old_out(std::cout.rdbuf(&nb))
        , old_err(std::cerr.rdbuf(&nb))
    {
    }
    ~Quiet()
    {
        std::cout.rdbuf(old_out);
        std::cerr.rdbuf(old_err);
    }
};

} // namespace

extern "C" {

// ---------------------------------------------------------------- attention
void ref_paged_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                         const int *block_table, float *att_scores, int num_tokens, int block_size, int head_dim,
                         int n_heads, int n_kv_heads)
{
    Attention::paged_attention(out, q, key_cache, value_cache, block_table, att_scores, num_tokens, block_size,
                               head_dim, n_heads, n_kv_heads);
}

void ref_standard_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                            float *att_scores, int pos, int head_dim, int n_heads, int n_kv_heads, int max_seq_len)
{
    Attention::standard_attention(out, q, key_cache, value_cache, att_scores, pos, head_dim, n_heads, n_kv_heads,
                                  max_seq_len);
}

// Batched convenience for the CPU baseline: one reference call per sequence,
// in order, on the calling thread (the reference is single-threaded).
// block_tables is row-padded [num_seqs, max_blocks]; q/out are [num_seqs, n_heads, head_dim].
void ref_paged_attention_batch(float *out, const float *q, const float *key_cache, const float *value_cache,
                               const int *block_tables, int max_blocks, const int *ctx_lens, int seq_begin,
                               int seq_end, float *att_scores, int block_size, int head_dim, int n_heads,
                               int n_kv_heads)
{
    for (int s = seq_begin; s < seq_end; s++) {
        Attention::paged_attention(out + (size_t)s * n_heads * head_dim, q + (size_t)s * n_heads * head_dim,
                                   key_cache, value_cache, block_tables + (size_t)s * max_blocks, att_scores,
                                   ctx_lens[s], block_size, head_dim, n_heads, n_kv_heads);
    }
}

// LlamaModelChunked::chunked_attention, experimental/chunked_prefill/model_chunked.hpp:167-224, on a
// one-layer contiguous cache [max_seq_len, n_kv_heads, head_dim]; q/out are [chunk_size, n_heads, head_dim].
void ref_chunked_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                           int chunk_size, int start_pos, int head_dim, int n_heads, int n_kv_heads, int max_seq_len)
{
    LlamaModelChunked m;
    m.config.n_layers    = 1;
    m.config.n_heads     = n_heads;
    m.config.n_kv_heads  = n_kv_heads;
    m.config.head_dim    = head_dim;
    m.config.dim         = n_heads * head_dim;
    m.config.max_seq_len = max_seq_len;
    const size_t cache   = (size_t)max_seq_len * n_kv_heads * head_dim;
    m.state.key_cache.assign(key_cache, key_cache + cache);
    m.state.value_cache.assign(value_cache, value_cache + cache);
    m.chunk_state.q_batch.assign(q, q + (size_t)chunk_size * n_heads * head_dim);
    m.chunk_state.att_batch.resize((size_t)chunk_size * n_heads * max_seq_len);
    m.chunked_attention(0, chunk_size, start_pos, out);
}

// Ops::apply_rope, include/ops/positional.hpp:13-39 (reached through core/model.hpp)
void ref_apply_rope(float *q, float *k, int pos, int head_dim, int n_heads, int n_kv_heads, float theta)
{
    Ops::apply_rope(q, k, pos, head_dim, n_heads, n_kv_heads, theta);
}

// Ops::rms_norm / Ops::matmul (include/ops/normalization.hpp:13-23, include/ops/linear.hpp:12-22) per row, and
// BatchOps::batch_rms_norm / batch_matmul (experimental/chunked_prefill/batch_ops.hpp:58-76, :8-23)
void ref_rms_norm_rows(float *out, const float *in, const float *weight, int batch_size, int dim)
{
    for (int b = 0; b < batch_size; b++)
        Ops::rms_norm(out + (size_t)b * dim, in + (size_t)b * dim, weight, dim);
}

void ref_matmul_rows(float *out, const float *in, const float *weight, int batch_size, int in_dim, int out_dim)
{
    for (int b = 0; b < batch_size; b++)
        Ops::matmul(out + (size_t)b * out_dim, in + (size_t)b * in_dim, weight, in_dim, out_dim);
}

void ref_batch_rms_norm(float *out, const float *in, const float *weight, int batch_size, int dim)
{
    BatchOps::batch_rms_norm(out, in, weight, batch_size, dim);
}

void ref_batch_matmul(float *out, const float *in, const float *weight, int batch_size, int in_dim, int out_dim)
{
    BatchOps::batch_matmul(out, in, weight, batch_size, in_dim, out_dim);
}

// ---------------------------------------------------------------- Tokenizer
// class Tokenizer, include/core/tokenizer.hpp (reached through core/runner-free includes below)
void *ref_tokenizer_create(const char *path, int vocab_size)
{
    Quiet hush;
    try {
        return new Tokenizer(path, vocab_size);
    }
    catch (const std::exception &) {
        return nullptr;
    }
}

void ref_tokenizer_destroy(void *t) { delete static_cast<Tokenizer *>(t); }

int ref_tokenizer_encode(void *t, const char *text, int text_len, int bos, int eos, int *ids, int cap)
{
    std::vector<int> v = static_cast<Tokenizer *>(t)->encode(std::string(text, (size_t)text_len), bos != 0, eos != 0);
    for (int i = 0; i < (int)v.size() && i < cap; i++)
        ids[i] = v[(size_t)i];
    return (int)v.size();
}

int ref_tokenizer_decode(void *t, int token, char *out, int cap)
{
    const std::string piece = static_cast<Tokenizer *>(t)->decode(token);
    const int         n     = (int)piece.size();
    if (cap > 0) {
        const int c = std::min(n, cap - 1);
        std::memcpy(out, piece.data(), (size_t)c);
        out[c] = '\0';
    }
    return n;
}

// ------------------------------------------------------------- BlockManager
void *ref_bm_create(int num_blocks, int block_size)
{
    Quiet hush;
    return new BlockManager(num_blocks, block_size);
}
void ref_bm_destroy(void *bm) { delete static_cast<BlockManager *>(bm); }

int ref_bm_allocate_block(void *bm)
{
    Quiet hush;
    return static_cast<BlockManager *>(bm)->allocate_block();
}

// 0 ok / -1 threw
int ref_bm_free_block(void *bm, int id)
{
    Quiet hush;
    try {
        static_cast<BlockManager *>(bm)->free_block(id);
        return 0;
    }
    catch (const std::exception &) {
        return -1;
    }
}

// returns the number of ids written, or -1 if the reference threw
int ref_bm_allocate_sequence(void *bm, int num_tokens, int *ids, int cap)
{
    Quiet hush;
    try {
        std::vector<int> v = static_cast<BlockManager *>(bm)->allocate_sequence(num_tokens);
        int              n = (int)v.size();
        for (int i = 0; i < n && i < cap; i++)
            ids[i] = v[i];
        return n;
    }
    catch (const std::exception &) {
        return -1;
    }
}

int ref_bm_free_sequence(void *bm, const int *ids, int n)
{
    Quiet hush;
    try {
        static_cast<BlockManager *>(bm)->free_sequence(std::vector<int>(ids, ids + n));
        return 0;
    }
    catch (const std::exception &) {
        return -1;
    }
}

int   ref_bm_num_free(void *bm) { return static_cast<BlockManager *>(bm)->get_num_free_blocks(); }
int   ref_bm_num_blocks(void *bm) { return static_cast<BlockManager *>(bm)->get_num_blocks(); }
int   ref_bm_block_size(void *bm) { return static_cast<BlockManager *>(bm)->get_block_size(); }
int   ref_bm_is_free(void *bm, int id) { return static_cast<BlockManager *>(bm)->is_free(id) ? 1 : 0; }
float ref_bm_utilization(void *bm) { return static_cast<BlockManager *>(bm)->get_utilization(); }

int ref_bm_allocate_for_request(void *bm, int request_id, int num_tokens, int *ids, int cap)
{
    Quiet            hush;
    std::vector<int> v = static_cast<BlockManager *>(bm)->allocate_for_request(request_id, num_tokens);
    int              n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++)
        ids[i] = v[i];
    return n;
}
int ref_bm_allocate_block_for_request(void *bm, int request_id)
{
    Quiet hush;
    return static_cast<BlockManager *>(bm)->allocate_block_for_request(request_id);
}
void ref_bm_free_request(void *bm, int request_id)
{
    Quiet hush;
    static_cast<BlockManager *>(bm)->free_request(request_id);
}
int ref_bm_get_request_blocks(void *bm, int request_id, int *ids, int cap)
{
    std::vector<int> v = static_cast<BlockManager *>(bm)->get_request_blocks(request_id);
    int              n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++)
        ids[i] = v[i];
    return n;
}
int ref_bm_get_request_block_count(void *bm, int request_id)
{
    return static_cast<BlockManager *>(bm)->get_request_block_count(request_id);
}
int ref_bm_num_active_requests(void *bm) { return static_cast<BlockManager *>(bm)->get_num_active_requests(); }
int ref_bm_allocate_block_safe(void *bm)
{
    Quiet hush;
    return static_cast<BlockManager *>(bm)->allocate_block_safe();
}
void ref_bm_free_block_safe(void *bm, int id)
{
    Quiet hush;
    static_cast<BlockManager *>(bm)->free_block_safe(id);
}

// ------------------------------------------------------------------- model
// Whole-model wrappers, used only to generate the token-stream fixture (F4/F5).
void *ref_model_create(const char *path, int use_paged, int block_size, int num_blocks)
{
    Quiet hush;
    try {
        LlamaModel *m = new LlamaModel();
        m->load(path);
        m->config.use_paged_attention = use_paged != 0;
        if (block_size > 0)
            m->config.block_size = block_size;
        if (num_blocks > 0)
            m->config.num_blocks = num_blocks;
        m->initialize_paged_attention();
        return m;
    }
    catch (const std::exception &e) {
        std::fprintf(stderr, "ref_model_create: %s\n", e.what());
        return nullptr;
    }
}
void ref_model_destroy(void *m)
{
    LlamaModel *mm = static_cast<LlamaModel *>(m);
    delete mm->block_manager;
    delete mm;
}

// 7 header ints + head_dim
void ref_model_config(void *m, int *out8)
{
    const Config &c = static_cast<LlamaModel *>(m)->config;
    out8[0]         = c.dim;
    out8[1]         = c.hidden_dim;
    out8[2]         = c.n_layers;
    out8[3]         = c.n_heads;
    out8[4]         = c.n_kv_heads;
    out8[5]         = c.vocab_size;
    out8[6]         = c.max_seq_len;
    out8[7]         = c.head_dim;
}

// 0 ok, -1 threw ("Out of memory: no free blocks")
int ref_model_forward(void *m, int token, int pos, float *logits_out)
{
    Quiet       hush;
    LlamaModel *mm = static_cast<LlamaModel *>(m);
    try {
        mm->forward(token, pos);
    }
    catch (const std::exception &) {
        return -1;
    }
    if (logits_out)
        std::memcpy(logits_out, mm->state.logits.data(), mm->state.logits.size() * sizeof(float));
    return 0;
}

int ref_model_block_table(void *m, int layer, int *ids, int cap)
{
    LlamaModel *mm = static_cast<LlamaModel *>(m);
    if (layer < 0 || layer >= (int)mm->block_tables.size())
        return -1;
    int n = (int)mm->block_tables[layer].size();
    for (int i = 0; i < n && i < cap; i++)
        ids[i] = mm->block_tables[layer][i];
    return n;
}

// last attention output of the final layer (state.xb2), for tighter pinning
void ref_model_last_attn_out(void *m, float *out, int n)
{
    LlamaModel *mm = static_cast<LlamaModel *>(m);
    std::memcpy(out, mm->state.xb2.data(), (size_t)n * sizeof(float));
}

int ref_greedy_sample(float *logits, int vocab_size)
{
    Sampler s(vocab_size, 0.0f, 0.9f, 0ULL);
    return s.sample(logits);
}

} // extern "C"
