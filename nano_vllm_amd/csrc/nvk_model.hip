// nvk_model.hip -- a llama2.c checkpoint resident on the device and its forward pass as batched steps
// (the on-disk checkpoint format; and the caller of the whole path,
// LlamaModel::forward, include/core/model.hpp:150-247).
//
// The file format is the reference's (include/core/model.hpp:118-148, 316-365): seven int32
// (dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, max_seq_len), then fp32 tensors grouped by
// parameter type -- token_embedding, rms_att[L], wq[L], wk[L], wv[L], wo[L], rms_ffn[L], w_gate[L],
// w_down[L], w_up[L], rms_final, and lm_head only if the file still has vocab*dim floats left (else the
// classifier shares the embedding table, :349-364).
//
// A step takes one token -- or a chunk of prompt tokens -- for each listed request and runs
// model.hpp:150-247 for all their rows at once, everything on the device: embedding rows ->
// per layer { nvk_batch_layer_hidden (rms_norm, q/k/v, RoPE, append, paged attention), wo, residual + rms_norm
// (one kernel), w_gate | w_up, SwiGLU, w_down, residual (fused into the next layer's attention norm) } ->
// final rms_norm of each request's last row ->
// classifier -> greedy argmax (include/core/sampler.hpp:27-29: the first maximum).
// Host code over the public C ABI of this library plus four elementwise kernels.  No CPU path.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nvk_device_scope.h"
#include "nvk_internal.h"

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return nvk::fail(e__ == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP, "%s failed: %s (%s:%d)", #expr,  \
                             hipGetErrorString(e__), __FILE__, __LINE__);                                              \
    } while (0)

namespace {

// x[r] = token_embedding_table[tokens[r]]  (model.hpp:153-154), float4 per thread
__global__ void nvk_embed_kernel(float *__restrict__ x, const float *__restrict__ table, const int *__restrict__ tokens,
                                 int dim, int vocab)
{
    const int     tok = min(max(tokens[blockIdx.x], 0), vocab - 1);
    const float4 *src = reinterpret_cast<const float4 *>(table + (int64_t)tok * dim);
    float4       *dst = reinterpret_cast<float4 *>(x + (int64_t)blockIdx.x * dim);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x)
        dst[i] = src[i];
}

// x += y  (model.hpp:227-228, 238-239)
__global__ void nvk_residual_kernel(float *__restrict__ x, const float *__restrict__ y, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4       a = reinterpret_cast<float4 *>(x)[i];
        const float4 b = reinterpret_cast<const float4 *>(y)[i];
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        reinterpret_cast<float4 *>(x)[i] = a;
    }
}

// dst[i] = src[rows[i]]: each request's last row (the only one whose logits are wanted)
__global__ void nvk_take_rows_kernel(float *__restrict__ dst, const float *__restrict__ src, const int *__restrict__ rows,
                                     int dim)
{
    const float4 *s = reinterpret_cast<const float4 *>(src + (int64_t)rows[blockIdx.x] * dim);
    float4       *d = reinterpret_cast<float4 *>(dst + (int64_t)blockIdx.x * dim);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x)
        d[i] = s[i];
}

// next[r] = index of the first maximum of logits[r]  (Sampler, include/core/sampler.hpp:27-29: std::max_element)
// One workgroup of 1024 threads per row, float4 loads (a 32000-entry row is 8 loads per thread in flight at once;
// with 256 scalar-loading threads the kernel took 41 us of pure latency).
constexpr int kArgmaxThreads = 1024;
__device__ __forceinline__ void argmax_take(float v, int i, float &best, int &at)
{
    if (v > best || (v == best && i < at)) // NaNs never win, like operator< in std::max_element
        best = v, at = i;
}
__global__ void __launch_bounds__(kArgmaxThreads) nvk_argmax_kernel(int *__restrict__ next, const float *__restrict__ logits,
                                                                    int vocab)
{
    __shared__ float s_v[kArgmaxThreads / 64];
    __shared__ int   s_i[kArgmaxThreads / 64];
    const float     *row  = logits + (int64_t)blockIdx.x * vocab;
    float            best = -INFINITY;
    int              at   = 0x7fffffff;
    const bool       vec  = vocab % 4 == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0;
    const int        n4   = vec ? vocab / 4 : 0;
    for (int i = threadIdx.x; i < n4; i += kArgmaxThreads) {
        const float4 v = reinterpret_cast<const float4 *>(row)[i];
        argmax_take(v.x, 4 * i, best, at), argmax_take(v.y, 4 * i + 1, best, at);
        argmax_take(v.z, 4 * i + 2, best, at), argmax_take(v.w, 4 * i + 3, best, at);
    }
    for (int i = 4 * n4 + threadIdx.x; i < vocab; i += kArgmaxThreads)
        argmax_take(row[i], i, best, at);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(best, o);
        const int   i = __shfl_xor(at, o);
        argmax_take(v, i, best, at);
    }
    if ((threadIdx.x & 63) == 0)
        s_v[threadIdx.x >> 6] = best, s_i[threadIdx.x >> 6] = at;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kArgmaxThreads / 64; w++)
            argmax_take(s_v[w], s_i[w], best, at);
        next[blockIdx.x] = at == 0x7fffffff ? 0 : at; // a row of -inf / NaN only: the first element, as max_element
    }
}

struct LayerWeights
{
    const float *rms_att, *wq, *wk, *wv, *wo, *rms_ffn, *w_gate, *w_down, *w_up;
};

} // namespace

struct nvk_model
{
    int  dim = 0, hidden_dim = 0, n_layers = 0, n_heads = 0, n_kv_heads = 0, vocab = 0, max_seq_len = 0, head_dim = 0;
    bool shared = false;
    int  device = 0, max_seqs = 0, max_rows = 0;
    float rope_theta = 10000.0f, rms_eps = 1e-5f; // model.hpp:40, normalization.hpp:13

    float                    *weights = nullptr; // one allocation, file order
    const float              *tok_emb = nullptr, *rms_final = nullptr, *lm_head = nullptr;
    std::vector<LayerWeights> layers;

    nvk_pool          *pool  = nullptr;
    nvk_block_manager *bm    = nullptr;
    nvk_batch         *batch = nullptr;
    hipStream_t        stream = nullptr; // the pool's

    // activations for up to max_rows rows (grown on demand) and per-request outputs
    float *x = nullptr, *xb = nullptr, *xn = nullptr, *att = nullptr, *hb = nullptr, *hb2 = nullptr;
    float *xl = nullptr, *logits = nullptr; // [max_seqs, dim], [max_seqs, vocab]
    int   *tokens_dev = nullptr, *last_rows_dev = nullptr, *next_dev = nullptr;
    int   *pinned = nullptr; // tokens | last rows | next tokens
    size_t pinned_ints = 0;

    // The body of a pure decode step (embedding .. sampler, ~12 launches per layer) captured as a hipGraph and replayed
    // while nothing it bakes in has changed: with few requests a step is hundreds of short dependent kernels and
    // the launch gaps, not the kernels, set its time.
    bool            use_graphs = true;
    bool            use_linked = false; // opt-in (nvk_model_use_linked_norms): the norms of 1-16 row steps ride between the
                                        // GEMVs (nvk::NormLink) -- faster, a few ulp from the separate norm kernels
    int             graph_failures = 0;
    hipGraphExec_t  graph_exec = nullptr;
    long long       graph_sig[9] = {0};
    long long       graph_seen[9] = {0}; // the last pure decode step's shape, recorded or not
};

namespace {

void release(nvk_model *m)
{
    if (!m)
        return;
    nvk::DeviceScope scope(m->device); // (a release path: goes on whatever the switch returned)
    (void)hipDeviceSynchronize();
    if (m->graph_exec)
        (void)hipGraphExecDestroy(m->graph_exec);
    if (m->batch)
        (void)nvk_batch_destroy(m->batch);
    if (m->pool)
        (void)nvk_pool_destroy(m->pool);
    if (m->bm)
        nvk_bm_destroy(m->bm);
    for (float *p : {m->weights, m->x, m->xb, m->xn, m->att, m->hb, m->hb2, m->xl, m->logits})
        if (p)
            (void)hipFree(p);
    for (int *p : {m->tokens_dev, m->last_rows_dev, m->next_dev})
        if (p)
            (void)hipFree(p);
    if (m->pinned)
        (void)hipHostFree(m->pinned);
    delete m;
}

int reserve_rows(nvk_model *m, int rows)
{
    if (rows <= m->max_rows)
        return NVK_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (m->graph_exec) { // it holds the old buffers' addresses
        (void)hipGraphExecDestroy(m->graph_exec);
        m->graph_exec = nullptr;
    }
    for (float **p : {&m->x, &m->xb, &m->xn, &m->att, &m->hb, &m->hb2})
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    if (m->tokens_dev)
        (void)hipFree(m->tokens_dev);
    if (m->pinned)
        (void)hipHostFree(m->pinned);
    m->tokens_dev = nullptr, m->pinned = nullptr, m->max_rows = 0;
    const size_t r = (size_t)std::max(rows, m->max_seqs);
    HIP_TRY(hipMalloc((void **)&m->x, r * m->dim * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&m->xb, r * m->dim * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&m->xn, r * m->dim * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&m->att, r * (size_t)m->n_heads * m->head_dim * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&m->hb, r * m->hidden_dim * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&m->hb2, r * m->hidden_dim * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&m->tokens_dev, r * sizeof(int)));
    m->pinned_ints = r + 2 * (size_t)m->max_seqs;
    HIP_TRY(hipHostMalloc((void **)&m->pinned, m->pinned_ints * sizeof(int), hipHostMallocDefault));
    m->max_rows = (int)r;
    return NVK_OK;
}

} // namespace

extern "C" {

int nvk_model_load(const char *checkpoint_path, int block_size, int num_blocks, int max_seqs, int device,
                   nvk_model **out_model)
{
    if (!out_model)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: out_model is null");
    *out_model = nullptr;
    if (!checkpoint_path || block_size <= 0 || num_blocks <= 0 || max_seqs <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: bad arguments");
    FILE *f = fopen(checkpoint_path, "rb");
    if (!f)
        return nvk::fail(NVK_ERR_INVALID_ARG, "Failed to open model file: %s", checkpoint_path); // model.hpp:123
    int hdr[7];
    if (fread(hdr, sizeof(int), 7, f) != 7) {
        fclose(f);
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: %s is shorter than its header", checkpoint_path);
    }
    const int D = hdr[0], Hd = hdr[1], L = hdr[2], H = hdr[3], KV = hdr[4], V = hdr[5], S = hdr[6];
    if (D <= 0 || Hd <= 0 || L <= 0 || H <= 0 || KV <= 0 || V <= 0 || S <= 0 || D % H != 0 || H % KV != 0) {
        fclose(f);
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: implausible header (dim %d, hidden %d, layers %d, heads %d/%d, "
                                              "vocab %d, max_seq_len %d)", D, Hd, L, H, KV, V, S);
    }
    const int hd = D / H; // model.hpp:129
    const size_t QD = (size_t)H * hd, KD = (size_t)KV * hd;
    if (D % 4 != 0 || Hd % 4 != 0 || QD % 4 != 0) {
        fclose(f);
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: dim, hidden_dim and n_heads*head_dim must be multiples of 4");
    }
    // tensor sizes in file order (model.hpp:322-347)
    const size_t sizes[11] = {(size_t)V * D, (size_t)L * D,      (size_t)L * QD * D, (size_t)L * KD * D,
                              (size_t)L * KD * D, (size_t)L * D * QD, (size_t)L * D, (size_t)L * Hd * D,
                              (size_t)L * D * Hd, (size_t)L * Hd * D, (size_t)D};
    size_t body = 0;
    for (size_t s : sizes)
        body += s;
    fseek(f, 0, SEEK_END);
    const long long file_floats = ((long long)ftell(f) - 7 * (long long)sizeof(int)) / (long long)sizeof(float);
    if (file_floats < (long long)body) {
        fclose(f);
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: %s holds %lld floats, the header asks for %zu",
                         checkpoint_path, file_floats, body);
    }
    const bool   has_head = file_floats - (long long)body >= (long long)V * D; // model.hpp:356
    const size_t total    = body + (has_head ? (size_t)V * D : 0);
    fseek(f, 7 * sizeof(int), SEEK_SET);

    int rc = NVK_OK;
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            fclose(f);
            return nvk::fail(NVK_ERR_NO_DEVICE, "nvk_model_load: no HIP device visible (this library has no CPU path)");
        }
        if (device < 0 || device >= ndev) {
            fclose(f);
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: device %d outside [0,%d)", device, ndev);
        }
    }
    nvk_model *m = new (std::nothrow) nvk_model();
    if (!m) {
        fclose(f);
        return nvk::fail(NVK_ERR_OOM, "nvk_model_load: host allocation failed");
    }
    m->dim = D, m->hidden_dim = Hd, m->n_layers = L, m->n_heads = H, m->n_kv_heads = KV, m->vocab = V, m->max_seq_len = S;
    m->head_dim = hd, m->shared = !has_head, m->device = device, m->max_seqs = max_seqs;
    NVK_SCOPE(device);

    // stream the file to the device in 64 MiB pieces through one pinned buffer
    hipError_t e = hipMalloc((void **)&m->weights, total * sizeof(float));
    float     *stage = nullptr;
    const size_t piece = (size_t)16 << 20; // floats
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&stage, std::min(piece, total) * sizeof(float), hipHostMallocDefault);
    for (size_t at = 0; e == hipSuccess && at < total; at += piece) {
        const size_t n = std::min(piece, total - at);
        if (fread(stage, sizeof(float), n, f) != n) {
            rc = nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_load: short read in %s", checkpoint_path);
            break;
        }
        e = hipMemcpy(m->weights + at, stage, n * sizeof(float), hipMemcpyHostToDevice);
    }
    fclose(f);
    if (stage)
        (void)hipHostFree(stage);
    if (e != hipSuccess)
        rc = nvk::fail(e == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP, "nvk_model_load: %s", hipGetErrorString(e));
    if (rc != NVK_OK) {
        release(m);
        return rc;
    }
    const float *p[12];
    size_t       at = 0;
    for (int i = 0; i < 11; i++) {
        p[i] = m->weights + at;
        at += sizes[i];
    }
    p[11]        = has_head ? m->weights + at : p[0];
    m->tok_emb   = p[0];
    m->rms_final = p[10];
    m->lm_head   = p[11];
    m->layers.resize((size_t)L);
    for (int l = 0; l < L; l++) {
        LayerWeights &w = m->layers[(size_t)l];
        w.rms_att = p[1] + (size_t)l * D, w.wq = p[2] + (size_t)l * QD * D, w.wk = p[3] + (size_t)l * KD * D;
        w.wv = p[4] + (size_t)l * KD * D, w.wo = p[5] + (size_t)l * D * QD, w.rms_ffn = p[6] + (size_t)l * D;
        w.w_gate = p[7] + (size_t)l * Hd * D, w.w_down = p[8] + (size_t)l * D * Hd, w.w_up = p[9] + (size_t)l * Hd * D;
    }

    // the paged KV pool, its allocator and the batch (initialize_paged_attention, model.hpp:249-277)
    rc = nvk_pool_create(L, num_blocks, block_size, KV, hd, device, &m->pool);
    if (rc == NVK_OK)
        rc = nvk_bm_create(num_blocks, block_size, &m->bm);
    if (rc == NVK_OK)
        rc = nvk_batch_create(m->pool, m->bm, max_seqs, (S + block_size - 1) / block_size, &m->batch);
    void *st = nullptr;
    if (rc == NVK_OK)
        rc = nvk_pool_stream(m->pool, &st);
    m->stream = (hipStream_t)st;
    if (rc == NVK_OK) {
        e = hipMalloc((void **)&m->xl, (size_t)max_seqs * D * sizeof(float));
        if (e == hipSuccess)
            e = hipMalloc((void **)&m->logits, (size_t)max_seqs * V * sizeof(float));
        if (e == hipSuccess)
            e = hipMalloc((void **)&m->last_rows_dev, (size_t)max_seqs * sizeof(int));
        if (e == hipSuccess)
            e = hipMalloc((void **)&m->next_dev, (size_t)max_seqs * sizeof(int));
        if (e != hipSuccess)
            rc = nvk::fail(e == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP, "nvk_model_load: %s", hipGetErrorString(e));
    }
    if (rc == NVK_OK)
        rc = reserve_rows(m, max_seqs);
    if (rc != NVK_OK) {
        release(m);
        return rc;
    }
    *out_model = m;
    return NVK_OK;
}

int nvk_model_destroy(nvk_model *model)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    release(model);
    return NVK_OK;
}

int nvk_model_config(const nvk_model *m, int *out8)
{
    if (!m || !out8)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_config: null argument");
    out8[0] = m->dim, out8[1] = m->hidden_dim, out8[2] = m->n_layers, out8[3] = m->n_heads, out8[4] = m->n_kv_heads;
    out8[5] = m->vocab, out8[6] = m->max_seq_len, out8[7] = m->shared ? 1 : 0;
    return NVK_OK;
}

int nvk_model_add_request(nvk_model *m, int request_id)
{
    if (!m)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_add_request: null model");
    return nvk_batch_add_request(m->batch, request_id);
}

int nvk_model_finish_request(nvk_model *m, int request_id)
{
    if (!m)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_finish_request: null model");
    return nvk_batch_finish_request(m->batch, request_id);
}

int nvk_model_request_len(const nvk_model *m, int request_id)
{
    if (!m)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_request_len: null model");
    return nvk_batch_request_len(m->batch, request_id);
}

int nvk_model_use_linked_norms(nvk_model *m, int on)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!m)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_use_linked_norms: null model");
    m->use_linked = on != 0;
    if (m->graph_exec) { // recorded with the other sequence of kernels
        (void)hipGraphExecDestroy(m->graph_exec);
        m->graph_exec = nullptr;
    }
    return NVK_OK;
}

int nvk_model_use_graphs(nvk_model *m, int on)
{
    if (!m)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_use_graphs: null model");
    m->use_graphs     = on != 0;
    m->graph_failures = 0;
    if (!on && m->graph_exec) {
        NVK_SCOPE(m->device);
        (void)hipStreamSynchronize(m->stream);
        (void)hipGraphExecDestroy(m->graph_exec);
        m->graph_exec = nullptr;
    }
    return NVK_OK;
}

int nvk_model_free_blocks(const nvk_model *m) { return m ? nvk_bm_num_free_blocks(m->bm) : 0; }

int nvk_model_step_chunks(nvk_model *m, const int *request_ids, const int *chunk_lens, const int *tokens, int n,
                          int *next_tokens, float *logits)
{
    if (!m || n < 0 || (n > 0 && (!request_ids || !chunk_lens || !tokens)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_step_chunks: bad arguments");
    if (n == 0)
        return NVK_OK;
    if (n > m->max_seqs)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_step_chunks: %d requests, %d slots", n, m->max_seqs);
    long long rows = 0;
    for (int i = 0; i < n; i++) {
        if (chunk_lens[i] < 1)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_step_chunks: request %d brings %d tokens", request_ids[i],
                             chunk_lens[i]);
        const int len = nvk_batch_request_len(m->batch, request_ids[i]);
        if (len < 0)
            return len;
        if ((long long)len + chunk_lens[i] > m->max_seq_len)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_step_chunks: request %d would reach %lld tokens, max_seq_len is %d",
                             request_ids[i], (long long)len + chunk_lens[i], m->max_seq_len);
        rows += chunk_lens[i];
    }
    if (rows > 0x3fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_step_chunks: %lld rows in one step", rows);
    NVK_SCOPE(m->device);
    int         rc = reserve_rows(m, (int)rows);
    if (rc != NVK_OK)
        return rc;
    const int   R = (int)rows, D = m->dim, Hd = m->hidden_dim, V = m->vocab, QD = m->n_heads * m->head_dim;
    hipStream_t st = m->stream;

    // the step's rows follow the batch's order: decode (one-token) requests must come first
    int *h_tok = m->pinned, *h_last = m->pinned + m->max_rows, *h_next = h_last + m->max_seqs;
    HIP_TRY(hipStreamSynchronize(st)); // the pinned buffer is ours
    int r = 0;
    for (int i = 0; i < n; i++) {
        r += chunk_lens[i];
        h_last[i] = r - 1;
    }
    std::memcpy(h_tok, tokens, (size_t)R * sizeof(int));
    rc = nvk_batch_begin_step_chunks(m->batch, request_ids, chunk_lens, n, m->n_heads, (void *)st);
    if (rc != NVK_OK)
        return rc;
    // a failure inside the step: no request advances; K/V rows it already wrote are overwritten by the next step
    auto fail_step = [&](int code) {
        (void)hipStreamSynchronize(st);
        (void)nvk_batch_abort_step(m->batch);
        return code;
    };
    // everything between the batch's bookkeeping and the host's wait: tokens in, embedding .. sampler, next tokens out
    // Up to 16 decoding rows (the reference's own case is one): 6 launches per layer instead of 10.  The residual adds sit in the
    // epilogues of the wo and w_down GEMVs, which also leave x * gain and the partial sums of squares for the next
    // norm; the GEMV that consumes them multiplies its product by 1 / rms (nvk::NormLink).  xn holds x * gain, xb the
    // partial sums.  Layer 0's first norm has no producer and stays a kernel.
    const bool linked = m->use_linked && R == n && nvk::linked_gemv_ok() && m->rope_theta > 0.f && m->head_dim % 2 == 0
                        && D % 4 == 0 && Hd % 4 == 0 && QD % 4 == 0 && (R <= 2 || (V % 4 == 0 && m->head_dim % 4 == 0 && D % 16 == 0))
                        && R <= nvk::kFusedRowsMax && nvk::qkv_rope_append_ok(R, D, QD, m->n_kv_heads * m->head_dim, m->device);
    auto enqueue_linked = [&]() -> int {
        hipError_t e1 = hipMemcpyAsync(m->tokens_dev, h_tok, (size_t)R * sizeof(int), hipMemcpyHostToDevice, st);
        if (e1 != hipSuccess)
            return nvk::fail(NVK_ERR_HIP, "nvk_model_step_chunks: staging the tokens failed: %s", hipGetErrorString(e1));
        hipLaunchKernelGGL(nvk_embed_kernel, dim3((unsigned)R), dim3(64), 0, st, m->x, m->tok_emb, m->tokens_dev, D, V);
        const int tiles_d = nvk::linked_ss_n(R, D);
        auto      produce = [&](const float *gain) { // x += in . w^T; xn = x * gain; xb = partial sums of x^2
            nvk::NormLink l;
            l.gain_out = gain, l.xg_out = m->xn, l.ss_out = m->xb, l.eps = m->rms_eps;
            return l;
        };
        nvk::NormLink consume; // ... / rms(x), x as the last producer left it
        consume.ss_in = m->xb, consume.ss_n = tiles_d, consume.eps = m->rms_eps;
        int r2 = NVK_OK;
        for (int l = 0; l < m->n_layers && r2 == NVK_OK; l++) {
            const LayerWeights &w = m->layers[(size_t)l];
            if (l == 0) // model.hpp:161-221
                r2 = nvk_batch_layer_hidden(m->batch, l, m->x, D, w.rms_att, w.wq, w.wk, w.wv, m->rms_eps, m->att,
                                            m->rope_theta, (void *)st);
            else
                r2 = nvk::batch_layer_hidden_linked(m->batch, l, m->xn, D, consume, w.wq, w.wk, w.wv, m->att, m->rope_theta,
                                                    (void *)st);
            if (r2 == NVK_OK) // :224, :227-228 and the operands of :231
                r2 = nvk::linked_gemv(produce(w.rms_ffn), false, m->x, m->att, w.wo, nullptr, R, QD, D, m->device, (void *)st);
            if (r2 == NVK_OK) // :231-234
                r2 = nvk::linked_gemv(consume, true, m->hb, m->xn, w.w_gate, w.w_up, R, D, Hd, m->device, (void *)st, m->hb2);
            if (r2 == NVK_OK) // :235, :238-239 and the operands of the next norm (the next layer's :161, or :243)
                r2 = nvk::linked_gemv(produce(l + 1 < m->n_layers ? m->layers[(size_t)l + 1].rms_att : m->rms_final), false,
                                      m->x, m->hb, w.w_down, nullptr, R, Hd, D, m->device, (void *)st);
        }
        if (r2 == NVK_OK) // :243-246
            r2 = nvk::linked_gemv(consume, false, m->logits, m->xn, m->lm_head, nullptr, R, D, V, m->device, (void *)st);
        if (r2 != NVK_OK)
            return r2;
        hipLaunchKernelGGL(nvk_argmax_kernel, dim3((unsigned)n), dim3(kArgmaxThreads), 0, st, m->next_dev, m->logits, V);
        hipError_t e2 = hipGetLastError();
        if (e2 == hipSuccess)
            e2 = hipMemcpyAsync(h_next, m->next_dev, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e2 != hipSuccess)
            return nvk::fail(NVK_ERR_HIP, "nvk_model_step_chunks: %s", hipGetErrorString(e2));
        return NVK_OK;
    };
    auto enqueue_plain = [&]() -> int {
        hipError_t e1 = hipMemcpyAsync(m->tokens_dev, h_tok, (size_t)R * sizeof(int), hipMemcpyHostToDevice, st);
        if (e1 == hipSuccess)
            e1 = hipMemcpyAsync(m->last_rows_dev, h_last, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st);
        if (e1 != hipSuccess)
            return nvk::fail(NVK_ERR_HIP, "nvk_model_step_chunks: staging the tokens failed: %s", hipGetErrorString(e1));
        hipLaunchKernelGGL(nvk_embed_kernel, dim3((unsigned)R), dim3(64), 0, st, m->x, m->tok_emb, m->tokens_dev, D, V);
        const unsigned ew_blocks = 1024;
        // The residual adds (model.hpp:227-228, 238-239) ride in the norm that follows them: after layer l's w_down the
        // sum x += xb and layer l + 1's attention norm are one kernel (its output feeds nvk_batch_layer_hidden already
        // normalised); only the last layer's add stands alone.
        const float *attn_in = m->x; // layer 0 normalises x itself
        int          r2      = NVK_OK;
        for (int l = 0; l < m->n_layers && r2 == NVK_OK; l++) {
            const LayerWeights &w = m->layers[(size_t)l];
            r2 = nvk_batch_layer_hidden(m->batch, l, attn_in, D, l == 0 ? w.rms_att : nullptr, w.wq, w.wk, w.wv, m->rms_eps,
                                        m->att, m->rope_theta, (void *)st); // model.hpp:161-221
            if (r2 == NVK_OK)
                r2 = nvk_batch_matmul_device(m->xb, m->att, w.wo, R, QD, D, m->device, (void *)st); // :224
            if (r2 == NVK_OK) // :227-231: x += xb; xn = rms_norm(x, rms_ffn)
                r2 = nvk_batch_add_rms_norm_device(m->xn, m->x, m->xb, w.rms_ffn, R, D, m->rms_eps, m->device, (void *)st);
            if (r2 == NVK_OK) // :232-234: hb = silu(w_gate . xn) * (w_up . xn)
                r2 = nvk_batch_swiglu_ffn_device(m->hb, m->hb2, m->xn, w.w_gate, w.w_up, R, D, Hd, m->device, (void *)st);
            if (r2 != NVK_OK)
                break;
            r2 = nvk_batch_matmul_device(m->xb, m->hb, w.w_down, R, Hd, D, m->device, (void *)st);                    // :235
            if (r2 != NVK_OK)
                break;
            if (l + 1 < m->n_layers) { // :238-239 + the next layer's :161
                r2 = nvk_batch_add_rms_norm_device(m->xn, m->x, m->xb, m->layers[(size_t)l + 1].rms_att, R, D, m->rms_eps,
                                                   m->device, (void *)st);
                attn_in = m->xn;
            }
            else if (R == n) // every row is a request's last row: the last add rides in the final norm (:238-239, :243)
                r2 = nvk_batch_add_rms_norm_device(m->xl, m->x, m->xb, m->rms_final, n, D, m->rms_eps, m->device, (void *)st);
            else
                hipLaunchKernelGGL(nvk_residual_kernel, dim3(ew_blocks), dim3(256), 0, st, m->x, m->xb, (int64_t)R * D / 4);
        }
        if (r2 != NVK_OK)
            return r2;
        // final norm + classifier for each request's last row only (model.hpp:243-246), then the greedy sampler
        if (R != n) {
            hipLaunchKernelGGL(nvk_take_rows_kernel, dim3((unsigned)n), dim3(64), 0, st, m->xl, m->x, m->last_rows_dev, D);
            r2 = nvk_batch_rms_norm_device(m->xl, m->xl, m->rms_final, n, D, m->rms_eps, m->device, (void *)st);
        }
        if (r2 == NVK_OK)
            r2 = nvk_batch_matmul_device(m->logits, m->xl, m->lm_head, n, D, V, m->device, (void *)st);
        if (r2 != NVK_OK)
            return r2;
        hipLaunchKernelGGL(nvk_argmax_kernel, dim3((unsigned)n), dim3(kArgmaxThreads), 0, st, m->next_dev, m->logits, V);
        hipError_t e2 = hipGetLastError();
        if (e2 == hipSuccess)
            e2 = hipMemcpyAsync(h_next, m->next_dev, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e2 != hipSuccess)
            return nvk::fail(NVK_ERR_HIP, "nvk_model_step_chunks: %s", hipGetErrorString(e2));
        return NVK_OK;
    };

    auto enqueue_body = [&]() -> int { return linked ? enqueue_linked() : enqueue_plain(); };

    bool replayed = false;
    if (m->use_graphs && R == n) { // a pure decode step
        long long sig[9];
        sig[8] = R;
        if (nvk_batch_graph_signature(m->batch, sig) == NVK_OK) {
            const bool stale = !m->graph_exec || std::memcmp(sig, m->graph_sig, sizeof sig) != 0;
            if (stale && m->graph_exec) {
                (void)hipGraphExecDestroy(m->graph_exec);
                m->graph_exec = nullptr;
            }
            // A shape is recorded the second time it is seen in a row: its first step runs directly, so that the
            // first-use allocations inside the entry points (scratch rows, the rope table) happen outside a capture.
            const bool seen = std::memcmp(sig, m->graph_seen, sizeof sig) == 0;
            std::memcpy(m->graph_seen, sig, sizeof sig);
            if (stale && seen) {
                // record the body (nothing runs while capturing); any refusal -- a first-use allocation inside an entry
                // point, say -- ends the capture and this step runs directly
                hipGraph_t graph = nullptr;
                // (nobody else of the library on the device meanwhile: nvk_device_scope.h, the recording gate)
                nvk::RecordAlone alone;
                if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const int  rb = enqueue_body();
                    hipError_t ec = hipStreamEndCapture(st, &graph);
                    if (rb == NVK_OK && ec == hipSuccess && graph
                        && hipGraphInstantiate(&m->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess)
                    {
                        std::memcpy(m->graph_sig, sig, sizeof sig);
                        if (std::getenv("NVK_GRAPH_DEBUG"))
                            std::fprintf(stderr, "nvk_model: recorded a %d-request decode step\n", n);
                    }
                    else {
                        if (std::getenv("NVK_GRAPH_DEBUG"))
                            std::fprintf(stderr, "nvk_model: capture refused (body %d: %s; end %s)\n", rb,
                                         rb == NVK_OK ? "ok" : nvk_last_error(), hipGetErrorString(ec));
                        m->graph_exec = nullptr;
                        (void)hipGetLastError(); // clear the sticky capture error
                        if (++m->graph_failures >= 4)
                            m->use_graphs = false;
                        // a recording invalidated from outside the library leaves the stream unusable: take a fresh one
                        if (nvk::pool_renew_stream_if_stuck(m->pool) == NVK_OK) {
                            void *fresh = nullptr;
                            if (nvk_pool_stream(m->pool, &fresh) == NVK_OK && fresh != (void *)st) {
                                st = m->stream = (hipStream_t)fresh;
                                if (std::getenv("NVK_GRAPH_DEBUG"))
                                    std::fprintf(stderr, "nvk_model: the recording was invalidated from outside; new stream\n");
                            }
                        }
                    }
                    if (graph)
                        (void)hipGraphDestroy(graph);
                }
            }
            if (m->graph_exec) {
                if (hipGraphLaunch(m->graph_exec, st) == hipSuccess)
                    replayed = true;
                else {
                    (void)hipGetLastError();
                    (void)hipGraphExecDestroy(m->graph_exec);
                    m->graph_exec = nullptr;
                    m->use_graphs = false;
                }
            }
        }
    }
    if (!replayed) {
        rc = enqueue_body();
        if (rc != NVK_OK)
            return fail_step(rc);
    }
    hipError_t e = hipSuccess;
    if (logits)
        e = hipMemcpyAsync(logits, m->logits, (size_t)n * V * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return fail_step(nvk::fail(NVK_ERR_HIP, "nvk_model_step_chunks: %s", hipGetErrorString(e)));
    rc = nvk_batch_end_step(m->batch);
    if (rc != NVK_OK)
        return rc;
    if (next_tokens)
        std::memcpy(next_tokens, h_next, (size_t)n * sizeof(int));
    return NVK_OK;
}

int nvk_model_step(nvk_model *m, const int *request_ids, const int *tokens, int n, int *next_tokens, float *logits)
{
    if (!m || n < 0 || n > 1 << 20)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_step: bad arguments");
    std::vector<int> ones((size_t)n, 1);
    return nvk_model_step_chunks(m, request_ids, ones.data(), tokens, n, next_tokens, logits);
}

int nvk_model_forward(nvk_model *m, int token, int pos, float *logits)
{
    if (!m)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_forward: null model");
    // LlamaModel::forward(token, pos): one sequence, request 0, created on first use; pos is its length
    int len = nvk_batch_request_len(m->batch, 0);
    if (len < 0) {
        int rc = nvk_batch_add_request(m->batch, 0);
        if (rc != NVK_OK)
            return rc;
        len = 0;
    }
    if (pos != len)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_forward: pos %d, the sequence holds %d tokens", pos, len);
    const int id = 0, one = 1;
    return nvk_model_step_chunks(m, &id, &one, &token, 1, nullptr, logits);
}

} // extern "C"
