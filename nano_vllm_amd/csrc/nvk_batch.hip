// nvk_batch.hip -- the decode batch: per-request block tables kept on the device
// and one append + attention call per layer for every DECODING request
// (SURVEY.md 8f row f2).
//
// The reference's scheduler already forms ScheduledBatch::decode_requests
// (include/scheduler/scheduler.hpp:24-43,66-77) and Request carries block_ids
// (include/scheduler/request.hpp:74), but nothing executes them
// (include/scheduler/batched_runner.hpp:84-85): every request runs alone through
// LlamaModel::forward, whose block tables belong to the model, not to the request
// (include/core/model.hpp:113,181-187).  This object is the missing piece: it owns,
// per running request, the block list the per-request BlockManager API hands out
// (include/scheduler/block_manager.hpp:145-171), mirrors the step's tables,
// positions and context lengths in HBM, and drives the kernels of
// nvk_paged_attn.hip through the public C ABI only.
//
// One block id names the same slot in every layer's slab (tables are shared by the
// layers of a step, which is what lets one decode plan serve all of them); the
// reference's model instead draws a separate block per layer from the shared list.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <unordered_map>
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

// Same request list as the previous step: every sequence moves on by one token and the
// freshly allocated blocks (row, column, id) are written into the resident tables.
__global__ void nvk_batch_advance_kernel(int *__restrict__ tables, int max_blocks, int *__restrict__ positions,
                                         int *__restrict__ ctx_lens, int n, const int *__restrict__ patches,
                                         int n_patches)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        positions[i] += 1;
        ctx_lens[i] += 1;
    }
    if (i < n_patches)
        tables[(int64_t)patches[3 * i] * max_blocks + patches[3 * i + 1]] = patches[3 * i + 2];
}

struct Slot
{
    int              request_id = -1;
    int              len        = 0; // tokens whose K/V are in the pool
    std::vector<int> blocks;         // = Request::block_ids
};

} // namespace

struct nvk_batch
{
    nvk_pool          *pool = nullptr;
    nvk_block_manager *bm   = nullptr;
    int                max_seqs = 0, max_blocks = 0;
    int                block_size = 0, device = 0;

    std::vector<Slot>            slots; // dense; a finished request's slot is taken by the last one
    std::unordered_map<int, int> slot_of;

    int row_len = 0, head_dim = 0; // floats per K/V token row, per head

    // the step in flight
    bool             in_step = false;
    int              n_heads = 0;
    // a step that brings chunks (prefill) as well: the first n_decode requests take one token each, the rest
    // take step_chunks[i] tokens at positions step_starts[i].. (host copies, what the prefill entry points want)
    bool             chunked  = false;
    int              n_decode = 0;
    std::vector<int> step_chunks, step_starts;
    std::vector<int> step_ids, resident_ids; // resident_ids: the list whose tables/positions are in HBM now
    bool             resident_valid = false;

    // pinned host staging and its device mirrors
    int *h_tables = nullptr, *h_pos = nullptr, *h_ctx = nullptr, *h_patches = nullptr;
    int *d_tables = nullptr, *d_pos = nullptr, *d_ctx = nullptr, *d_patches = nullptr;
    hipEvent_t staged = nullptr; // the last copies out of the pinned buffers have finished
    nvk_plan  *plan   = nullptr;

    // q / k / v / normalised-x rows of nvk_batch_layer_hidden (grown on demand, kept)
    float *h_q = nullptr, *h_k = nullptr, *h_v = nullptr, *h_xb = nullptr;
    size_t h_rows = 0, h_dim = 0, h_qdim = 0;
    long long hidden_generation = 0; // bumped whenever the scratch rows move

    // counters for tests and tuning: how the tables reached the device
    long long full_uploads = 0, patch_steps = 0;
};

namespace {

hipStream_t stream_or_pool(nvk_batch *b, void *stream)
{
    if (stream)
        return (hipStream_t)stream;
    void *s = nullptr;
    (void)nvk_pool_stream(b->pool, &s);
    return (hipStream_t)s;
}

void release(nvk_batch *b)
{
    if (!b)
        return;
    nvk::DeviceScope scope(b->device); // (a release path: goes on whatever the switch returned)
    (void)hipDeviceSynchronize();
    if (b->plan)
        (void)nvk_plan_destroy(b->plan);
    for (int *p : {b->h_tables, b->h_pos, b->h_ctx, b->h_patches})
        if (p)
            (void)hipHostFree(p);
    for (int *p : {b->d_tables, b->d_pos, b->d_ctx, b->d_patches})
        if (p)
            (void)hipFree(p);
    for (float *p : {b->h_q, b->h_k, b->h_v, b->h_xb})
        if (p)
            (void)hipFree(p);
    if (b->staged)
        (void)hipEventDestroy(b->staged);
    delete b;
}

} // namespace

extern "C" {

int nvk_batch_create(nvk_pool *pool, nvk_block_manager *bm, int max_seqs, int max_blocks_per_seq, nvk_batch **out_batch)
{
    if (!out_batch)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_create: out_batch is null");
    *out_batch = nullptr;
    if (!pool || !bm || max_seqs <= 0 || max_blocks_per_seq <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_create: bad arguments");
    int geo[6];
    int rc = nvk_pool_geometry(pool, geo);
    if (rc != NVK_OK)
        return rc;
    if (nvk_bm_block_size(bm) != geo[2] || nvk_bm_num_blocks(bm) > geo[1])
        return nvk::fail(NVK_ERR_INVALID_ARG,
                         "nvk_batch_create: block manager (%d blocks of %d) does not fit the pool (%d blocks of %d)",
                         nvk_bm_num_blocks(bm), nvk_bm_block_size(bm), geo[1], geo[2]);
    if ((long long)max_seqs * max_blocks_per_seq > 0x3fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_create: max_seqs x max_blocks_per_seq too large");

    nvk_batch *b = new (std::nothrow) nvk_batch();
    if (!b)
        return nvk::fail(NVK_ERR_OOM, "nvk_batch_create: host allocation failed");
    b->pool = pool, b->bm = bm, b->max_seqs = max_seqs, b->max_blocks = max_blocks_per_seq;
    b->block_size = geo[2], b->device = geo[5];
    b->row_len = geo[3] * geo[4], b->head_dim = geo[4];
    NVK_SCOPE(b->device);
    const size_t tab = (size_t)max_seqs * max_blocks_per_seq * sizeof(int);
    const size_t vec = (size_t)max_seqs * sizeof(int);
    hipError_t   e   = hipHostMalloc((void **)&b->h_tables, tab, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&b->h_pos, vec, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&b->h_ctx, vec, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&b->h_patches, 3 * vec, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->d_tables, tab);
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->d_pos, vec);
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->d_ctx, vec);
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->d_patches, 3 * vec);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&b->staged, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipMemset(b->d_tables, 0, tab);
    if (e == hipSuccess)
        e = hipDeviceSynchronize(); // (the fill may still be in flight, and the batch's streams do not wait for the null stream)
    if (e != hipSuccess) {
        const int code = nvk::fail(e == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP, "nvk_batch_create: %s",
                                   hipGetErrorString(e));
        release(b);
        return code;
    }
    rc = nvk_plan_create(pool, &b->plan);
    if (rc != NVK_OK) {
        release(b);
        return rc;
    }
    *out_batch = b;
    return NVK_OK;
}

int nvk_batch_destroy(nvk_batch *batch)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    release(batch);
    return NVK_OK;
}

int nvk_batch_add_request(nvk_batch *b, int request_id)
{
    if (!b)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_add_request: null batch");
    if (b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_add_request: a step is in flight");
    if (b->slot_of.count(request_id))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_add_request: request %d is already running", request_id);
    if ((int)b->slots.size() >= b->max_seqs)
        return nvk::fail(NVK_ERR_OOM, "nvk_batch_add_request: all %d sequence slots are taken", b->max_seqs);
    Slot s;
    s.request_id           = request_id;
    b->slot_of[request_id] = (int)b->slots.size();
    b->slots.push_back(std::move(s));
    return NVK_OK;
}

int nvk_batch_finish_request(nvk_batch *b, int request_id)
{
    if (!b)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_finish_request: null batch");
    if (b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_finish_request: a step is in flight");
    auto it = b->slot_of.find(request_id);
    if (it == b->slot_of.end())
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_finish_request: request %d is not running", request_id);
    // Kernels of earlier steps may still be reading this request's blocks: they must be done
    // before the ids can be handed to another request.
    {
        NVK_SCOPE(b->device);
        HIP_TRY(hipDeviceSynchronize());
    }
    nvk_bm_free_request(b->bm, request_id); // ref: BlockManager::free_request, block_manager.hpp:157-171
    const int slot = it->second;
    b->slot_of.erase(it);
    if (slot != (int)b->slots.size() - 1) {
        b->slots[(size_t)slot]                    = std::move(b->slots.back());
        b->slot_of[b->slots[(size_t)slot].request_id] = slot;
    }
    b->slots.pop_back();
    b->resident_valid = false;
    return NVK_OK;
}

int nvk_batch_num_requests(const nvk_batch *b) { return b ? (int)b->slots.size() : 0; }

int nvk_batch_request_len(const nvk_batch *b, int request_id)
{
    if (!b)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_request_len: null batch");
    auto it = b->slot_of.find(request_id);
    if (it == b->slot_of.end())
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_request_len: request %d is not running", request_id);
    return b->slots[(size_t)it->second].len;
}

int nvk_batch_request_blocks(const nvk_batch *b, int request_id, int *ids, int cap)
{
    if (!b || cap < 0 || (cap > 0 && !ids))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_request_blocks: bad arguments");
    auto it = b->slot_of.find(request_id);
    if (it == b->slot_of.end())
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_request_blocks: request %d is not running", request_id);
    const std::vector<int> &v = b->slots[(size_t)it->second].blocks;
    for (int i = 0; i < (int)v.size() && i < cap; i++)
        ids[i] = v[(size_t)i];
    return (int)v.size();
}

int nvk_batch_stats(const nvk_batch *b, long long *out2)
{
    if (!b || !out2)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_stats: null argument");
    out2[0] = b->full_uploads, out2[1] = b->patch_steps;
    return NVK_OK;
}

int nvk_batch_begin_step(nvk_batch *b, const int *request_ids, int n, int n_heads, void *stream)
{
    if (!b || n < 0 || (n > 0 && !request_ids) || n_heads <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step: bad arguments");
    if (b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step: the previous step was not ended");
    if (n > b->max_seqs)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step: %d requests, %d slots", n, b->max_seqs);

    // 1. validate the list and count the blocks this step needs (model.hpp:181: a new block when pos % block_size == 0)
    std::vector<int> slot_idx((size_t)n);
    int              need = 0;
    {
        std::unordered_map<int, int> seen;
        for (int i = 0; i < n; i++) {
            auto it = b->slot_of.find(request_ids[i]);
            if (it == b->slot_of.end())
                return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step: request %d is not running", request_ids[i]);
            if (!seen.emplace(request_ids[i], i).second)
                return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step: request %d listed twice", request_ids[i]);
            slot_idx[(size_t)i] = it->second;
            const Slot &s       = b->slots[(size_t)it->second];
            // held blocks do not cover token `len` (a step that failed after allocating keeps its blocks: no second draw)
            if ((long long)s.blocks.size() * b->block_size <= s.len) {
                if (s.len / b->block_size >= b->max_blocks)
                    return nvk::fail(NVK_ERR_INVALID_ARG,
                                     "nvk_batch_begin_step: request %d is at %d tokens, its table holds %d blocks",
                                     s.request_id, s.len, b->max_blocks);
                need++;
            }
        }
    }
    if (need > nvk_bm_num_free_blocks(b->bm))
        return nvk::fail(NVK_ERR_OOM, "Out of memory: no free blocks (%d needed, %d free)", need,
                         nvk_bm_num_free_blocks(b->bm)); // the reference's text, model.hpp:184

    NVK_SCOPE(b->device);
    hipStream_t st = stream_or_pool(b, stream);
    HIP_TRY(hipEventSynchronize(b->staged)); // the pinned buffers are ours again

    // 2. allocate, in list order, through the per-request API (block_manager.hpp:145-154)
    const bool same_list = b->resident_valid && (int)b->resident_ids.size() == n
                           && std::equal(request_ids, request_ids + n, b->resident_ids.begin());
    b->resident_valid = false; // until this step's tables are known to be on their way
    int n_patches = 0, max_ctx = 0;
    for (int i = 0; i < n; i++) {
        Slot &s = b->slots[(size_t)slot_idx[(size_t)i]];
        if ((long long)s.blocks.size() * b->block_size <= s.len) {
            const int id = nvk_bm_allocate_block_for_request(b->bm, s.request_id);
            if (id < 0) { // another thread took blocks between the check and here: undo this step's allocations
                for (int k = 0; k < n_patches; k++) {
                    Slot &u = b->slots[(size_t)slot_idx[(size_t)b->h_patches[3 * k]]];
                    // out of the request's list and back to the free set in one locked step: an id left in the list
                    // would be freed again by free_request once it belongs to someone else
                    (void)nvk_bm_release_block_for_request(b->bm, u.request_id, u.blocks.back());
                    u.blocks.pop_back();
                }
                return nvk::fail(NVK_ERR_OOM, "Out of memory: no free blocks");
            }
            b->h_patches[3 * n_patches]     = i;
            b->h_patches[3 * n_patches + 1] = s.len / b->block_size;
            b->h_patches[3 * n_patches + 2] = id;
            n_patches++;
            s.blocks.push_back(id);
        }
        max_ctx = std::max(max_ctx, s.len + 1);
    }
    // The launch shape (chunking, grid bound) is derived from max_ctx: keep it the same for 256 steps at a time, so a
    // caller that captured the step's layer calls in a hipGraph (nvk_batch_graph_signature) can replay it that long.
    max_ctx = (int)std::min<long long>(((long long)max_ctx + 255) / 256 * 256, (long long)b->max_blocks * b->block_size);

    // 3. bring the step's tables / positions / context lengths to the device
    if (n > 0) {
        if (same_list) {
            if (n_patches > 0)
                HIP_TRY(hipMemcpyAsync(b->d_patches, b->h_patches, (size_t)3 * n_patches * sizeof(int),
                                       hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(nvk_batch_advance_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b->d_tables,
                               b->max_blocks, b->d_pos, b->d_ctx, n, b->d_patches, n_patches);
            HIP_TRY(hipGetLastError());
            b->patch_steps++;
        }
        else {
            for (int i = 0; i < n; i++) {
                const Slot &s   = b->slots[(size_t)slot_idx[(size_t)i]];
                int        *row = b->h_tables + (size_t)i * b->max_blocks;
                std::copy(s.blocks.begin(), s.blocks.end(), row);
                std::fill(row + s.blocks.size(), row + b->max_blocks, 0);
                b->h_pos[i] = s.len;
                b->h_ctx[i] = s.len + 1;
            }
            HIP_TRY(hipMemcpyAsync(b->d_tables, b->h_tables, (size_t)n * b->max_blocks * sizeof(int),
                                   hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(b->d_pos, b->h_pos, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(b->d_ctx, b->h_ctx, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
            b->full_uploads++;
        }
        HIP_TRY(hipEventRecord(b->staged, st));
    }
    b->step_ids.assign(request_ids, request_ids + n);
    b->resident_ids = b->step_ids;

    // 4. one plan for all the layers of this step
    int rc = nvk_plan_build(b->plan, n, n_heads, b->d_ctx, max_ctx, NVK_DECODE_AUTO, 0, (void *)st);
    if (rc != NVK_OK)
        return rc; // positions in HBM may have moved on without the step: resident_valid stays false
    b->resident_valid = n > 0;
    b->n_heads = n_heads;
    b->chunked = false;
    b->in_step = true;
    return NVK_OK;
}

int nvk_batch_begin_step_chunks(nvk_batch *b, const int *request_ids, const int *chunk_lens, int n, int n_heads,
                                void *stream)
{
    if (!b || n < 0 || (n > 0 && (!request_ids || !chunk_lens)) || n_heads <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step_chunks: bad arguments");
    bool all_one = true;
    for (int i = 0; i < n; i++) {
        if (chunk_lens[i] < 1)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step_chunks: request %d brings %d tokens",
                             request_ids[i], chunk_lens[i]);
        all_one = all_one && chunk_lens[i] == 1;
    }
    if (all_one)
        return nvk_batch_begin_step(b, request_ids, n, n_heads, stream); // a pure decode step: the resident-table path
    if (b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step_chunks: the previous step was not ended");
    if (n > b->max_seqs)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step_chunks: %d requests, %d slots", n, b->max_seqs);

    // 1. validate; blocks needed so that every request holds ceil((len + chunk) / block_size) of them
    std::vector<int> slot_idx((size_t)n), fresh((size_t)n);
    long long        need = 0, rows = 0;
    for (int i = 0; i < n; i++)
        rows += chunk_lens[i];
    if (rows > 0x7fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step_chunks: %lld rows in one step", rows);
    {
        std::unordered_map<int, int> seen;
        for (int i = 0; i < n; i++) {
            auto it = b->slot_of.find(request_ids[i]);
            if (it == b->slot_of.end())
                return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step_chunks: request %d is not running",
                                 request_ids[i]);
            if (!seen.emplace(request_ids[i], i).second)
                return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_begin_step_chunks: request %d listed twice",
                                 request_ids[i]);
            slot_idx[(size_t)i] = it->second;
            const Slot     &s    = b->slots[(size_t)it->second];
            const long long want = ((long long)s.len + chunk_lens[i] + b->block_size - 1) / b->block_size;
            if (want > b->max_blocks)
                return nvk::fail(NVK_ERR_INVALID_ARG,
                                 "nvk_batch_begin_step_chunks: request %d would reach %lld tokens, its table holds %d blocks",
                                 s.request_id, (long long)s.len + chunk_lens[i], b->max_blocks);
            fresh[(size_t)i] = (int)std::max(0LL, want - (long long)s.blocks.size());
            need += fresh[(size_t)i];
        }
    }
    if (need > nvk_bm_num_free_blocks(b->bm))
        return nvk::fail(NVK_ERR_OOM, "Out of memory: no free blocks (%lld needed, %d free)", need,
                         nvk_bm_num_free_blocks(b->bm));

    NVK_SCOPE(b->device);
    hipStream_t st = stream_or_pool(b, stream);
    HIP_TRY(hipEventSynchronize(b->staged));
    b->resident_valid = false; // rows of a mixed step are not the decode list: the next decode step re-sends its tables

    // 2. allocate in list order (block_manager.hpp:145-154, as many times as the chunk needs)
    for (int i = 0; i < n; i++) {
        Slot &s = b->slots[(size_t)slot_idx[(size_t)i]];
        for (int k = 0; k < fresh[(size_t)i]; k++) {
            const int id = nvk_bm_allocate_block_for_request(b->bm, s.request_id);
            if (id < 0) { // raced with another thread: give this step's blocks back
                for (int u = 0; u <= i; u++) {
                    Slot &t = b->slots[(size_t)slot_idx[(size_t)u]];
                    for (int r = (u == i ? k : fresh[(size_t)u]); r > 0; r--) {
                        (void)nvk_bm_release_block_for_request(b->bm, t.request_id, t.blocks.back());
                        t.blocks.pop_back();
                    }
                }
                return nvk::fail(NVK_ERR_OOM, "Out of memory: no free blocks");
            }
            s.blocks.push_back(id);
        }
    }

    // 3. the step's rows: the leading one-token requests are the decode part (ScheduledBatch::decode_requests),
    //    everything behind them the prefill part (ScheduledBatch::prefill_requests)
    int nd = 0;
    while (nd < n && chunk_lens[nd] == 1)
        nd++;
    b->step_chunks.assign(chunk_lens, chunk_lens + n);
    b->step_starts.resize((size_t)n);
    int max_ctx = 0;
    for (int i = 0; i < n; i++) {
        const Slot &s   = b->slots[(size_t)slot_idx[(size_t)i]];
        int        *row = b->h_tables + (size_t)i * b->max_blocks;
        std::copy(s.blocks.begin(), s.blocks.end(), row);
        std::fill(row + s.blocks.size(), row + b->max_blocks, 0);
        b->step_starts[(size_t)i] = s.len;
        if (i < nd) {
            b->h_pos[i] = s.len;
            b->h_ctx[i] = s.len + 1;
            max_ctx     = std::max(max_ctx, s.len + 1);
        }
    }
    HIP_TRY(hipMemcpyAsync(b->d_tables, b->h_tables, (size_t)n * b->max_blocks * sizeof(int), hipMemcpyHostToDevice, st));
    if (nd > 0) {
        HIP_TRY(hipMemcpyAsync(b->d_pos, b->h_pos, (size_t)nd * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_ctx, b->h_ctx, (size_t)nd * sizeof(int), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipEventRecord(b->staged, st));
    b->full_uploads++;
    b->step_ids.assign(request_ids, request_ids + n);
    if (nd > 0) {
        int rc = nvk_plan_build(b->plan, nd, n_heads, b->d_ctx, max_ctx, NVK_DECODE_AUTO, 0, (void *)st);
        if (rc != NVK_OK)
            return rc;
    }
    b->n_heads  = n_heads;
    b->chunked  = true;
    b->n_decode = nd;
    b->in_step  = true;
    return NVK_OK;
}

int nvk_batch_step_rows(const nvk_batch *b)
{
    if (!b || !b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_step_rows: no step in flight");
    if (!b->chunked)
        return (int)b->step_ids.size();
    long long rows = 0;
    for (int c : b->step_chunks)
        rows += c;
    return (int)rows;
}

int nvk_batch_layer(nvk_batch *b, int layer, float *q_dev, const float *k_rows_dev, const float *v_rows_dev,
                    float *out_dev, float rope_theta, void *stream)
{
    if (!b || !b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_layer: no step in flight (call nvk_batch_begin_step)");
    const int n = (int)b->step_ids.size();
    if (n == 0)
        return NVK_OK;
    if (!q_dev || !k_rows_dev || !v_rows_dev || !out_dev || rope_theta < 0.f)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_layer: bad arguments");
    NVK_SCOPE(b->device);
    hipStream_t st = stream_or_pool(b, stream);
    if (b->chunked) {
        // decode part: rows [0, nd), one per request; prefill part: the chunks' rows packed behind them
        const int    nd = b->n_decode, np = n - nd;
        const size_t qrow = (size_t)b->n_heads * b->head_dim;
        if (nd > 0) {
            int rc = nvk_pool_rope_append_batch(b->pool, layer, nd, b->n_heads, q_dev, k_rows_dev, v_rows_dev, b->d_tables,
                                                b->max_blocks, b->d_pos, rope_theta, NVK_PTR_DEVICE, (void *)st);
            if (rc == NVK_OK)
                rc = nvk_decode_planned(b->pool, layer, b->plan, q_dev, b->d_tables, b->max_blocks, out_dev, (void *)st);
            if (rc != NVK_OK)
                return rc;
        }
        if (np > 0) {
            float       *q2 = q_dev + (size_t)nd * qrow;
            const float *k2 = k_rows_dev + (size_t)nd * b->row_len, *v2 = v_rows_dev + (size_t)nd * b->row_len;
            const int   *t2 = b->d_tables + (size_t)nd * b->max_blocks;
            // forward_chunk's order (model_chunked.hpp:79-96): rope + append the whole chunk, then chunked attention
            int rc = nvk_pool_rope_append_chunks(b->pool, layer, np, b->n_heads, q2, k2, v2, t2, b->max_blocks,
                                                 b->step_starts.data() + nd, b->step_chunks.data() + nd, rope_theta,
                                                 NVK_PTR_DEVICE, (void *)st);
            if (rc == NVK_OK)
                rc = nvk_prefill_device(b->pool, layer, np, b->n_heads, q2, t2, b->max_blocks, b->step_starts.data() + nd,
                                        b->step_chunks.data() + nd, out_dev + (size_t)nd * qrow, NVK_PREFILL_AUTO,
                                        (void *)st);
            if (rc != NVK_OK)
                return rc;
        }
        return NVK_OK;
    }
    // KV append for every sequence of the step (model.hpp:190-207), with the rotation of model.hpp:169-175 when asked
    int rc = nvk_pool_rope_append_batch(b->pool, layer, n, b->n_heads, q_dev, k_rows_dev, v_rows_dev, b->d_tables,
                                        b->max_blocks, b->d_pos, rope_theta, NVK_PTR_DEVICE, (void *)st);
    if (rc != NVK_OK)
        return rc;
    // attention over pos + 1 tokens: the new token attends to itself (model.hpp:221, :397)
    return nvk_decode_planned(b->pool, layer, b->plan, q_dev, b->d_tables, b->max_blocks, out_dev, (void *)st);
}

} // extern "C"

namespace {

// grow nvk_batch_layer_hidden's scratch rows (q / k / v / normalised x) to `rows` x the step's dimensions
int reserve_hidden_rows(nvk_batch *b, int rows, int dim)
{
    const size_t qdim = (size_t)b->n_heads * b->head_dim;
    if ((size_t)rows <= b->h_rows && (size_t)dim <= b->h_dim && qdim <= b->h_qdim)
        return NVK_OK;
    HIP_TRY(hipDeviceSynchronize()); // earlier launches may still use the old ones
    for (float **p : {&b->h_q, &b->h_k, &b->h_v, &b->h_xb})
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    b->h_rows = b->h_dim = b->h_qdim = 0;
    const size_t r = std::max((size_t)rows, (size_t)b->max_seqs), d = (size_t)dim;
    HIP_TRY(hipMalloc((void **)&b->h_q, r * qdim * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&b->h_k, r * b->row_len * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&b->h_v, r * b->row_len * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&b->h_xb, r * d * sizeof(float)));
    b->h_rows = r, b->h_dim = d, b->h_qdim = qdim;
    b->hidden_generation++;
    return NVK_OK;
}

// every row of the step decodes, and they are few enough for the GEMV with the RoPE + append epilogue
bool decode_rows_fuse(const nvk_batch *b, int rows, int dim, float rope_theta)
{
    const bool all_decode = !b->chunked || b->n_decode == (int)b->step_ids.size();
    return all_decode && rope_theta > 0.f && b->head_dim % 2 == 0 && (rows <= 2 || b->head_dim % 4 == 0)
           && nvk::qkv_rope_append_ok(rows, dim, b->n_heads * b->head_dim, b->row_len, b->device);
}

// model.hpp:164-175, 190-207, 221 for such a step: q | k | v GEMV with the rotation and the append in its epilogue, then
// the attention
int project_rope_append_attend(nvk_batch *b, int layer, const float *act, int dim, const nvk::NormLink *link,
                               const float *wq_dev, const float *wk_dev, const float *wv_dev, float *out_dev,
                               float rope_theta, hipStream_t st)
{
    nvk::RopeAppendTarget target;
    int rc = nvk::pool_rope_append_target(b->pool, layer, rope_theta, b->d_tables, b->max_blocks, b->d_pos, &target);
    if (rc == NVK_OK)
        rc = nvk::qkv_rope_append(target, b->h_q, act, wq_dev, wk_dev, wv_dev, nvk_batch_step_rows(b), dim,
                                  b->n_heads * b->head_dim, b->device, (void *)st, link);
    if (rc != NVK_OK)
        return rc;
    return nvk_decode_planned(b->pool, layer, b->plan, b->h_q, b->d_tables, b->max_blocks, out_dev, (void *)st);
}

} // namespace

int nvk::batch_layer_hidden_linked(nvk_batch *b, int layer, const float *xg_dev, int dim, const NormLink &link,
                                   const float *wq_dev, const float *wk_dev, const float *wv_dev, float *out_dev,
                                   float rope_theta, void *stream)
{
    if (!b || !b->in_step || b->step_ids.empty() || !xg_dev || dim <= 0 || !wq_dev || !wk_dev || !wv_dev || !out_dev
        || !link.ss_in)
        return nvk::fail(NVK_ERR_INVALID_ARG, "batch_layer_hidden_linked: bad arguments");
    const int rows = nvk_batch_step_rows(b);
    if (rows < 0)
        return rows;
    if (!decode_rows_fuse(b, rows, dim, rope_theta))
        return nvk::fail(NVK_ERR_INVALID_ARG, "batch_layer_hidden_linked: not a step of one or two decoding rows");
    NVK_SCOPE(b->device);
    int              rc = reserve_hidden_rows(b, rows, dim);
    if (rc != NVK_OK)
        return rc;
    return project_rope_append_attend(b, layer, xg_dev, dim, &link, wq_dev, wk_dev, wv_dev, out_dev, rope_theta,
                                      stream_or_pool(b, stream));
}

extern "C" {

int nvk_batch_layer_hidden(nvk_batch *b, int layer, const float *x_dev, int dim, const float *rms_weight_dev,
                           const float *wq_dev, const float *wk_dev, const float *wv_dev, float rms_eps, float *out_dev,
                           float rope_theta, void *stream)
{
    if (!b || !b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_layer_hidden: no step in flight (call nvk_batch_begin_step)");
    if (b->step_ids.empty())
        return NVK_OK;
    if (!x_dev || dim <= 0 || !wq_dev || !wk_dev || !wv_dev || !out_dev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_layer_hidden: bad arguments");
    const int rows = nvk_batch_step_rows(b);
    if (rows < 0)
        return rows;
    const size_t qdim = (size_t)b->n_heads * b->head_dim;
    NVK_SCOPE(b->device);
    hipStream_t  st = stream_or_pool(b, stream);
    int rc0 = reserve_hidden_rows(b, rows, dim);
    if (rc0 != NVK_OK)
        return rc0;
    // One or two decoding rows: the projections, the rotation and the append are one launch (the GEMV's epilogue
    // writes where nvk_rope_append_kernel would: same arithmetic, same bits, one kernel and two scratch rows less).
    if (decode_rows_fuse(b, rows, dim, rope_theta)) {
        const float *act = x_dev;
        if (rms_weight_dev) { // model.hpp:161
            int rc = nvk_batch_rms_norm_device(b->h_xb, x_dev, rms_weight_dev, rows, dim, rms_eps, b->device, (void *)st);
            if (rc != NVK_OK)
                return rc;
            act = b->h_xb;
        }
        return project_rope_append_attend(b, layer, act, dim, nullptr, wq_dev, wk_dev, wv_dev, out_dev, rope_theta, st);
    }
    // model.hpp:161-166: xb = rms_norm(x), q / k / v = wq / wk / wv . xb -- then :169-221 as nvk_batch_layer
    int rc = nvk_qkv_project_device(b->h_q, b->h_k, b->h_v, x_dev, b->h_xb, rms_weight_dev, wq_dev, wk_dev, wv_dev, rows,
                                    dim, (int)qdim, b->row_len, rms_eps, b->device, (void *)st);
    if (rc != NVK_OK)
        return rc;
    return nvk_batch_layer(b, layer, b->h_q, b->h_k, b->h_v, out_dev, rope_theta, (void *)st);
}

int nvk_batch_graph_signature(const nvk_batch *b, long long *out8)
{
    if (!b || !out8 || !b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_graph_signature: no step in flight");
    int info[4] = {0, 0, 0, 0};
    if (!b->step_ids.empty() && !(b->chunked && b->n_decode == 0)) {
        int rc = nvk_plan_info(b->plan, info);
        if (rc != NVK_OK)
            return rc;
    }
    out8[0] = (long long)b->step_ids.size();
    out8[1] = b->chunked ? 1 : 0;
    out8[2] = info[0], out8[3] = info[1], out8[4] = info[2], out8[5] = info[3];
    // everything a captured step bakes in by address, as generation counters (an address can come back from
    // hipMalloc while its neighbours moved): the pool's workspace, the plan's work list, the scratch rows
    out8[6] = nvk_pool_workspace_generation(b->pool) * 1000003LL + nvk::plan_generation(b->plan);
    out8[7] = b->hidden_generation ^ ((long long)b->n_heads << 48);
    return NVK_OK;
}

int nvk_batch_abort_step(nvk_batch *b)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!b)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_abort_step: null batch");
    // no request advances; blocks drawn for the step stay with their requests (the next step needs them anyway)
    b->in_step        = false;
    b->chunked        = false;
    b->resident_valid = false;
    return NVK_OK;
}

int nvk_batch_end_step(nvk_batch *b)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!b || !b->in_step)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_end_step: no step in flight");
    for (size_t i = 0; i < b->step_ids.size(); i++)
        b->slots[(size_t)b->slot_of[b->step_ids[i]]].len += b->chunked ? b->step_chunks[i] : 1;
    b->in_step = false;
    b->chunked = false;
    return NVK_OK;
}

} // extern "C"
