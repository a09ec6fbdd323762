// nvk_shard.hip -- the head-sharded split of the decode layer-call behind the C ABI.
//
// The reference is one process and one thread; its attention loops over q-heads outermost
// (include/core/attention.hpp:102) with kv-head h / kv_mul (:105), so heads are independent: rank g of G holds
// kv-heads [g*KV/G, (g+1)*KV/G) of EVERY sequence in a pool of [num_blocks, block_size, KV/G, head_dim], computes
// out[:, g*H/G : (g+1)*H/G, :], and the G slices are joined by ONE all-gather per layer-call into
// [G, num_seqs, H/G, head_dim] (transpose-free: the consumer indexes head h at [h / (H/G)][s][h % (H/G)]).
//
// The collective is RCCL's ncclAllGather over xGMI, called directly (librccl.so is opened at the first
// nvk_shard_create, so single-GPU users of the library never load it).  xGMI is point-to-point, 7 links x ~153 GB/s
// per GPU: the gather of 4096 sequences at G = 8 moves 8 MiB per rank (~55 us one-shot) against ~1.07 ms of
// attention per layer-call, so it is put on a stream of its own behind an event and hidden under the next
// layer-call's attention (two `out` buffers in turn: nvk_shard_decode_gather / nvk_shard_wait_gather).
// The sequence-sharded split needs nothing from here: its ranks share nothing (no collective).

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "nvk_device_scope.h"
#include "nvk_internal.h"

namespace {

// the slice of RCCL's C API this file uses (rccl/rccl.h, ABI-stable NCCL 2 signatures)
typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId128
{
    char internal[128];
};
typedef int (*fn_get_unique_id)(ncclUniqueId128 *);
typedef int (*fn_comm_init_rank)(ncclComm_t *, int, ncclUniqueId128, int);
typedef int (*fn_comm_destroy)(ncclComm_t);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
typedef const char *(*fn_error_string)(int);
constexpr int kNcclFloat = 7; // ncclFloat32

struct Rccl
{
    void             *handle = nullptr;
    fn_get_unique_id  get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy   comm_destroy = nullptr;
    fn_all_gather     all_gather = nullptr;
    fn_error_string   error_string = nullptr;
};
Rccl       g_rccl;
std::mutex g_rccl_mutex;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle)
        return NVK_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
        h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
        return nvk::fail(NVK_ERR_HIP, "nvk_shard: cannot open librccl.so (%s)", dlerror());
    Rccl r;
    r.handle         = h;
    r.get_unique_id  = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    r.comm_destroy   = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    r.all_gather     = (fn_all_gather)dlsym(h, "ncclAllGather");
    r.error_string   = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.error_string) {
        dlclose(h);
        return nvk::fail(NVK_ERR_HIP, "nvk_shard: librccl.so lacks an expected symbol");
    }
    g_rccl = r;
    return NVK_OK;
}

int rccl_fail(int code, const char *what)
{
    return nvk::fail(NVK_ERR_HIP, "nvk_shard: %s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(code) : "?");
}

} // namespace

#define HIP_TRY_S(expr)                                                                                                \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return nvk::fail(NVK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

struct nvk_shard
{
    int         rank = 0, world = 1, device = 0, n_heads = 0, n_kv_heads = 0;
    ncclComm_t  comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t  computed[2] = {nullptr, nullptr}; // out buffer b holds this layer-call's slice
    hipEvent_t  gathered[2] = {nullptr, nullptr}; // gathered buffer b is complete (and out buffer b free again)
    bool        pending[2]  = {false, false};
    hipStream_t waited_on[2] = {nullptr, nullptr}; // the stream that has already been made to wait for gathered[b]
    int         mode = NVK_SHARD_GATHER_AUTO;
};

namespace {

// Where the collective of a layer-call runs.  Handing a layer-call's rows to a second stream and taking the gathered
// rows back costs the compute stream 8-9 us per layer-call on this runtime even when nothing has to wait (a record
// and a wait packet in its queue; tools/shard_sync_probe.hip, profiles/r03_shard_sync_probe.log: 27.7 us per step
// plain, 36.8 handed over, 29.3 with the copy in line), so it pays only where there is a collective to hide: with
// one rank the "gather" is a device copy, and it goes in line.
bool gather_overlapped(const nvk_shard *s)
{
    if (s->mode == NVK_SHARD_GATHER_AUTO)
        return s->world > 1;
    return s->mode == NVK_SHARD_GATHER_OVERLAPPED;
}

} // namespace

extern "C" {

int nvk_shard_head_range(int rank, int world, int n_heads, int n_kv_heads, int *out4)
{
    if (!out4 || world <= 0 || rank < 0 || rank >= world || n_kv_heads <= 0 || n_heads <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_head_range: bad rank/world/heads");
    if (n_heads % n_kv_heads != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_head_range: n_heads=%d is not a multiple of n_kv_heads=%d",
                         n_heads, n_kv_heads);
    if (n_kv_heads % world != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_head_range: world=%d must divide n_kv_heads=%d", world,
                         n_kv_heads);
    const int lkv = n_kv_heads / world, lh = n_heads / world;
    out4[0] = lh, out4[1] = lkv, out4[2] = rank * lh, out4[3] = rank * lkv;
    return NVK_OK;
}

long long nvk_shard_gather_bytes(int num_seqs, int n_heads, int head_dim, int world)
{
    if (num_seqs < 0 || n_heads <= 0 || head_dim <= 0 || world <= 0 || n_heads % world != 0)
        return -1;
    return (long long)num_seqs * (n_heads / world) * head_dim * 4; // what each rank contributes; x world = what it holds
}

int nvk_shard_unique_id(void *id128)
{
    if (!id128)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_unique_id: null buffer (128 bytes needed)");
    int rc = load_rccl();
    if (rc != NVK_OK)
        return rc;
    ncclUniqueId128 id;
    const int       e = g_rccl.get_unique_id(&id);
    if (e != 0)
        return rccl_fail(e, "ncclGetUniqueId");
    std::memcpy(id128, &id, sizeof id);
    return NVK_OK;
}

int nvk_shard_create(int rank, int world, const void *id128, int device, int n_heads, int n_kv_heads,
                     nvk_shard **out_shard)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!out_shard)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_create: out_shard is null");
    *out_shard = nullptr;
    int range[4];
    int rc = nvk_shard_head_range(rank, world, n_heads, n_kv_heads, range);
    if (rc != NVK_OK)
        return rc;
    if (!id128)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_create: null unique id");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nvk::fail(NVK_ERR_NO_DEVICE, "nvk_shard_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_create: device %d outside [0,%d)", device, ndev);
    rc = load_rccl();
    if (rc != NVK_OK)
        return rc;
    nvk::DeviceScope scope(device);
    if (!scope.ok)
        return nvk::fail(NVK_ERR_HIP, "nvk_shard_create: hipSetDevice(%d) failed", device);
    nvk_shard *s = new (std::nothrow) nvk_shard();
    if (!s)
        return nvk::fail(NVK_ERR_OOM, "nvk_shard_create: host allocation failed");
    s->rank = rank, s->world = world, s->device = device, s->n_heads = n_heads, s->n_kv_heads = n_kv_heads;
    ncclUniqueId128 id;
    std::memcpy(&id, id128, sizeof id);
    const int e = g_rccl.comm_init_rank(&s->comm, world, id, rank);
    if (e != 0) {
        delete s;
        return rccl_fail(e, "ncclCommInitRank");
    }
    hipError_t h = hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking);
    for (int b = 0; b < 2 && h == hipSuccess; b++) {
        h = hipEventCreateWithFlags(&s->computed[b], hipEventDisableTiming);
        if (h == hipSuccess)
            h = hipEventCreateWithFlags(&s->gathered[b], hipEventDisableTiming);
    }
    if (h != hipSuccess) {
        const int code = nvk::fail(NVK_ERR_HIP, "nvk_shard_create: %s", hipGetErrorString(h));
        nvk_shard_destroy(s);
        return code;
    }
    *out_shard = s;
    return NVK_OK;
}

int nvk_shard_destroy(nvk_shard *s)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!s)
        return NVK_OK;
    nvk::DeviceScope scope(s->device);
    if (s->comm_stream)
        (void)hipStreamSynchronize(s->comm_stream);
    if (s->comm)
        (void)g_rccl.comm_destroy(s->comm);
    for (int b = 0; b < 2; b++) {
        if (s->computed[b])
            (void)hipEventDestroy(s->computed[b]);
        if (s->gathered[b])
            (void)hipEventDestroy(s->gathered[b]);
    }
    if (s->comm_stream)
        (void)hipStreamDestroy(s->comm_stream);
    delete s;
    return NVK_OK;
}

int nvk_shard_info(const nvk_shard *s, int *out6)
{
    if (!s || !out6)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_info: null argument");
    int r[4];
    int rc = nvk_shard_head_range(s->rank, s->world, s->n_heads, s->n_kv_heads, r);
    if (rc != NVK_OK)
        return rc;
    out6[0] = s->rank, out6[1] = s->world, out6[2] = r[0], out6[3] = r[1], out6[4] = r[2], out6[5] = r[3];
    return NVK_OK;
}

int nvk_shard_all_gather_out(nvk_shard *s, const float *out_local_dev, float *gathered_dev, int num_seqs, int head_dim,
                             void *stream)
{
    if (!s || num_seqs < 0 || head_dim <= 0 || (num_seqs > 0 && (!out_local_dev || !gathered_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_all_gather_out: bad arguments");
    if (num_seqs == 0)
        return NVK_OK;
    NVK_SCOPE(s->device);
    const size_t count = (size_t)num_seqs * (s->n_heads / s->world) * head_dim;
    const int    e = g_rccl.all_gather(out_local_dev, gathered_dev, count, kNcclFloat, s->comm,
                                       stream ? (hipStream_t)stream : s->comm_stream);
    if (e != 0)
        return rccl_fail(e, "ncclAllGather");
    return NVK_OK;
}

int nvk_shard_decode_gather(nvk_shard *s, nvk_pool *pool, int layer, int num_seqs, const float *q_local_dev,
                            const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                            float *out_local_dev, float *gathered_dev, int buffer, int flags, int chunk_tokens,
                            void *compute_stream)
{
    if (!s || !pool || (buffer != 0 && buffer != 1))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_decode_gather: bad arguments (buffer must be 0 or 1)");
    int geo[6];
    int rc = nvk_pool_geometry(pool, geo);
    if (rc != NVK_OK)
        return rc;
    if (geo[3] != s->n_kv_heads / s->world || geo[5] != s->device)
        return nvk::fail(NVK_ERR_INVALID_ARG,
                         "nvk_shard_decode_gather: the pool holds %d kv-heads on device %d, this rank's shard is %d on %d",
                         geo[3], geo[5], s->n_kv_heads / s->world, s->device);
    NVK_SCOPE(s->device);
    void *pool_stream = nullptr;
    rc                = nvk_pool_stream(pool, &pool_stream);
    if (rc != NVK_OK)
        return rc;
    hipStream_t cs = compute_stream ? (hipStream_t)compute_stream : (hipStream_t)pool_stream;
    const bool  overlapped = gather_overlapped(s);
    // out buffer `buffer` is free again once the gather that last read it has finished (nothing to wait for when
    // this stream has already been made to wait for that gather -- its consumer ran here -- or, inline, when the
    // gather itself ran on this stream)
    if (s->pending[buffer] && s->waited_on[buffer] != cs)
        HIP_TRY_S(hipStreamWaitEvent(cs, s->gathered[buffer], 0));
    rc = nvk_decode_device(pool, layer, num_seqs, s->n_heads / s->world, q_local_dev, block_tables_dev, max_blocks,
                           ctx_lens_dev, max_ctx, out_local_dev, flags, chunk_tokens, (void *)cs);
    if (rc != NVK_OK)
        return rc;
    if (!overlapped) {
        // the collective right behind the attention on the compute stream: no hand-over, nothing hidden either
        rc = nvk_shard_all_gather_out(s, out_local_dev, gathered_dev, num_seqs, geo[4], (void *)cs);
        if (rc != NVK_OK)
            return rc;
        HIP_TRY_S(hipEventRecord(s->gathered[buffer], cs)); // (for waiters on other streams and for the host)
        s->pending[buffer]   = true;
        s->waited_on[buffer] = cs; // stream order does it
        return NVK_OK;
    }
    HIP_TRY_S(hipEventRecord(s->computed[buffer], cs));
    HIP_TRY_S(hipStreamWaitEvent(s->comm_stream, s->computed[buffer], 0));
    rc = nvk_shard_all_gather_out(s, out_local_dev, gathered_dev, num_seqs, geo[4], (void *)s->comm_stream);
    if (rc != NVK_OK)
        return rc;
    HIP_TRY_S(hipEventRecord(s->gathered[buffer], s->comm_stream));
    s->pending[buffer]   = true;
    s->waited_on[buffer] = nullptr;
    return NVK_OK;
}

int nvk_shard_wait_gather(nvk_shard *s, int buffer, void *stream)
{
    if (!s || (buffer != 0 && buffer != 1))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_wait_gather: bad arguments");
    if (!s->pending[buffer])
        return NVK_OK;
    NVK_SCOPE(s->device);
    if (stream) {
        if (s->waited_on[buffer] != (hipStream_t)stream) { // (a stream waits for a gather once)
            HIP_TRY_S(hipStreamWaitEvent((hipStream_t)stream, s->gathered[buffer], 0));
            s->waited_on[buffer] = (hipStream_t)stream;
        }
    }
    else
        HIP_TRY_S(hipEventSynchronize(s->gathered[buffer]));
    return NVK_OK;
}

int nvk_shard_set_gather_mode(nvk_shard *s, int mode)
{
    if (!s || (mode != NVK_SHARD_GATHER_AUTO && mode != NVK_SHARD_GATHER_OVERLAPPED && mode != NVK_SHARD_GATHER_INLINE))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_set_gather_mode: bad arguments");
    s->mode = mode;
    return NVK_OK;
}

int nvk_shard_gather_mode(const nvk_shard *s)
{
    if (!s)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_shard_gather_mode: null shard");
    return gather_overlapped(s) ? NVK_SHARD_GATHER_OVERLAPPED : NVK_SHARD_GATHER_INLINE;
}

} // extern "C"
