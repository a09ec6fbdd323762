// nvk_paged_attn.hip -- host side of libnvk_paged_attn.so: the device-resident
// paged KV pool, kernel dispatch and the C ABI declared in include/nvk_paged_attn.h.
// Device code is in nvk_kernels.hip.h.  gfx950 only; there is no CPU fallback:
// every compute entry point fails with NVK_ERR_NO_DEVICE / NVK_ERR_HIP when no
// MI355X is visible.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "nvk_device_scope.h"
#include "nvk_internal.h"
#include "nvk_kernels.hip.h"

#define NVK_VERSION_STRING "nvk_paged_attn 0.1.0 (gfx950)"

#ifndef NVK_WG_WAVES
#define NVK_WG_WAVES 8 // kv-heads (= waves) per workgroup, at most
#endif
#ifndef NVK_MAX_QPK
#define NVK_MAX_QPK 8 // q-heads one wave carries at most (register budget: q and the accumulators grow with it)
#endif
#ifndef NVK_AUTO_MIN_CHUNK
#define NVK_AUTO_MIN_CHUNK 128 // shortest chunk NVK_DECODE_AUTO picks (a multiple of every kernel step: 4..16 tokens)
#endif
#ifndef NVK_AUTO_MIN_CHUNK_WIDE
#define NVK_AUTO_MIN_CHUNK_WIDE 32 // the same when the merge is nvk_merge_wide_kernel (few rows; tools/merge_sweep.sh)
#endif
#ifndef NVK_TINY_CHUNK_BYTES
#define NVK_TINY_CHUNK_BYTES 0 // K + V bytes per workgroup up to which few rows are cut into <= 16 chunks so that the decode
                               // kernel combines them itself (no merge launch).  Built and measured in round 3 with 256 KiB:
                               // one TinyLlama request at 1K context 1.340 -> 1.393 ms per model step -- 16 workgroups of 64
                               // tokens are slower than 32 of 32 plus the merge launch (profiles/r03_model_engine.log): off
#endif
#ifndef NVK_MERGE_WIDE_ROWS
#define NVK_MERGE_WIDE_ROWS 1024 // (rows x q-heads) up to which the merge runs one workgroup per (row, head); at 2048
                                 // (64 rows x 32 heads, 4 chunks) the two forms measure 5.5 and 5.1 us
#endif
#ifndef NVK_AUTO_MAX_CHUNK
#define NVK_AUTO_MAX_CHUNK 2048 // longest chunk NVK_DECODE_AUTO picks
#endif
#ifndef NVK_WIDE_STEPS
#define NVK_WIDE_STEPS 1 // 1: chunks of NVK_WIDE_MIN_CHUNK tokens and more run nvk_decode_wide_kernel (16-token steps); 0 for A/B
#endif
#ifndef NVK_WIDE_MIN_CHUNK
#define NVK_WIDE_MIN_CHUNK 1024
#endif
#ifndef NVK_MFMA_MIN_KVMUL
#define NVK_MFMA_MIN_KVMUL 5 // q-heads per kv-head from which decode runs on fp32 MFMA tiles (head_dim 64 or 128, <= 16 heads)
#endif
#ifndef NVK_POOL_LAYOUT
#define NVK_POOL_LAYOUT 1 // 1: one allocation, [layer][K slab | V slab]; 0: K pool and V pool, [layer][slab] each (A/B)
#endif
#ifndef NVK_XCD_RANGES
#define NVK_XCD_RANGES 1 // 1: uniform batches deal each XCD a contiguous eighth of the items (xcd_range_item)
#endif
#ifndef NVK_XCD_SKEW_PERMILLE
#define NVK_XCD_SKEW_PERMILLE 40 // thousandths of a chunk that an even XCD's chunk hands to the odd XCD's behind it ...
#endif
#ifndef NVK_XCD_SKEW_ALL_SHAPES
#define NVK_XCD_SKEW_ALL_SHAPES 0 // 1: also token-split workgroups, 64-float rows and every MFMA shape (A/B: they lose)
#endif
#ifndef NVK_XCD_SKEW_MIN_CHUNK
#define NVK_XCD_SKEW_MIN_CHUNK 1024 // ... where sequences are cut into chunks of at least this many tokens (make_shape)
#endif
#ifndef NVK_POOL_SKEW_BYTES
#define NVK_POOL_SKEW_BYTES 0 // bytes of padding behind every slab of the one-allocation layout (a multiple of 256)
#endif
#ifndef NVK_KV_PAIRS
#define NVK_KV_PAIRS 1 // 1: rows of 64 / 192 B are streamed a pair of adjacent kv-heads per wave (whole lines)
#endif
#ifndef NVK_MIRROR_TABLE_CHECK
#define NVK_MIRROR_TABLE_CHECK 1 // 0 (A/B): the append-only host mirror trusts its rows on "one token more" alone
#endif
#ifndef NVK_FUSE_MAX_CHUNKS
#define NVK_FUSE_MAX_CHUNKS 16 // chunks per sequence up to which the split-K combine runs inside the decode kernel (the
                               // workgroup that completes a sequence merges it: one CU's worth of reading, fine for a
                               // few chunks); beyond that the merge kernels spread a sequence's chunks over many CUs
#endif

// ------------------------------------------------------------------ errors
namespace nvk {

static thread_local char g_last_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
    return code;
}

} // namespace nvk

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return nvk::fail(e__ == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP, "%s failed: %s (%s:%d)", #expr,  \
                             hipGetErrorString(e__), __FILE__, __LINE__);                                              \
    } while (0)

namespace {

// ceil(2^32 / block_size) for nvk::token_block; 0 stands for block_size 1
unsigned division_magic(int block_size)
{
    if (block_size <= 1)
        return 0u;
    return (unsigned)(((1ULL << 32) + (unsigned)block_size - 1) / (unsigned)block_size);
}

int round_up(int v, int to) { return (v + to - 1) / to * to; }

} // namespace

// -------------------------------------------------------------------- pool
struct nvk_pool
{
    int    n_layers = 0, num_blocks = 0, block_size = 0, n_kv_heads = 0, head_dim = 0, device = 0;
    unsigned bs_magic   = 0;
    size_t layer_floats = 0; // num_blocks * block_size * n_kv_heads * head_dim
    float *k = nullptr, *v = nullptr;
    size_t layer_stride = 0; // floats from one layer's K (V) slab to the next layer's
    bool   one_allocation = false; // NVK_POOL_LAYOUT 1: [layer][K slab | V slab] in one allocation (v = k + layer_floats)
    hipStream_t stream = nullptr;
    int         num_cus = 256;

    // split-K workspace (grown on demand, reused across calls)
    float *ws_o = nullptr, *ws_ml = nullptr;
    size_t ws_o_floats = 0, ws_ml_floats = 0;
    long long ws_generation = 0; // bumped whenever the workspace moves (captured graphs hold its address)
    // arrival counters of the in-kernel split-K combine: [rows, kv-head groups x q sub-groups], zero between
    // launches (zeroed when allocated; the workgroup that completes a row zeroes its counter again)
    int   *arrivals     = nullptr;
    size_t arrivals_cap = 0;

    // the plan nvk_decode / nvk_decode_device rebuild on every call (callers that
    // reuse one plan across the layers of a model step own an nvk_plan instead)
    nvk_plan *own_plan = nullptr;

    // staging for the host-pointer entry points
    char  *stage     = nullptr;
    size_t stage_cap = 0;
    // pinned host twin of the head of `stage` for nvk_decode: q | tables | lengths go to the device in ONE copy and
    // `out` comes back through it (copies from pageable memory cost the runtime a staging step and a wait each)
    char  *pin     = nullptr;
    size_t pin_cap = 0;

    hipEvent_t ev_a = nullptr, ev_b = nullptr;

    // chunk descriptors of the prefill entry points: pinned host staging and its device mirror
    int       *desc_host = nullptr, *desc_dev = nullptr;
    size_t     desc_cap  = 0;       // ints
    hipEvent_t desc_ev   = nullptr; // the last copy out of desc_host has finished

    // RoPE inverse frequencies for (rope_theta, head_dim), computed on the host with libm
    float *rope_inv_freq = nullptr; // [head_dim / 2] on the device
    float  rope_theta    = 0.f;     // 0 = table not built

    float *k_layer(int layer) const { return k + (size_t)layer * layer_stride; }
    float *v_layer(int layer) const { return v + (size_t)layer * layer_stride; }
    int    row_len() const { return n_kv_heads * head_dim; }
};

namespace {

int ensure_stage(nvk_pool *pool, size_t bytes)
{
    if (bytes <= pool->stage_cap)
        return NVK_OK;
    if (pool->stage) {
        HIP_TRY(hipStreamSynchronize(pool->stream));
        HIP_TRY(hipFree(pool->stage));
        pool->stage     = nullptr;
        pool->stage_cap = 0;
    }
    const size_t want = std::max(bytes, (size_t)1 << 20);
    HIP_TRY(hipMalloc(&pool->stage, want));
    pool->stage_cap = want;
    return NVK_OK;
}

int ensure_pin(nvk_pool *pool, size_t bytes)
{
    if (bytes <= pool->pin_cap)
        return NVK_OK;
    if (pool->pin) {
        HIP_TRY(hipStreamSynchronize(pool->stream));
        HIP_TRY(hipHostFree(pool->pin));
        pool->pin     = nullptr;
        pool->pin_cap = 0;
    }
    const size_t want = std::max(bytes, (size_t)1 << 18);
    HIP_TRY(hipHostMalloc((void **)&pool->pin, want, hipHostMallocDefault));
    pool->pin_cap = want;
    return NVK_OK;
}

// A buffer that has to grow is freed and allocated again: never while `stream` is being captured into a graph
// (hipFree / hipMalloc / a device-wide wait are not capturable, and the graph would keep the old address).
// Refused before anything is touched, with the remedy in the message.
int refuse_growth_under_capture(hipStream_t stream, const char *what)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        return nvk::fail(NVK_ERR_INVALID_ARG,
                         "%s would have to grow while the stream is being captured: size it first, outside the capture "
                         "(nvk_pool_reserve_workspace, or one direct call of the same shape)", what);
    return NVK_OK;
}

int ensure_workspace(nvk_pool *pool, size_t o_floats, size_t ml_floats, size_t arrival_ints, hipStream_t stream)
{
    if (o_floats <= pool->ws_o_floats && ml_floats <= pool->ws_ml_floats && arrival_ints <= pool->arrivals_cap)
        return NVK_OK;
    int rc = refuse_growth_under_capture(stream, "the split-K workspace");
    if (rc != NVK_OK)
        return rc;
    HIP_TRY(hipDeviceSynchronize()); // launches on any stream may still use the old buffers
    if (o_floats > pool->ws_o_floats) {
        if (pool->ws_o)
            HIP_TRY(hipFree(pool->ws_o));
        pool->ws_o = nullptr, pool->ws_o_floats = 0;
        HIP_TRY(hipMalloc(&pool->ws_o, o_floats * sizeof(float)));
        pool->ws_o_floats = o_floats;
        pool->ws_generation++;
    }
    if (ml_floats > pool->ws_ml_floats) {
        if (pool->ws_ml)
            HIP_TRY(hipFree(pool->ws_ml));
        pool->ws_ml = nullptr, pool->ws_ml_floats = 0;
        HIP_TRY(hipMalloc(&pool->ws_ml, ml_floats * sizeof(float)));
        pool->ws_ml_floats = ml_floats;
        pool->ws_generation++;
    }
    if (arrival_ints > pool->arrivals_cap) {
        if (pool->arrivals)
            HIP_TRY(hipFree(pool->arrivals));
        pool->arrivals = nullptr, pool->arrivals_cap = 0;
        const size_t want = std::max(arrival_ints, (size_t)4096);
        HIP_TRY(hipMalloc(&pool->arrivals, want * sizeof(int)));
        // (hipMemset on the null stream may return before the fill has run, and the pool's stream is non-blocking:
        // it does not wait for the null stream.  The first launch must find zeros: fill, then wait for the device.)
        HIP_TRY(hipMemset(pool->arrivals, 0, want * sizeof(int)));
        HIP_TRY(hipDeviceSynchronize());
        pool->arrivals_cap = want;
        pool->ws_generation++;
    }
    return NVK_OK;
}

// inv_freq[p] = 1 / powf(theta, 2p / head_dim): the reference's own expression (ops/positional.hpp:16,
// batch_ops.hpp:34) through the host libm, rebuilt only when theta changes.  The device buffer exists from
// pool creation (theta 10000, the reference's config.rope_theta), so the RoPE entry points neither allocate
// nor copy synchronously on their hot path -- they can run under stream capture.
int ensure_rope_table(nvk_pool *pool, float rope_theta)
{
    if (pool->rope_inv_freq && pool->rope_theta == rope_theta)
        return NVK_OK;
    const int          hd = pool->head_dim, half = hd / 2;
    std::vector<float> freq((size_t)std::max(half, 1));
    for (int i = 0; i + 1 < hd; i += 2)
        freq[(size_t)(i / 2)] = 1.0f / powf(rope_theta, (float)i / hd);
    if (!pool->rope_inv_freq)
        HIP_TRY(hipMalloc(&pool->rope_inv_freq, freq.size() * sizeof(float)));
    HIP_TRY(hipDeviceSynchronize()); // an earlier launch on another stream may still read the old table
    HIP_TRY(hipMemcpy(pool->rope_inv_freq, freq.data(), freq.size() * sizeof(float), hipMemcpyHostToDevice));
    pool->rope_theta = rope_theta;
    return NVK_OK;
}

int check_pool_layer(const nvk_pool *pool, int layer, const char *who)
{
    if (!pool)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: null pool", who);
    if (layer < 0 || layer >= pool->n_layers)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: layer %d outside [0,%d)", who, layer, pool->n_layers);
    return NVK_OK;
}

#ifdef NVK_STAMPS
unsigned long long *g_debug_stamps = nullptr; // diagnostic build only: [workgroups, 16] device buffer
#endif

// ------------------------------------------------------------ launch plan
struct LaunchShape
{
    int chunk_tokens = 0; // multiple of kStepTokens
    int max_chunks   = 1;
    int waves        = 1; // kv-heads per workgroup (waves along the kv axis)
    int parts        = 1; // token-split factor: waves * parts waves per workgroup
    int merge_off    = 0; // int offset of the token-split merge area in LDS
    int chunk_skew   = 0; // tokens an even XCD's full chunk hands to the odd XCD's behind it (long chunks; enqueue_planned)
    int kv_groups    = 1;
    int qpk          = 1; // q-heads per wave
    int vpl          = 1;
    bool full        = false;
    bool mfma        = false; // nvk_decode_mfma_kernel: one wave carries the whole GQA group on 16-token MFMA tiles
    int  kpw         = 1;     // kv-heads per wave: 2 where a row is an odd number of half-lines (head_dim 16 / 48)
    bool wide        = false; // nvk_decode_wide_kernel: 16-token steps (long chunks of whole 128-float rows, no token parts)
    bool row_pairs   = false; // NVK_DECODE_ROW_PAIRS: nvk_decode_rowpairs_kernel (128-float rows, kv-head pairs, 4 waves)
    bool free_forced = false; // NVK_DECODE_FREE_RUNNING: nvk_decode_free_kernel (no lockstep, 8-token steps) wherever it exists
    size_t lds_bytes = 0;
    int  qgroups     = 1;     // (kv-head group, q sub-group) pairs per item = arrival counters per row
    bool fuse_merge  = false; // max_chunks > 1: the decode kernel combines a row's chunks itself
    bool force_serial_merge = false; // NVK_DECODE_MERGE_SEPARATE: nvk_merge_kernel whatever the shape
};

// flags = mode (low byte) | modifiers
bool decode_flags_ok(int flags)
{
    const int mode = flags & 0xff, mods = flags & ~0xff;
    if (mode != NVK_DECODE_AUTO && mode != NVK_DECODE_SINGLE_PASS && mode != NVK_DECODE_SPLIT_K)
        return false;
    if (mods & ~(NVK_DECODE_MERGE_IN_KERNEL | NVK_DECODE_MERGE_SEPARATE | NVK_DECODE_UNIFORM | NVK_DECODE_NO_PLAN
                 | NVK_DECODE_VECTOR_ONLY | NVK_DECODE_NO_PAIRS | NVK_DECODE_FREE_RUNNING | NVK_DECODE_ROW_PAIRS))
        return false;
    return !((mods & NVK_DECODE_MERGE_IN_KERNEL) && (mods & NVK_DECODE_MERGE_SEPARATE));
}

constexpr int kMaxTableSliceEntries = 8192; // 32 KiB of LDS per workgroup at most
constexpr int kLdsPerLaunch = 64 * 1024;    // dynamic LDS a launch may ask for without hipFuncSetAttribute

int make_shape(const nvk_pool *pool, int num_seqs, int n_heads, int max_ctx, int flags, int chunk_tokens,
               LaunchShape *plan)
{
    const int hd   = pool->head_dim;
    const int mods = flags & ~0xff;
    flags &= 0xff;
    if (n_heads <= 0 || n_heads % pool->n_kv_heads != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "n_heads=%d is not a positive multiple of n_kv_heads=%d", n_heads,
                         pool->n_kv_heads);
    if (hd % 4 != 0 || hd > 256)
        return nvk::fail(NVK_ERR_INVALID_ARG, "head_dim=%d unsupported (need a multiple of 4, at most 256)", hd);
    const int kv_mul = n_heads / pool->n_kv_heads;
    plan->vpl        = hd <= 64 ? 1 : (hd <= 128 ? 2 : 4);
    plan->full       = hd == 64 * plan->vpl;
    // q-heads per wave: the whole GQA group when it has at most 8 heads (K/V then cross HBM and L2 once),
    // else its largest divisor up to 8 (the q sub-groups of an item re-read K/V; the grid pairs them on one XCD)
    // (register budget: 256-float rows stop at 6 q-heads; 128-float rows take 7 or 8 with one-token tiles)
    plan->qpk = 1;
    for (int d = std::min(kv_mul, plan->vpl == 4 ? std::min(NVK_MAX_QPK, 6) : NVK_MAX_QPK); d >= 1; d--)
        if (kv_mul % d == 0) {
            plan->qpk = d;
            break;
        }
    plan->mfma = plan->full && (hd == 64 || hd == 128) && kv_mul >= NVK_MFMA_MIN_KVMUL && kv_mul <= 16
                 && !(mods & NVK_DECODE_VECTOR_ONLY);
    if (plan->mfma)
        plan->qpk = kv_mul;
    // whole 128-float rows as kv-head pairs in 4-wave workgroups, two to a CU (asked for by the caller: short ragged batches)
    plan->row_pairs = (mods & NVK_DECODE_ROW_PAIRS) && !plan->mfma && plan->vpl == 2 && plan->full && plan->qpk <= 4
                      && plan->qpk == kv_mul && pool->n_kv_heads % 2 == 0 && pool->n_kv_heads <= 8;
    const int wg_waves = plan->mfma ? 8 : plan->row_pairs ? 4 : std::min(NVK_WG_WAVES, nvk::wg_waves(plan->vpl, plan->qpk, plan->full));
    // rows of 64 or 192 B (head_dim 16 / 48) share their 128-B lines with the neighbour kv-head: a wave then takes
    // the PAIR of heads (whole lines per token, each requested once); needs token rows that start on a line
    plan->kpw = (NVK_KV_PAIRS && plan->vpl == 1 && hd % 32 == 16 && pool->n_kv_heads % 2 == 0 && plan->qpk <= 4
                 && !(mods & NVK_DECODE_NO_PAIRS))
                    ? 2
                    : 1;
    if (plan->row_pairs)
        plan->kpw = 2;
    const int kv_slots = pool->n_kv_heads / plan->kpw; // waves along the kv axis in all
    plan->waves      = std::min(kv_slots, wg_waves);
    plan->kv_groups  = (kv_slots + plan->waves - 1) / plan->waves;
    // fewer kv-heads than a workgroup has waves (a head-sharded rank, a small model): the other waves split
    // the chunk's tokens, so the workgroup still has all its waves' loads in flight
    plan->parts      = plan->kv_groups == 1 ? std::max(1, wg_waves / plan->waves) : 1;

    const int step    = plan->mfma ? 16 : nvk::step_tokens(plan->vpl, plan->qpk) / plan->kpw;
    const int ctx_cap = round_up(std::max(max_ctx, 1), step);
    // token-split workgroups keep one (m, l, o) record per wave and q-head in LDS behind the table slice (the MFMA
    // kernel's part 0 merges from registers and writes none: parts - 1 records per kv-head there)
    size_t merge_bytes = 0;
    if (plan->parts > 1)
        merge_bytes = (size_t)plan->waves * (plan->mfma ? plan->parts - 1 : plan->parts) * plan->kpw * plan->qpk
                      * (4 + plan->vpl * 64) * sizeof(float);
    // a workgroup stages its chunk's table slice in LDS: bound the chunk by that, and slice + merge area by the
    // 64 KiB a launch gets without opting in (16 q-heads x 128 floats x 7 records are 58 KiB on their own)
    const long long slice_entries =
        std::min<long long>(kMaxTableSliceEntries, ((long long)kLdsPerLaunch - (long long)merge_bytes) / 4 - 8);
    if (slice_entries < 4)
        return nvk::fail(NVK_ERR_INVALID_ARG, "decode: %zu bytes of token-split records do not fit the workgroup's LDS",
                         merge_bytes);
    const long long slice_cap = std::max<long long>(step, slice_entries * pool->block_size / step * step);

    long long chunk = 0;
    if (flags == NVK_DECODE_SINGLE_PASS) {
        chunk = ctx_cap;
    }
    else if (flags == NVK_DECODE_SPLIT_K && chunk_tokens > 0) {
        chunk = round_up(chunk_tokens, step);
    }
    else {
        // AUTO (and SPLIT_K with no size).  Measured (profiles/r01_*): a workgroup should stream as
        // long a chunk as it can while the grid still covers every CU about once (8 waves at
        // 168-192 VGPRs = one workgroup per CU); with the longest-first item order a ragged
        // batch tolerates chunks up to 2048 tokens; 128 keeps the prologue/epilogue share small.
        const long long wg_per_chunk = std::max<long long>(1, (long long)plan->kv_groups * (kv_mul / plan->qpk));
        const long long tokens_upper = (long long)num_seqs * ctx_cap * wg_per_chunk;
        // few (row, head) pairs: the merge takes its chunks side by side (nvk_merge_wide_kernel), so chunks may be
        // shorter and more numerous -- the only way one request reaches more than a handful of CUs
        const bool wide      = (long long)num_seqs * pool->n_kv_heads * kv_mul <= NVK_MERGE_WIDE_ROWS;
        const int  min_chunk = wide ? NVK_AUTO_MIN_CHUNK_WIDE : NVK_AUTO_MIN_CHUNK;
        const int  most      = wide ? nvk::kMergeWideMaxChunks : 128;
        chunk = round_up((int)std::min<long long>(1 << 20, (tokens_upper + pool->num_cus - 1) / pool->num_cus), min_chunk);
        chunk = std::max<long long>(min_chunk, std::min<long long>(chunk, NVK_AUTO_MAX_CHUNK));
        // a bound on the chunks per sequence: nvk_merge_kernel walks a sequence's partials one after the other (one
        // long sequence alone, 32K tokens: 171 us with 256 chunks of 128, 134 us with 128 chunks of 256), and
        // nvk_merge_wide_kernel holds one chunk per thread
        chunk = std::max<long long>(chunk, round_up((ctx_cap + most - 1) / most, min_chunk));
        // (A/B knob, off: few rows cut into no more chunks than the decode kernel combines itself, so that the merge
        // launch goes away -- measured slower, see NVK_TINY_CHUNK_BYTES)
        if (wide && NVK_TINY_CHUNK_BYTES > 0 && !(mods & NVK_DECODE_MERGE_SEPARATE)) {
            const long long fused = round_up((ctx_cap + NVK_FUSE_MAX_CHUNKS - 1) / NVK_FUSE_MAX_CHUNKS, step);
            if (fused > chunk && fused * 2 * pool->row_len() * (long long)sizeof(float) <= NVK_TINY_CHUNK_BYTES)
                chunk = fused;
        }
    }
    chunk = std::min<long long>(chunk, ctx_cap);
    chunk = std::min<long long>(chunk, slice_cap);
    chunk = std::max<long long>(chunk, step);
    plan->chunk_tokens = (int)chunk;
    plan->max_chunks   = (ctx_cap + plan->chunk_tokens - 1) / plan->chunk_tokens;
    if (plan->max_chunks > 32767)
        return nvk::fail(NVK_ERR_INVALID_ARG, "chunk_tokens=%d splits max_ctx=%d into more than 32767 chunks",
                         plan->chunk_tokens, max_ctx);
    if ((unsigned long long)max_ctx * (unsigned)pool->block_size >= (1ULL << 32))
        return nvk::fail(NVK_ERR_INVALID_ARG, "max_ctx=%d too long for block_size=%d", max_ctx, pool->block_size);
    // long chunks of whole 128-float rows that the workgroup's waves walk together: 16-token steps (measured: config 3
    // 0.84 -> 0.86 of the HBM peak, config 4 +2 %; 512-token chunks and token-split shapes do better with 8)
    plan->free_forced = (mods & NVK_DECODE_FREE_RUNNING) != 0;
    plan->wide = NVK_WIDE_STEPS && !plan->free_forced && !plan->mfma && plan->kpw == 1 && plan->vpl == 2 && plan->full && plan->qpk <= 4
                 && plan->kv_groups == 1 && plan->parts == 1 && plan->chunk_tokens >= NVK_WIDE_MIN_CHUNK
                 && plan->chunk_tokens % 16 == 0;
    // Long chunks, one workgroup per item whose id is the item's place in the list (so the place's parity is the XCD's):
    // every even XCD's full chunk hands a few tokens (whole steps) to the odd XCD's chunk behind it -- the even XCDs of
    // an MI355X stream 3-6 % slower than the odd ones and a launch ends with the slowest.  Measured: config 4 0.86-0.87
    // -> 0.88-0.89 of the HBM peak; 512-token chunks gain nothing (profiles/r03_xcd_balance.log).  The table slice
    // in LDS gets room for the longer chunks.  Only where it measured better: workgroups that stream whole 4-KiB token
    // rows (8 kv-heads x 128 floats: the wide kernel, and the MFMA kernel at 64/8/128: +1.3 %); token-split workgroups
    // (1-4 kv-heads) and 64-float rows LOSE 1-2.5 % to it (same log: NVK_XCD_SKEW_ALL_SHAPES).
    plan->chunk_skew = 0;
    {
        int permille = NVK_XCD_SKEW_PERMILLE, min_chunk = NVK_XCD_SKEW_MIN_CHUNK;
#ifdef NVK_XCD_RANGES_ENV // A/B build
        if (const char *e = getenv("NVK_XCD_SKEW"))
            permille = atoi(e);
        if (const char *e = getenv("NVK_XCD_SKEW_MIN_CHUNK"))
            min_chunk = atoi(e);
#endif
        if (permille > 0 && plan->kv_groups == 1 && plan->qpk == kv_mul && plan->chunk_tokens >= min_chunk
            && (NVK_XCD_SKEW_ALL_SHAPES || plan->wide || (plan->mfma && plan->parts == 1 && hd == 128))) {
            const int step = plan->wide || plan->mfma ? 16 : 8, room = plan->chunk_tokens / 16;
            plan->chunk_skew = std::min((int)((long long)plan->chunk_tokens * permille / 1000 + step / 2) / step * step, room / step * step);
        }
    }
    plan->merge_off    = ((plan->chunk_tokens + plan->chunk_skew) / pool->block_size + 2 + 3) / 4 * 4; // 16-B aligned behind the table slice
    plan->lds_bytes    = (size_t)plan->merge_off * sizeof(int) + merge_bytes;
    if (plan->chunk_skew > 0 && plan->lds_bytes > (size_t)kLdsPerLaunch) { // (a chunk as long as the LDS allows: no room to skew)
        plan->chunk_skew = 0;
        plan->merge_off  = (plan->chunk_tokens / pool->block_size + 2 + 3) / 4 * 4;
        plan->lds_bytes  = (size_t)plan->merge_off * sizeof(int) + merge_bytes;
    }
    if (plan->lds_bytes > (size_t)kLdsPerLaunch) // (cannot happen: the chunk was bounded for it above)
        return nvk::fail(NVK_ERR_INVALID_ARG, "decode: the launch needs %zu bytes of LDS, more than %d", plan->lds_bytes,
                         kLdsPerLaunch);
    plan->qgroups            = plan->kv_groups * (kv_mul / plan->qpk);
    plan->force_serial_merge = (mods & NVK_DECODE_MERGE_SEPARATE) != 0;
    plan->fuse_merge         = plan->max_chunks > 1 && !plan->force_serial_merge
                       && ((mods & NVK_DECODE_MERGE_IN_KERNEL) || plan->max_chunks <= NVK_FUSE_MAX_CHUNKS);
    return NVK_OK;
}

// Which of the two vector kernels a shape launches: nvk_decode_kernel keeps the workgroup's waves in step and is for
// workgroups that hold every kv-head of their token rows; several kv-head groups (and 256-float rows, whose code is
// the same in both) run nvk_decode_free_kernel / nvk_decode_kernel without it.
// (6 q-heads over 128-float rows stay with nvk_decode_kernel, whose barrier is then off at run time: the free-running
// instantiation of that one shape comes out of the register allocator with 4 registers in scratch.)
constexpr bool has_free_kernel(int vpl, int qpk) { return vpl < 4 && !(vpl == 2 && qpk == 6); }
bool free_running(const LaunchShape &plan)
{
    return (plan.kv_groups > 1 || plan.free_forced) && has_free_kernel(plan.vpl, plan.qpk);
}

template <int VPL, int QPK> hipError_t launch_decode_full(bool full, bool free_run, dim3 grid, dim3 block, size_t lds,
                                                         hipStream_t stream, const nvk::DecodeParams &p)
{
    if constexpr (has_free_kernel(VPL, QPK)) {
        if (free_run) { // several kv-head groups, or NVK_DECODE_FREE_RUNNING: see free_running
            if (full)
                hipLaunchKernelGGL((nvk::nvk_decode_free_kernel<VPL, QPK, true>), grid, block, lds, stream, p);
            else
                hipLaunchKernelGGL((nvk::nvk_decode_free_kernel<VPL, QPK, false>), grid, block, lds, stream, p);
            return hipGetLastError();
        }
    }
    if (full)
        hipLaunchKernelGGL((nvk::nvk_decode_kernel<VPL, QPK, true>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((nvk::nvk_decode_kernel<VPL, QPK, false>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <int VPL> hipError_t launch_decode_qpk(int qpk, bool full, bool free_run, dim3 grid, dim3 block, size_t lds,
                                                hipStream_t stream, const nvk::DecodeParams &p)
{
    if constexpr (VPL <= 2) {
        if (qpk == 8)
            return launch_decode_full<VPL, 8>(full, free_run, grid, block, lds, stream, p);
        if (qpk == 7)
            return launch_decode_full<VPL, 7>(full, free_run, grid, block, lds, stream, p);
    }
    switch (qpk) {
    case 6: return launch_decode_full<VPL, 6>(full, free_run, grid, block, lds, stream, p);
    case 5: return launch_decode_full<VPL, 5>(full, free_run, grid, block, lds, stream, p);
    case 4: return launch_decode_full<VPL, 4>(full, free_run, grid, block, lds, stream, p);
    case 3: return launch_decode_full<VPL, 3>(full, free_run, grid, block, lds, stream, p);
    case 2: return launch_decode_full<VPL, 2>(full, free_run, grid, block, lds, stream, p);
    default: return launch_decode_full<VPL, 1>(full, free_run, grid, block, lds, stream, p);
    }
}

hipError_t launch_decode(const LaunchShape &plan, dim3 grid, dim3 block, hipStream_t stream, const nvk::DecodeParams &p)
{
    if (plan.row_pairs) { // make_shape: vpl 2, full, at most 4 q-heads per wave, kv-head pairs
        switch (plan.qpk) {
        case 4: hipLaunchKernelGGL((nvk::nvk_decode_rowpairs_kernel<4>), grid, block, plan.lds_bytes, stream, p); break;
        case 3: hipLaunchKernelGGL((nvk::nvk_decode_rowpairs_kernel<3>), grid, block, plan.lds_bytes, stream, p); break;
        case 2: hipLaunchKernelGGL((nvk::nvk_decode_rowpairs_kernel<2>), grid, block, plan.lds_bytes, stream, p); break;
        default: hipLaunchKernelGGL((nvk::nvk_decode_rowpairs_kernel<1>), grid, block, plan.lds_bytes, stream, p); break;
        }
        return hipGetLastError();
    }
    if (plan.kpw == 2) { // make_shape: vpl 1, not full, at most 4 q-heads per wave
        switch (plan.qpk) {
        case 4: hipLaunchKernelGGL((nvk::nvk_decode_pairs_kernel<4>), grid, block, plan.lds_bytes, stream, p); break;
        case 3: hipLaunchKernelGGL((nvk::nvk_decode_pairs_kernel<3>), grid, block, plan.lds_bytes, stream, p); break;
        case 2: hipLaunchKernelGGL((nvk::nvk_decode_pairs_kernel<2>), grid, block, plan.lds_bytes, stream, p); break;
        default: hipLaunchKernelGGL((nvk::nvk_decode_pairs_kernel<1>), grid, block, plan.lds_bytes, stream, p); break;
        }
        return hipGetLastError();
    }
    if (plan.mfma) {
        if (plan.vpl == 1)
            hipLaunchKernelGGL((nvk::nvk_decode_mfma_kernel<64>), grid, block, plan.lds_bytes, stream, p);
        else
            hipLaunchKernelGGL((nvk::nvk_decode_mfma_kernel<128>), grid, block, plan.lds_bytes, stream, p);
        return hipGetLastError();
    }
    if (plan.wide) { // make_shape: vpl 2, full, at most 4 q-heads per wave, one kv-head group, no token parts
        switch (plan.qpk) {
        case 4: hipLaunchKernelGGL((nvk::nvk_decode_wide_kernel<4>), grid, block, plan.lds_bytes, stream, p); break;
        case 3: hipLaunchKernelGGL((nvk::nvk_decode_wide_kernel<3>), grid, block, plan.lds_bytes, stream, p); break;
        case 2: hipLaunchKernelGGL((nvk::nvk_decode_wide_kernel<2>), grid, block, plan.lds_bytes, stream, p); break;
        default: hipLaunchKernelGGL((nvk::nvk_decode_wide_kernel<1>), grid, block, plan.lds_bytes, stream, p); break;
        }
        return hipGetLastError();
    }
    switch (plan.vpl) {
    case 1: return launch_decode_qpk<1>(plan.qpk, plan.full, free_running(plan), grid, block, plan.lds_bytes, stream, p);
    case 2: return launch_decode_qpk<2>(plan.qpk, plan.full, free_running(plan), grid, block, plan.lds_bytes, stream, p);
    default: return launch_decode_qpk<4>(plan.qpk, plan.full, free_running(plan), grid, block, plan.lds_bytes, stream, p);
    }
}

} // namespace

// A decode plan: the batch's work items (built on the device from its context
// lengths) plus the launch shape.  Depends on the lengths only, so one plan
// serves every layer of a model step.
struct nvk_plan
{
    nvk_pool     *pool = nullptr;
    int           num_seqs = 0, n_heads = 0, max_ctx = 0;
    LaunchShape   shape;
    bool          built = false;
    nvk::WorkItem *items = nullptr;
    int          *counters = nullptr; // [0] = n_items, [1 + s] = chunks of sequence s
    size_t        cap_items = 0, cap_seqs = 0;
    int           item_bound = 0; // grid.x: at least as many workgroups as items
    long long     generation = 0; // bumped whenever items / counters move (a captured graph holds their addresses)
    // direct indexing (NVK_DECODE_UNIFORM / NVK_DECODE_NO_PLAN): no work list; the kernels read the lengths themselves
    bool          direct = false;
    const int    *ctx_lens_dev = nullptr; // direct: the caller's lengths (null: every row holds uniform_ctx tokens)
    int           uniform_ctx = 0;
};

namespace {

int plan_reserve(nvk_plan *plan, size_t items, size_t seqs, hipStream_t stream)
{
    if (items > plan->cap_items || seqs > plan->cap_seqs) {
        int rc = refuse_growth_under_capture(stream, "the decode plan");
        if (rc != NVK_OK)
            return rc;
    }
    if (items > plan->cap_items) {
        if (plan->items) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(plan->items));
            plan->items     = nullptr;
            plan->cap_items = 0;
        }
        const size_t want = std::max(items, (size_t)1024);
        HIP_TRY(hipMalloc(&plan->items, want * sizeof(nvk::WorkItem)));
        plan->cap_items = want;
        plan->generation++;
    }
    if (seqs > plan->cap_seqs) {
        if (plan->counters) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(plan->counters));
            plan->counters = nullptr;
            plan->cap_seqs = 0;
        }
        const size_t want = std::max(seqs, (size_t)1024);
        HIP_TRY(hipMalloc(&plan->counters, (want + 1) * sizeof(int)));
        plan->cap_seqs = want;
        plan->generation++;
    }
    return NVK_OK;
}

// The host half of a plan: launch shape and planning mode for a batch.  Touches no device memory.
int plan_prepare(nvk_plan *plan, int num_seqs, int n_heads, const int *ctx_dev, int max_ctx, int flags, int chunk_tokens)
{
    nvk_pool *pool = plan->pool;
    plan->built    = false;
    int rc         = make_shape(pool, num_seqs, n_heads, max_ctx, flags, chunk_tokens, &plan->shape);
    if (rc != NVK_OK)
        return rc;
    const LaunchShape &sh = plan->shape;
    if ((long long)num_seqs * sh.max_chunks > 0x7fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "num_seqs=%d x %d chunks exceeds the grid", num_seqs, sh.max_chunks);
    plan->num_seqs = num_seqs, plan->n_heads = n_heads, plan->max_ctx = max_ctx;
    plan->item_bound = num_seqs * sh.max_chunks;
    const int mods = flags & ~0xff;
    plan->direct       = (mods & (NVK_DECODE_UNIFORM | NVK_DECODE_NO_PLAN)) != 0;
    plan->ctx_lens_dev = (mods & NVK_DECODE_UNIFORM) ? nullptr : ctx_dev;
    plan->uniform_ctx  = max_ctx;
    if (num_seqs > 0 && plan->direct && !(mods & NVK_DECODE_UNIFORM) && !ctx_dev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "decode: NVK_DECODE_NO_PLAN needs the lengths on the device");
    return NVK_OK;
}

// The device half: the work list, written into plan->items / plan->counters by one small kernel on `stream`
// (nothing to do for a plan-free launch).
int plan_launch(nvk_plan *plan, const int *ctx_dev, hipStream_t stream)
{
    if (!plan->direct && plan->num_seqs > 0) {
        hipLaunchKernelGGL(nvk::nvk_plan_kernel, dim3(1), dim3(nvk::kPlanThreads), 0, stream, ctx_dev, plan->num_seqs,
                           plan->shape.chunk_tokens, plan->shape.max_chunks, plan->items, plan->counters,
                           plan->counters + 1, plan->shape.chunk_skew);
        HIP_TRY(hipGetLastError());
    }
    plan->built = true;
    return NVK_OK;
}

// Builds `plan` for a batch on `stream`, growing the plan's and the pool's buffers where needed.
int build_plan(nvk_plan *plan, int num_seqs, int n_heads, const int *ctx_dev, int max_ctx, int flags, int chunk_tokens,
               hipStream_t stream)
{
    nvk_pool *pool = plan->pool;
    int       rc   = plan_prepare(plan, num_seqs, n_heads, ctx_dev, max_ctx, flags, chunk_tokens);
    if (rc != NVK_OK)
        return rc;
    const LaunchShape &sh = plan->shape;
    if (num_seqs == 0) {
        plan->built = true;
        return NVK_OK;
    }
    if (!plan->direct) {
        rc = plan_reserve(plan, (size_t)plan->item_bound, (size_t)num_seqs, stream);
        if (rc != NVK_OK)
            return rc;
    }
    if (sh.max_chunks > 1) {
        rc = ensure_workspace(pool, (size_t)num_seqs * sh.max_chunks * n_heads * pool->head_dim,
                              (size_t)num_seqs * sh.max_chunks * n_heads * 2,
                              sh.fuse_merge ? (size_t)num_seqs * sh.qgroups : 0, stream);
        if (rc != NVK_OK)
            return rc;
    }
    return plan_launch(plan, ctx_dev, stream);
}

// Where a launch keeps what it needs besides its inputs and outputs: the split-K partials and the arrival
// counters.  Normally the pool's own buffers (grown on demand); nvk_decode_device_noalloc brings slices of
// the library's static device array instead.
struct Workspace
{
    float *ws_o = nullptr, *ws_ml = nullptr;
    int   *arrivals = nullptr;
    size_t o_floats = 0, ml_floats = 0, arrival_ints = 0;
};

Workspace pool_workspace(const nvk_pool *pool)
{
    Workspace w;
    w.ws_o = pool->ws_o, w.ws_ml = pool->ws_ml, w.arrivals = pool->arrivals;
    w.o_floats = pool->ws_o_floats, w.ml_floats = pool->ws_ml_floats, w.arrival_ints = pool->arrivals_cap;
    return w;
}

// Enqueue one layer-call of a built plan on `stream`.  All pointers are device pointers.
// ev_attn_begin/ev_attn_end (optional) bracket the attention kernel alone.
int enqueue_planned(nvk_pool *pool, int layer, const nvk_plan *plan, const float *q_dev, const int *tables_dev,
                    int max_blocks, float *out_dev, hipStream_t stream, hipEvent_t ev_attn_begin,
                    hipEvent_t ev_attn_end, const int *table_rows_dev = nullptr, const Workspace *given = nullptr)
{
    const Workspace ws = given ? *given : pool_workspace(pool);
    if (!plan->built || plan->pool != pool)
        return nvk::fail(NVK_ERR_INVALID_ARG, "decode: the plan was not built for this pool");
    if (plan->num_seqs == 0)
        return NVK_OK;
    const LaunchShape &sh = plan->shape;
    if ((long long)max_blocks * pool->block_size < plan->max_ctx)
        return nvk::fail(NVK_ERR_INVALID_ARG, "max_blocks=%d x block_size=%d cannot hold max_ctx=%d", max_blocks,
                         pool->block_size, plan->max_ctx);
    const int n_heads = plan->n_heads;
    const int kv_mul  = n_heads / pool->n_kv_heads;
    if (sh.max_chunks > 1
        && (ws.o_floats < (size_t)plan->num_seqs * sh.max_chunks * n_heads * pool->head_dim
            || ws.ml_floats < (size_t)plan->num_seqs * sh.max_chunks * n_heads * 2
            || (sh.fuse_merge && ws.arrival_ints < (size_t)plan->num_seqs * sh.qgroups)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "decode: split-K workspace smaller than the plan needs");

    nvk::DecodeParams p;
    p.k_layer      = pool->k_layer(layer);
    p.v_layer      = pool->v_layer(layer);
    p.q            = q_dev;
    p.block_tables = tables_dev;
    p.table_rows   = table_rows_dev;
    p.items        = plan->items;
    p.n_items      = plan->counters;
    p.out          = out_dev;
    p.ws_o         = ws.ws_o;
    p.ws_ml        = ws.ws_ml;
    p.max_blocks   = max_blocks;
    p.num_blocks   = pool->num_blocks;
    p.block_size   = pool->block_size;
    p.bs_magic     = pool->bs_magic;
    p.n_kv_heads   = pool->n_kv_heads;
    p.n_heads      = n_heads;
    p.kv_mul       = kv_mul;
    p.head_dim     = pool->head_dim;
    p.max_chunks   = sh.max_chunks;
    p.kv_waves     = sh.waves;
    p.merge_off    = sh.merge_off;
    p.q_scale      = 1.44269504088896340736f / sqrtf((float)pool->head_dim);
    p.direct       = plan->direct ? 1 : 0;
    p.num_seqs     = plan->num_seqs;
    p.chunk_tokens = sh.chunk_tokens;
    p.uniform_ctx  = plan->uniform_ctx;
    p.ctx_lens     = plan->ctx_lens_dev;
    p.arrivals     = sh.fuse_merge ? ws.arrivals : nullptr;
    // (only where a workgroup's waves are the kv-heads of whole token rows: with token-split waves -- a head-sharded
    // rank's 512-B rows -- the same dealing measured 0.61-0.76 against 0.81, profiles/r02_sweep_item_order.log)
    p.xcd_ranges = (NVK_XCD_RANGES && plan->direct && !plan->ctx_lens_dev && kv_mul == sh.qpk && sh.parts == 1) ? 1 : 0;
#ifdef NVK_XCD_RANGES_ENV // A/B build
    if (getenv("NVK_XCD_RANGES_OFF"))
        p.xcd_ranges = 0;
    if (const char *e = getenv("NVK_XCD_GROUP")) // the dealing alone: XCDs in pairs or fours
        if (p.xcd_ranges && (atoi(e) == 2 || atoi(e) == 4) && ((long long)plan->num_seqs * sh.max_chunks) % 8 == 0)
            p.xcd_ranges = atoi(e);
#endif
    // LaunchShape::chunk_skew without a plan: every chunk full, an even number of them per sequence, whole rounds of the
    // 8 XCDs -- the XCDs take their ranges in pairs, even chunks on even XCDs (with a plan the work list carries the
    // chunks' bounds: nvk::skewed_chunk)
    // (token-split shapes keep item = workgroup id, which puts even chunks on even XCDs by itself)
    p.chunk_skew = 0;
    if (plan->direct && !plan->ctx_lens_dev && sh.chunk_skew > 0 && sh.max_chunks % 2 == 0
        && plan->uniform_ctx == sh.max_chunks * sh.chunk_tokens && ((long long)plan->num_seqs * sh.max_chunks) % 8 == 0) {
        p.chunk_skew = sh.chunk_skew;
        if (p.xcd_ranges)
            p.xcd_ranges = 2;
    }
#ifdef NVK_STAMPS
    p.stamps = g_debug_stamps;
#endif

    const int       qsplit = kv_mul / sh.qpk;
    const int       padded = qsplit > 1 ? round_up(plan->item_bound, 8) : plan->item_bound;
    const long long wgs    = (long long)padded * qsplit * sh.kv_groups;
    if (wgs > 0x7fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "decode: %lld workgroups exceed the grid", wgs);
    p.items_padded = padded;
    const dim3 grid((unsigned)wgs);
    const dim3 block((unsigned)(sh.waves * sh.parts) * 64);
    if (ev_attn_begin)
        HIP_TRY(hipEventRecord(ev_attn_begin, stream));
    HIP_TRY(launch_decode(sh, grid, block, stream, p));
    if (ev_attn_end)
        HIP_TRY(hipEventRecord(ev_attn_end, stream));

    if (sh.max_chunks > 1 && !sh.fuse_merge) {
        // The combine as a kernel of its own.  Few rows cut into many chunks: one workgroup per (row, head), chunks
        // side by side; otherwise a thread per output float4 walking its chunks (NVK_MERGE_WIDE_ROWS = 0 switches
        // the first form off, NVK_DECODE_MERGE_SEPARATE asks for the second).  Without a plan the kernels work
        // the chunk counts out of the lengths themselves.
        const int *seq_chunks = plan->direct ? nullptr : plan->counters + 1;
        if (!sh.force_serial_merge && sh.max_chunks >= 4 && sh.max_chunks <= nvk::kMergeWideMaxChunks && n_heads <= 65535
            && (long long)plan->num_seqs * n_heads <= NVK_MERGE_WIDE_ROWS) {
            const dim3 mgrid((unsigned)plan->num_seqs, (unsigned)n_heads);
            hipLaunchKernelGGL(nvk::nvk_merge_wide_kernel, mgrid, dim3(256), 0, stream, ws.ws_o, ws.ws_ml,
                               seq_chunks, plan->ctx_lens_dev, plan->uniform_ctx, sh.chunk_tokens, out_dev, n_heads,
                               pool->head_dim, sh.max_chunks);
        }
        else {
            const int  vecs = n_heads * (pool->head_dim / 4);
            const dim3 mgrid((unsigned)plan->num_seqs, (unsigned)((vecs + 255) / 256));
            hipLaunchKernelGGL(nvk::nvk_merge_kernel, mgrid, dim3(256), 0, stream, ws.ws_o, ws.ws_ml, seq_chunks,
                               plan->ctx_lens_dev, plan->uniform_ctx, sh.chunk_tokens, out_dev, n_heads, pool->head_dim,
                               sh.max_chunks);
        }
        HIP_TRY(hipGetLastError());
    }
    return NVK_OK;
}

int own_plan_of(nvk_pool *pool, nvk_plan **out)
{
    if (!pool->own_plan) {
        pool->own_plan = new (std::nothrow) nvk_plan();
        if (!pool->own_plan)
            return nvk::fail(NVK_ERR_OOM, "host allocation failed");
        pool->own_plan->pool = pool;
    }
    *out = pool->own_plan;
    return NVK_OK;
}

void free_plan(nvk_plan *plan)
{
    if (!plan)
        return;
    if (plan->items)
        (void)hipFree(plan->items);
    if (plan->counters)
        (void)hipFree(plan->counters);
    delete plan;
}

// plan + run, the shape of nvk_decode_device
int enqueue_decode(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev, const int *tables_dev,
                   int max_blocks, const int *ctx_dev, int max_ctx, float *out_dev, int flags, int chunk_tokens,
                   hipStream_t stream)
{
    if (num_seqs == 0)
        return NVK_OK;
    nvk_plan *plan = nullptr;
    int       rc   = own_plan_of(pool, &plan);
    if (rc != NVK_OK)
        return rc;
    rc = build_plan(plan, num_seqs, n_heads, ctx_dev, max_ctx, flags, chunk_tokens, stream);
    if (rc != NVK_OK)
        return rc;
    return enqueue_planned(pool, layer, plan, q_dev, tables_dev, max_blocks, out_dev, stream, nullptr, nullptr);
}

// ---- chunked prefill: descriptors -------------------------------------------
// Sequence s brings lens[s] new tokens at positions starts[s] + b; their q/k/v/out rows are packed back
// to back in sequence order.  Row r of the packed arrays is described by (row_seq, row_pos, row_ctx).
struct ChunkRows
{
    int        total_q = 0, max_ctx = 0, max_chunk = 0;
    const int *row_seq = nullptr, *row_pos = nullptr, *row_ctx = nullptr; // device pointers, [total_q] each
};

int ensure_desc(nvk_pool *pool, size_t ints)
{
    if (!pool->desc_ev)
        HIP_TRY(hipEventCreateWithFlags(&pool->desc_ev, hipEventDisableTiming));
    HIP_TRY(hipEventSynchronize(pool->desc_ev)); // the pinned buffer is ours again
    if (ints <= pool->desc_cap)
        return NVK_OK;
    if (pool->desc_host) {
        HIP_TRY(hipDeviceSynchronize()); // kernels may still read the old device mirror
        HIP_TRY(hipHostFree(pool->desc_host));
        HIP_TRY(hipFree(pool->desc_dev));
        pool->desc_host = pool->desc_dev = nullptr;
        pool->desc_cap                   = 0;
    }
    const size_t want = std::max(ints, (size_t)16384);
    HIP_TRY(hipHostMalloc((void **)&pool->desc_host, want * sizeof(int), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&pool->desc_dev, want * sizeof(int)));
    pool->desc_cap = want;
    return NVK_OK;
}

// Validates the chunk list (host arrays), builds the per-row descriptors and uploads them on `st`.
int stage_chunk_rows(nvk_pool *pool, int num_seqs, const int *starts, const int *lens, int max_blocks, const char *who,
                     hipStream_t st, ChunkRows *rows)
{
    long long total = 0;
    int       max_ctx = 0, max_chunk = 0;
    for (int s = 0; s < num_seqs; s++) {
        if (starts[s] < 0 || lens[s] < 0)
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: chunk %d has start %d, length %d", who, s, starts[s], lens[s]);
        const long long end = (long long)starts[s] + lens[s];
        if (end > (long long)max_blocks * pool->block_size)
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: chunk %d ends at token %lld, its table row holds %d blocks of %d",
                             who, s, end, max_blocks, pool->block_size);
        total += lens[s];
        if (lens[s] > 0)
            max_ctx = std::max(max_ctx, (int)end);
        max_chunk = std::max(max_chunk, lens[s]);
    }
    if (total > 0x3fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: %lld query tokens in one call", who, total);
    rows->total_q = (int)total, rows->max_ctx = max_ctx, rows->max_chunk = max_chunk;
    if (total == 0)
        return NVK_OK;
    int rc = ensure_desc(pool, (size_t)3 * total);
    if (rc != NVK_OK)
        return rc;
    int *seq = pool->desc_host, *pos = seq + total, *ctx = pos + total;
    int  r = 0;
    for (int s = 0; s < num_seqs; s++)
        for (int b = 0; b < lens[s]; b++, r++) {
            seq[r] = s;
            pos[r] = starts[s] + b;
            ctx[r] = starts[s] + b + 1; // query b attends to t <= start + b (model_chunked.hpp:186)
        }
    HIP_TRY(hipMemcpyAsync(pool->desc_dev, pool->desc_host, (size_t)3 * total * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(pool->desc_ev, st));
    rows->row_seq = pool->desc_dev, rows->row_pos = pool->desc_dev + total, rows->row_ctx = pool->desc_dev + 2 * total;
    return NVK_OK;
}

constexpr int kPrefillMaxTableEntries = 4096; // table slice a tiled-prefill workgroup stages in LDS (16 KiB)

// TILED prefill: work items = (sequence, tile of 32*qsubs query tokens) x key-range parts, longest first.
// Built on the host from the chunk list and uploaded on `st`.  When the call has fewer (tile, head-group)
// pairs than about twice the chip's workgroup slots, every tile's key range is cut into parts of `kc` keys
// (split-K) and *row_parts_dev gets, per packed query row, the number of parts to merge (1 = none).
struct PrefillWork
{
    const nvk::PrefillItem *items_dev = nullptr;
    const int              *row_parts_dev = nullptr;
    int                     n_items = 0, max_ctx = 0, max_parts = 1, total_q = 0;
};

int stage_prefill_items(nvk_pool *pool, int num_seqs, const int *starts, const int *lens, int max_blocks, int qsubs,
                        int grid_y, hipStream_t st, PrefillWork *work)
{
    const int tile = 32 * qsubs;
    long long tiles = 0, key_work = 0; // key_work: keys scanned, summed over tiles
    int       top = 0, total_q = 0;
    for (int s = 0; s < num_seqs; s++) {
        if (starts[s] < 0 || lens[s] < 0 || (long long)starts[s] + lens[s] > (long long)max_blocks * pool->block_size)
            return nvk::fail(NVK_ERR_INVALID_ARG, "prefill: chunk %d (start %d, length %d) does not fit its table row", s,
                             starts[s], lens[s]);
        for (int q0 = 0; q0 < lens[s]; q0 += tile) {
            tiles++;
            key_work += starts[s] + std::min(q0 + tile, lens[s]);
        }
        total_q += lens[s];
        if (lens[s] > 0)
            top = std::max(top, starts[s] + lens[s]);
    }
    work->total_q = total_q, work->max_ctx = top, work->n_items = 0, work->max_parts = 1;
    if (tiles == 0)
        return NVK_OK;
    // two 4-wave workgroups fit a CU; aim at about two rounds of them before cutting key ranges
    const long long slots = 2LL * pool->num_cus;
    int             kc    = 0x7fffffff;
    if (tiles * grid_y < 2 * slots) {
        const long long want = (key_work * grid_y + 2 * slots - 1) / (2 * slots);
        kc                   = (int)std::max<long long>(256, (want + 31) / 32 * 32);
    }
    std::vector<nvk::PrefillItem> items;
    std::vector<int>              row_parts((size_t)total_q, 1);
    int                           row = 0;
    for (int s = 0; s < num_seqs; s++) {
        for (int q0 = 0; q0 < lens[s]; q0 += tile) {
            const int nq = std::min(tile, lens[s] - q0), keys = starts[s] + q0 + nq;
            const int np = (int)(((long long)keys + kc - 1) / kc);
            for (int part = 0; part < np; part++)
                items.push_back(nvk::PrefillItem{s, row + q0, starts[s] + q0, nq, part * kc,
                                                 (int)std::min<long long>((long long)(part + 1) * kc, keys), part, np});
            if (np > 1)
                std::fill(row_parts.begin() + row + q0, row_parts.begin() + row + q0 + nq, np);
            work->max_parts = std::max(work->max_parts, np);
        }
        row += lens[s];
    }
    std::stable_sort(items.begin(), items.end(), [](const nvk::PrefillItem &a, const nvk::PrefillItem &b) {
        return std::min(a.k1, a.pos0 + a.nq) - a.k0 > std::min(b.k1, b.pos0 + b.nq) - b.k0; // longest key ranges first
    });
    work->n_items = (int)items.size();
    const size_t item_ints = items.size() * (sizeof(nvk::PrefillItem) / sizeof(int));
    const size_t rows_ints = work->max_parts > 1 ? (size_t)total_q : 0;
    int rc = ensure_desc(pool, item_ints + rows_ints);
    if (rc != NVK_OK)
        return rc;
    std::memcpy(pool->desc_host, items.data(), item_ints * sizeof(int));
    if (rows_ints)
        std::memcpy(pool->desc_host + item_ints, row_parts.data(), rows_ints * sizeof(int));
    HIP_TRY(hipMemcpyAsync(pool->desc_dev, pool->desc_host, (item_ints + rows_ints) * sizeof(int), hipMemcpyHostToDevice,
                           st));
    HIP_TRY(hipEventRecord(pool->desc_ev, st));
    work->items_dev     = reinterpret_cast<const nvk::PrefillItem *>(pool->desc_dev);
    work->row_parts_dev = rows_ints ? pool->desc_dev + item_ints : nullptr;
    return NVK_OK;
}

// Host-side check of every table entry the chunks reach (what the reference leaves undefined).
int validate_chunk_tables(const nvk_pool *pool, int num_seqs, const int *tables, int max_blocks, const int *starts,
                          const int *lens, const char *who)
{
    for (int s = 0; s < num_seqs; s++) {
        if (lens[s] <= 0)
            continue;
        const int nblk = (starts[s] + lens[s] + pool->block_size - 1) / pool->block_size;
        for (int i = 0; i < nblk; i++) {
            const int pb = tables[(size_t)s * max_blocks + i];
            if (pb < 0 || pb >= pool->num_blocks)
                return nvk::fail(NVK_ERR_BAD_BLOCK, "%s: block_tables[%d][%d]=%d outside [0,%d)", who, s, i, pb,
                                 pool->num_blocks);
        }
    }
    return NVK_OK;
}

// thread-local binding for the drop-in entry point
thread_local nvk_pool *t_bound_pool  = nullptr;
thread_local int       t_bound_layer = 0;
// (the unbound, host-slab form of the drop-in keeps a thread-local staging pool: ThreadSlabs below)


// ---- host-slab mirrors (NVK_HOST_CACHE_APPEND_ONLY) --------------------------
// The drop-in entry points get the engine's HOST slabs.  Re-staging the whole context on every call
// is always correct and PCIe-bound; but the reference engine only ever appends: between two calls
// for the same slab exactly one new token row is written (model.hpp:190-207) and num_tokens grows by
// one (model.hpp:397).  Under that promise a device mirror per slab keeps what it has and only the
// rows it has not seen cross PCIe.  Any call that is not "the previous call plus one token" on the
// same slab restages everything, so a new request (pos back to 0) is handled without being told.
struct HostMirror
{
    const float *k_host = nullptr, *v_host = nullptr;
    nvk_pool    *pool   = nullptr; // one layer; local block ids
    int          last_tokens = -1;
    bool         contiguous  = false; // mirror of a contiguous cache (nvk_standard_attention): one row counter
    std::unordered_map<int, int> local_of; // physical block id of the host slab -> block of `pool`
    std::vector<int>             rows_staged; // per local block
    // the promise's witness: a copy of the last K row staged and where it sits in the host slab.  A slab that was
    // freed and allocated again at the same address, or rewritten in place, almost surely no longer holds these
    // bytes there: then nothing staged is trusted (a cheap check, one row; it cannot prove the promise kept).
    std::vector<float> witness;
    size_t             witness_at = 0; // float offset into k_host
    // the block table of the last call: "the previous call plus one token" means the SAME sequence -- the same blocks in
    // the same order, at most one more.  (Several requests decoding in turn over one layer's slab, one of them new on
    // blocks another has just freed and holding exactly one token more than the call before: without this its blocks'
    // old rows would be trusted.)
    std::vector<int> last_table;
};

// The calling thread's device-side state behind the host-slab forms of the drop-in entry points; released when
// the thread ends (each mirror and the staging pool hold device memory).
struct ThreadSlabs
{
    nvk_pool               *scratch_pool = nullptr;
    std::vector<HostMirror> mirrors;
    ~ThreadSlabs();
};
thread_local ThreadSlabs t_slabs;
#define t_scratch_pool (t_slabs.scratch_pool)
#define t_mirrors (t_slabs.mirrors)

// The calling thread's one-layer staging pool for the host-slab forms of the
// drop-in entry points, (re)created when the geometry changes or it is too small.
int scratch_pool_for(int block_size, int n_kv_heads, int head_dim, int nblk, nvk_pool **out)
{
    nvk_pool *sp = t_scratch_pool;
    if (!sp || sp->block_size != block_size || sp->head_dim != head_dim || sp->n_kv_heads != n_kv_heads
        || sp->num_blocks < nblk) {
        if (sp)
            nvk_pool_destroy(sp);
        t_scratch_pool = nullptr;
        int dev        = 0;
        if (hipGetDevice(&dev) != hipSuccess)
            dev = 0;
        int rc = nvk_pool_create(1, std::max(2 * nblk, 64), block_size, n_kv_heads, head_dim, dev, &sp);
        if (rc != NVK_OK)
            return rc;
        t_scratch_pool = sp;
    }
    *out = sp;
    return NVK_OK;
}

thread_local int                     t_host_policy = NVK_HOST_CACHE_RESTAGE;
thread_local long long               t_staged_bytes = 0; // H2D bytes of K/V rows moved by the drop-in entry points

void drop_mirror(HostMirror &m)
{
    if (m.pool)
        nvk_pool_destroy(m.pool);
    m = HostMirror();
}

ThreadSlabs::~ThreadSlabs()
{
    for (HostMirror &m : mirrors)
        drop_mirror(m);
    mirrors.clear();
    if (scratch_pool)
        nvk_pool_destroy(scratch_pool);
    scratch_pool = nullptr;
}

// Does the host slab still hold the row the mirror staged last, where it staged it?
bool witness_holds(const HostMirror &m)
{
    return m.witness.empty()
           || std::memcmp(m.k_host + m.witness_at, m.witness.data(), m.witness.size() * sizeof(float)) == 0;
}

void keep_witness(HostMirror &m, size_t row_at, size_t row_floats)
{
    m.witness.assign(m.k_host + row_at, m.k_host + row_at + row_floats);
    m.witness_at = row_at;
}

// Finds or creates the mirror of (key_cache, value_cache) with room for `need_blocks` blocks.
int mirror_for(const float *key_cache, const float *value_cache, int block_size, int n_kv_heads, int head_dim,
               int need_blocks, bool contiguous, HostMirror **out)
{
    HostMirror *hit = nullptr;
    for (HostMirror &m : t_mirrors)
        if (m.k_host == key_cache)
            hit = &m;
    if (hit
        && (hit->v_host != value_cache || hit->contiguous != contiguous || hit->pool->block_size != block_size || hit->pool->n_kv_heads != n_kv_heads
            || hit->pool->head_dim != head_dim || hit->pool->num_blocks < need_blocks)) {
        drop_mirror(*hit); // geometry changed or it has outgrown its pool: start over (everything restaged)
        hit->k_host = nullptr;
    }
    if (!hit || !hit->pool) {
        const bool fresh = !hit;
        if (!hit) {
            if (t_mirrors.size() >= 256) { // a bound on what one thread may pin: recycle the oldest
                drop_mirror(t_mirrors.front());
                t_mirrors.erase(t_mirrors.begin());
            }
            t_mirrors.emplace_back();
            hit = &t_mirrors.back();
        }
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess)
            dev = 0;
        int rc = nvk_pool_create(1, std::max(2 * need_blocks, 64), block_size, n_kv_heads, head_dim, dev, &hit->pool);
        if (rc != NVK_OK) {
            if (fresh)
                t_mirrors.pop_back();
            return rc;
        }
        hit->k_host = key_cache, hit->v_host = value_cache, hit->last_tokens = -1, hit->contiguous = contiguous;
        hit->local_of.clear();
        hit->rows_staged.clear();
    }
    *out = hit;
    return NVK_OK;
}

} // namespace

extern "C" {

int nvk_set_host_cache_policy(int policy)
{
    if (policy != NVK_HOST_CACHE_RESTAGE && policy != NVK_HOST_CACHE_APPEND_ONLY)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_set_host_cache_policy: policy %d", policy);
    if (policy == NVK_HOST_CACHE_RESTAGE) {
        for (HostMirror &m : t_mirrors)
            drop_mirror(m);
        t_mirrors.clear();
    }
    t_host_policy = policy;
    return NVK_OK;
}

long long nvk_host_cache_staged_bytes(int reset)
{
    const long long v = t_staged_bytes;
    if (reset)
        t_staged_bytes = 0;
    return v;
}

const char *nvk_version(void) { return NVK_VERSION_STRING; }
const char *nvk_last_error(void) { return nvk::g_last_error; }

int nvk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int nvk_pool_create(int n_layers, int num_blocks, int block_size, int n_kv_heads, int head_dim, int device,
                    nvk_pool **out_pool)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!out_pool)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_create: out_pool is null");
    *out_pool = nullptr;
    if (n_layers <= 0 || num_blocks <= 0 || block_size <= 0 || n_kv_heads <= 0 || head_dim <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_create: every dimension must be positive");
    if (head_dim % 4 != 0 || head_dim > 256)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_create: head_dim=%d unsupported (multiple of 4, <= 256)",
                         head_dim);
    if ((long long)num_blocks * block_size > 0x7fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_create: num_blocks*block_size must fit in int32");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nvk::fail(NVK_ERR_NO_DEVICE, "nvk_pool_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_create: device %d outside [0,%d)", device, ndev);

    nvk::DeviceScope scope(device);
    if (!scope.ok)
        return nvk::fail(NVK_ERR_HIP, "nvk_pool_create: hipSetDevice(%d) failed", device);

    nvk_pool *pool = new (std::nothrow) nvk_pool();
    if (!pool)
        return nvk::fail(NVK_ERR_OOM, "nvk_pool_create: host allocation failed");
    pool->n_layers = n_layers, pool->num_blocks = num_blocks, pool->block_size = block_size;
    pool->n_kv_heads = n_kv_heads, pool->head_dim = head_dim, pool->device = device;
    pool->bs_magic     = division_magic(block_size);
    pool->layer_floats = (size_t)num_blocks * block_size * n_kv_heads * head_dim;

    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        pool->num_cus = prop.multiProcessorCount;

    const size_t bytes = pool->layer_floats * (size_t)n_layers * sizeof(float);
    // A layer-call streams the layer's K slab and V slab side by side.  Measured (profiles/r02_sweep_pool_layers_5b.log,
    // r02_hbm_alloc_probe.log): two streams 16 GiB or more apart run at 0.79-0.80 of the HBM peak, two streams a slab
    // apart at 0.87 -- so a layer's V slab sits right behind its K slab, [layer][K | V] in ONE allocation, instead of
    // the reference's two arrays of [layer][slab] (model.hpp:95,262-265), in which a 32-layer pool puts them half the
    // pool apart.  (Two allocations remain the fallback when one of twice the size cannot be had.)
    hipError_t e = hipErrorOutOfMemory;
#if NVK_POOL_LAYOUT == 1
    size_t skew = NVK_POOL_SKEW_BYTES / sizeof(float); // floats between the end of a K slab and its V slab (and the next K)
#ifdef NVK_POOL_SKEW_ENV // A/B build: the skew from the environment
    if (const char *env = getenv("NVK_POOL_SKEW"))
        skew = (size_t)atoll(env) / sizeof(float);
#endif
    e = hipMalloc(&pool->k, 2 * (bytes + (size_t)n_layers * skew * sizeof(float)));
    if (e == hipSuccess) {
        pool->one_allocation = true;
        pool->layer_stride   = 2 * (pool->layer_floats + skew);
        pool->v              = pool->k + pool->layer_floats + skew;
    }
    else {
        (void)hipGetLastError();
        pool->k = nullptr;
    }
#endif
    if (!pool->one_allocation) {
        pool->layer_stride = pool->layer_floats;
        e                  = hipMalloc(&pool->k, bytes);
        if (e == hipSuccess)
            e = hipMalloc(&pool->v, bytes);
    }
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&pool->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreate(&pool->ev_a);
    if (e == hipSuccess)
        e = hipEventCreate(&pool->ev_b);
    // zero like std::vector::resize in initialize_paged_attention (model.hpp:267-268)
    if (e == hipSuccess)
        e = hipMemsetAsync(pool->k, 0, pool->one_allocation ? pool->layer_stride * (size_t)n_layers * sizeof(float) : bytes,
                           pool->stream);
    if (e == hipSuccess && !pool->one_allocation)
        e = hipMemsetAsync(pool->v, 0, bytes, pool->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(pool->stream);
    if (e != hipSuccess) {
        const int code = nvk::fail(e == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP,
                                   "nvk_pool_create: %s (2 x %zu bytes requested)", hipGetErrorString(e), bytes);
        nvk_pool_destroy(pool);
        return code;
    }
    {
        // set-up for nvk_decode_device_noalloc: this device's address of the static scratch array, resolved here,
        // with the device current, so that entry point's own calls only look it up
        unsigned char *unused = nullptr;
        const int      rc     = nvk::noalloc_scratch_on(device, &unused);
        if (rc != NVK_OK) {
            nvk_pool_destroy(pool);
            return rc;
        }
    }
    if (head_dim % 2 == 0) {
        const int rc = ensure_rope_table(pool, 10000.0f); // config.rope_theta, include/core/model.hpp:40
        if (rc != NVK_OK) {
            nvk_pool_destroy(pool);
            return rc;
        }
    }
    *out_pool = pool;
    return NVK_OK;
}

int nvk_pool_set_rope_theta(nvk_pool *pool, float rope_theta)
{
    if (!pool || !(rope_theta > 0.f) || pool->head_dim % 2 != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_set_rope_theta: bad arguments");
    NVK_SCOPE(pool->device);
    return ensure_rope_table(pool, rope_theta);
}

int nvk_pool_destroy(nvk_pool *pool)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!pool)
        return NVK_OK;
    nvk::DeviceScope scope(pool->device);
    if (t_bound_pool == pool)
        t_bound_pool = nullptr;
    if (pool->stream)
        (void)hipStreamSynchronize(pool->stream);
    if (pool->k)
        (void)hipFree(pool->k);
    if (pool->v && !pool->one_allocation)
        (void)hipFree(pool->v);
    if (pool->ws_o)
        (void)hipFree(pool->ws_o);
    if (pool->ws_ml)
        (void)hipFree(pool->ws_ml);
    if (pool->arrivals)
        (void)hipFree(pool->arrivals);
    free_plan(pool->own_plan);
    if (pool->rope_inv_freq)
        (void)hipFree(pool->rope_inv_freq);
    if (pool->desc_host)
        (void)hipHostFree(pool->desc_host);
    if (pool->desc_dev)
        (void)hipFree(pool->desc_dev);
    if (pool->desc_ev)
        (void)hipEventDestroy(pool->desc_ev);
    if (pool->stage)
        (void)hipFree(pool->stage);
    if (pool->pin)
        (void)hipHostFree(pool->pin);
    if (pool->ev_a)
        (void)hipEventDestroy(pool->ev_a);
    if (pool->ev_b)
        (void)hipEventDestroy(pool->ev_b);
    if (pool->stream)
        (void)hipStreamDestroy(pool->stream);
    delete pool;
    return NVK_OK;
}

int nvk_pool_geometry(const nvk_pool *pool, int *out6)
{
    if (!pool || !out6)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_geometry: null argument");
    out6[0] = pool->n_layers, out6[1] = pool->num_blocks, out6[2] = pool->block_size;
    out6[3] = pool->n_kv_heads, out6[4] = pool->head_dim, out6[5] = pool->device;
    return NVK_OK;
}

int nvk_pool_layer_ptrs(const nvk_pool *pool, int layer, void **k_dev, void **v_dev)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_layer_ptrs");
    if (rc != NVK_OK)
        return rc;
    if (k_dev)
        *k_dev = pool->k_layer(layer);
    if (v_dev)
        *v_dev = pool->v_layer(layer);
    return NVK_OK;
}

int nvk_pool_stream(const nvk_pool *pool, void **stream)
{
    if (!pool || !stream)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_stream: null argument");
    *stream = (void *)pool->stream;
    return NVK_OK;
}

long long nvk_pool_workspace_generation(const nvk_pool *pool) { return pool ? pool->ws_generation : -1; }

int nvk_pool_synchronize(nvk_pool *pool)
{
    if (!pool)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_synchronize: null pool");
    NVK_SCOPE(pool->device);
    HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_pool_append(nvk_pool *pool, int layer, int physical_block, int block_offset, const float *k_row,
                    const float *v_row)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_append");
    if (rc != NVK_OK)
        return rc;
    if (!k_row || !v_row)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_append: null row");
    if (physical_block < 0 || physical_block >= pool->num_blocks)
        return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_pool_append: block %d outside [0,%d)", physical_block,
                         pool->num_blocks);
    if (block_offset < 0 || block_offset >= pool->block_size)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_append: offset %d outside [0,%d)", block_offset,
                         pool->block_size);
    NVK_SCOPE(pool->device);
    const size_t row = (size_t)pool->row_len();
    const size_t at  = ((size_t)physical_block * pool->block_size + block_offset) * row;
    HIP_TRY(hipMemcpyAsync(pool->k_layer(layer) + at, k_row, row * sizeof(float), hipMemcpyHostToDevice, pool->stream));
    HIP_TRY(hipMemcpyAsync(pool->v_layer(layer) + at, v_row, row * sizeof(float), hipMemcpyHostToDevice, pool->stream));
    // the caller may reuse k_row/v_row as soon as we return (the reference's memcpy is synchronous too)
    HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_pool_append_batch(nvk_pool *pool, int layer, int num_seqs, const float *k_rows, const float *v_rows,
                          const int *block_tables, int max_blocks, const int *positions, int ptr_kind)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_append_batch");
    if (rc != NVK_OK)
        return rc;
    if (num_seqs < 0 || max_blocks <= 0 || (num_seqs > 0 && (!k_rows || !v_rows || !block_tables || !positions)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_append_batch: bad arguments");
    if (num_seqs == 0)
        return NVK_OK;
    NVK_SCOPE(pool->device);
    const int    row = pool->row_len();
    const float *dk = k_rows, *dv = v_rows;
    const int   *dt = block_tables, *dp = positions;
    if (ptr_kind == NVK_PTR_HOST) {
        // validate where the reference would have undefined behaviour or throw
        for (int s = 0; s < num_seqs; s++) {
            const int pos = positions[s];
            if (pos < 0 || pos / pool->block_size >= max_blocks)
                return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_append_batch: position %d of sequence %d outside its table",
                                 pos, s);
            const int pb = block_tables[(size_t)s * max_blocks + pos / pool->block_size];
            if (pb < 0 || pb >= pool->num_blocks)
                return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_pool_append_batch: block %d of sequence %d outside [0,%d)", pb, s,
                                 pool->num_blocks);
        }
        const size_t rows_b = (size_t)num_seqs * row * sizeof(float);
        const size_t tab_b  = (size_t)num_seqs * max_blocks * sizeof(int);
        const size_t pos_b  = (size_t)num_seqs * sizeof(int);
        const size_t a1     = (rows_b + 255) / 256 * 256; // 256-B aligned sections
        const size_t a2     = (tab_b + 255) / 256 * 256;
        rc              = ensure_stage(pool, 2 * a1 + a2 + pos_b + 256);
        if (rc != NVK_OK)
            return rc;
        char *base = pool->stage;
        HIP_TRY(hipMemcpyAsync(base, k_rows, rows_b, hipMemcpyHostToDevice, pool->stream));
        HIP_TRY(hipMemcpyAsync(base + a1, v_rows, rows_b, hipMemcpyHostToDevice, pool->stream));
        HIP_TRY(hipMemcpyAsync(base + 2 * a1, block_tables, tab_b, hipMemcpyHostToDevice, pool->stream));
        HIP_TRY(hipMemcpyAsync(base + 2 * a1 + a2, positions, pos_b, hipMemcpyHostToDevice, pool->stream));
        dk = (const float *)base;
        dv = (const float *)(base + a1);
        dt = (const int *)(base + 2 * a1);
        dp = (const int *)(base + 2 * a1 + a2);
    }
    else if (ptr_kind != NVK_PTR_DEVICE) {
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_append_batch: ptr_kind %d", ptr_kind);
    }
    hipLaunchKernelGGL(nvk::nvk_append_kernel, dim3((unsigned)num_seqs), dim3(256), 0, pool->stream, pool->k_layer(layer),
                       pool->v_layer(layer), dk, dv, dt, max_blocks, dp, pool->num_blocks, pool->block_size,
                       pool->bs_magic, row);
    HIP_TRY(hipGetLastError());
    if (ptr_kind == NVK_PTR_HOST)
        HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_pool_rope_append_batch(nvk_pool *pool, int layer, int num_seqs, int n_heads, float *q, const float *k_rows,
                               const float *v_rows, const int *block_tables, int max_blocks, const int *positions,
                               float rope_theta, int ptr_kind, void *stream)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_rope_append_batch");
    if (rc != NVK_OK)
        return rc;
    const bool rotate = rope_theta > 0.f; // 0: plain append (q may be null and is not touched)
    if (num_seqs < 0 || max_blocks <= 0 || n_heads <= 0 || !(rope_theta >= 0.f)
        || (num_seqs > 0 && ((rotate && !q) || !k_rows || !v_rows || !block_tables || !positions)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_rope_append_batch: bad arguments");
    if (pool->head_dim % 2 != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_rope_append_batch: head_dim must be even");
    if (num_seqs == 0)
        return NVK_OK;
    NVK_SCOPE(pool->device);
    hipStream_t st   = (ptr_kind == NVK_PTR_DEVICE && stream) ? (hipStream_t)stream : pool->stream;
    const int   hd   = pool->head_dim;
    const int   half = hd / 2;

    if (rotate) {
        rc = ensure_rope_table(pool, rope_theta);
        if (rc != NVK_OK)
            return rc;
    }

    const int    row = pool->row_len();
    float       *dq  = q;
    const float *dk = k_rows, *dv = v_rows;
    const int   *dt = block_tables, *dp = positions;
    const size_t q_b = (size_t)num_seqs * n_heads * hd * sizeof(float);
    if (ptr_kind == NVK_PTR_HOST) {
        for (int s = 0; s < num_seqs; s++) {
            const int pos = positions[s];
            if (pos < 0 || pos / pool->block_size >= max_blocks)
                return nvk::fail(NVK_ERR_INVALID_ARG,
                                 "nvk_pool_rope_append_batch: position %d of sequence %d outside its table", pos, s);
            const int pb = block_tables[(size_t)s * max_blocks + pos / pool->block_size];
            if (pb < 0 || pb >= pool->num_blocks)
                return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_pool_rope_append_batch: block %d of sequence %d outside [0,%d)",
                                 pb, s, pool->num_blocks);
        }
        const size_t rows_b = (size_t)num_seqs * row * sizeof(float);
        const size_t tab_b  = (size_t)num_seqs * max_blocks * sizeof(int);
        const size_t pos_b  = (size_t)num_seqs * sizeof(int);
        auto         al     = [](size_t b) { return (b + 255) / 256 * 256; };
        rc = ensure_stage(pool, al(q_b) + 2 * al(rows_b) + al(tab_b) + al(pos_b));
        if (rc != NVK_OK)
            return rc;
        char *base = pool->stage;
        dq         = (float *)base;
        dk         = (const float *)(base + al(q_b));
        dv         = (const float *)(base + al(q_b) + al(rows_b));
        dt         = (const int *)(base + al(q_b) + 2 * al(rows_b));
        dp         = (const int *)(base + al(q_b) + 2 * al(rows_b) + al(tab_b));
        if (rotate)
            HIP_TRY(hipMemcpyAsync(dq, q, q_b, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)dk, k_rows, rows_b, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)dv, v_rows, rows_b, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)dt, block_tables, tab_b, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)dp, positions, pos_b, hipMemcpyHostToDevice, st));
    }
    else if (ptr_kind != NVK_PTR_DEVICE) {
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_rope_append_batch: ptr_kind %d", ptr_kind);
    }
    hipLaunchKernelGGL(nvk::nvk_rope_append_kernel, dim3((unsigned)num_seqs), dim3(256), (size_t)half * sizeof(float2), st,
                       pool->k_layer(layer), pool->v_layer(layer), dq, dk, dv, dt, (const int *)nullptr, max_blocks, dp,
                       rotate ? pool->rope_inv_freq : (const float *)nullptr,
                       pool->num_blocks, pool->block_size, pool->bs_magic, n_heads, pool->n_kv_heads, hd);
    HIP_TRY(hipGetLastError());
    if (ptr_kind == NVK_PTR_HOST) {
        if (rotate)
            HIP_TRY(hipMemcpyAsync(q, dq, q_b, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return NVK_OK;
}

} // extern "C"

int nvk::pool_renew_stream_if_stuck(nvk_pool *pool)
{
    if (!pool)
        return nvk::fail(NVK_ERR_INVALID_ARG, "pool_renew_stream_if_stuck: null pool");
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(pool->stream, &status) == hipSuccess && status == hipStreamCaptureStatusNone)
        return NVK_OK;
    (void)hipGetLastError();
    NVK_SCOPE(pool->device);
    hipStream_t fresh = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    // (what was enqueued before the recording began has to be over before anything follows it on the new stream; the old
    //  stream itself can no longer be waited for)
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipStreamDestroy(fresh);
        return nvk::fail(NVK_ERR_HIP, "pool_renew_stream_if_stuck: hipDeviceSynchronize failed: %s", hipGetErrorString(e));
    }
    (void)hipStreamDestroy(pool->stream);
    (void)hipGetLastError();
    pool->stream = fresh;
    return NVK_OK;
}


int nvk::pool_rope_append_target(nvk_pool *pool, int layer, float rope_theta, const int *block_tables_dev, int max_blocks,
                                 const int *positions_dev, RopeAppendTarget *out)
{
    int rc = check_pool_layer(pool, layer, "pool_rope_append_target");
    if (rc != NVK_OK)
        return rc;
    if (!(rope_theta > 0.f) || !block_tables_dev || !positions_dev || max_blocks <= 0 || !out || pool->head_dim % 2 != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "pool_rope_append_target: bad arguments");
    NVK_SCOPE(pool->device);
    rc = ensure_rope_table(pool, rope_theta);
    if (rc != NVK_OK)
        return rc;
    out->k_layer = pool->k_layer(layer), out->v_layer = pool->v_layer(layer);
    out->block_tables = block_tables_dev, out->positions = positions_dev, out->inv_freq = pool->rope_inv_freq;
    out->max_blocks = max_blocks, out->num_blocks = pool->num_blocks, out->block_size = pool->block_size;
    out->head_dim = pool->head_dim, out->row_len = pool->row_len(), out->bs_magic = pool->bs_magic;
    return NVK_OK;
}

extern "C" {

int nvk_pool_upload_block(nvk_pool *pool, int layer, int physical_block, const float *k_block, const float *v_block)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_upload_block");
    if (rc != NVK_OK)
        return rc;
    if (physical_block < 0 || physical_block >= pool->num_blocks)
        return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_pool_upload_block: block %d outside [0,%d)", physical_block,
                         pool->num_blocks);
    NVK_SCOPE(pool->device);
    const size_t blk = (size_t)pool->block_size * pool->row_len();
    if (k_block)
        HIP_TRY(hipMemcpyAsync(pool->k_layer(layer) + physical_block * blk, k_block, blk * sizeof(float),
                               hipMemcpyHostToDevice, pool->stream));
    if (v_block)
        HIP_TRY(hipMemcpyAsync(pool->v_layer(layer) + physical_block * blk, v_block, blk * sizeof(float),
                               hipMemcpyHostToDevice, pool->stream));
    HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_pool_download_block(nvk_pool *pool, int layer, int physical_block, float *k_block, float *v_block)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_download_block");
    if (rc != NVK_OK)
        return rc;
    if (physical_block < 0 || physical_block >= pool->num_blocks)
        return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_pool_download_block: block %d outside [0,%d)", physical_block,
                         pool->num_blocks);
    NVK_SCOPE(pool->device);
    const size_t blk = (size_t)pool->block_size * pool->row_len();
    if (k_block)
        HIP_TRY(hipMemcpyAsync(k_block, pool->k_layer(layer) + physical_block * blk, blk * sizeof(float),
                               hipMemcpyDeviceToHost, pool->stream));
    if (v_block)
        HIP_TRY(hipMemcpyAsync(v_block, pool->v_layer(layer) + physical_block * blk, blk * sizeof(float),
                               hipMemcpyDeviceToHost, pool->stream));
    HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_pool_gather(nvk_pool *pool, int layer, const int *block_table, int num_tokens, int kv_head, float *k_out,
                    float *v_out)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_gather");
    if (rc != NVK_OK)
        return rc;
    if (num_tokens < 0 || kv_head < 0 || kv_head >= pool->n_kv_heads || (num_tokens > 0 && (!block_table || !k_out || !v_out)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_gather: bad arguments");
    if (num_tokens == 0)
        return NVK_OK;
    const int nblk = (num_tokens + pool->block_size - 1) / pool->block_size;
    for (int i = 0; i < nblk; i++)
        if (block_table[i] < 0 || block_table[i] >= pool->num_blocks)
            return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_pool_gather: table[%d]=%d outside [0,%d)", i, block_table[i],
                             pool->num_blocks);
    NVK_SCOPE(pool->device);
    const size_t out_b = (size_t)num_tokens * pool->head_dim * sizeof(float);
    const size_t out_a = (out_b + 255) / 256 * 256;
    const size_t tab_b = (size_t)nblk * sizeof(int);
    rc                 = ensure_stage(pool, 2 * out_a + tab_b);
    if (rc != NVK_OK)
        return rc;
    float *dk = (float *)pool->stage;
    float *dv = (float *)(pool->stage + out_a);
    int   *dt = (int *)(pool->stage + 2 * out_a);
    HIP_TRY(hipMemcpyAsync(dt, block_table, tab_b, hipMemcpyHostToDevice, pool->stream));
    hipLaunchKernelGGL(nvk::nvk_gather_kernel, dim3((unsigned)num_tokens), dim3(64), 0, pool->stream, pool->k_layer(layer),
                       pool->v_layer(layer), dt, num_tokens, kv_head, pool->num_blocks, pool->block_size, pool->bs_magic,
                       pool->n_kv_heads, pool->head_dim, dk, dv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(k_out, dk, out_b, hipMemcpyDeviceToHost, pool->stream));
    HIP_TRY(hipMemcpyAsync(v_out, dv, out_b, hipMemcpyDeviceToHost, pool->stream));
    HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_pool_fill_uniform(nvk_pool *pool, int layer, uint64_t seed_k, uint64_t seed_v)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_fill_uniform");
    if (rc != NVK_OK)
        return rc;
    NVK_SCOPE(pool->device);
    const dim3  grid((unsigned)(pool->num_cus * 8)), block(256);
    hipLaunchKernelGGL(nvk::nvk_fill_kernel, grid, block, 0, pool->stream, pool->k_layer(layer), seed_k,
                       (uint64_t)pool->layer_floats);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(nvk::nvk_fill_kernel, grid, block, 0, pool->stream, pool->v_layer(layer), seed_v,
                       (uint64_t)pool->layer_floats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_pool_reserve_workspace(nvk_pool *pool, int num_seqs, int n_heads, int max_ctx, int chunk_tokens)
{
    if (!pool || num_seqs < 0 || max_ctx < 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_reserve_workspace: bad arguments");
    NVK_SCOPE(pool->device);
    LaunchShape shape;
    int rc = make_shape(pool, num_seqs, n_heads, max_ctx, chunk_tokens > 0 ? NVK_DECODE_SPLIT_K : NVK_DECODE_AUTO,
                        chunk_tokens, &shape);
    if (rc != NVK_OK)
        return rc;
    nvk_plan *plan = nullptr;
    rc             = own_plan_of(pool, &plan);
    if (rc != NVK_OK)
        return rc;
    rc = plan_reserve(plan, (size_t)num_seqs * shape.max_chunks, (size_t)num_seqs, nullptr);
    if (rc != NVK_OK)
        return rc;
    return ensure_workspace(pool, (size_t)num_seqs * shape.max_chunks * n_heads * pool->head_dim,
                            (size_t)num_seqs * shape.max_chunks * n_heads * 2, (size_t)num_seqs * shape.qgroups, nullptr);
}

#ifdef NVK_STAMPS
int nvk_debug_set_stamps(void *dev_buffer)
{
    g_debug_stamps = (unsigned long long *)dev_buffer;
    return NVK_OK;
}
#endif

int nvk_plan_create(nvk_pool *pool, nvk_plan **out_plan)
{
    if (!pool || !out_plan)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_plan_create: null argument");
    nvk_plan *plan = new (std::nothrow) nvk_plan();
    if (!plan)
        return nvk::fail(NVK_ERR_OOM, "nvk_plan_create: host allocation failed");
    plan->pool = pool;
    *out_plan  = plan;
    return NVK_OK;
}

int nvk_plan_destroy(nvk_plan *plan)
{
    nvk::GateHold hold; // (nvk_device_scope.h: not while a stream of the process records a graph)
    if (!plan)
        return NVK_OK;
    nvk::DeviceScope scope(plan->pool->device);
    (void)hipDeviceSynchronize();
    free_plan(plan);
    return NVK_OK;
}

int nvk_plan_build(nvk_plan *plan, int num_seqs, int n_heads, const int *ctx_lens_dev, int max_ctx, int flags,
                   int chunk_tokens, void *stream)
{
    if (!plan || num_seqs < 0 || max_ctx < 0 || (num_seqs > 0 && !ctx_lens_dev && !(flags & NVK_DECODE_UNIFORM)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_plan_build: bad arguments");
    if (!decode_flags_ok(flags))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_plan_build: flags 0x%x", flags);
    NVK_SCOPE(plan->pool->device);
    return build_plan(plan, num_seqs, n_heads, ctx_lens_dev, max_ctx, flags, chunk_tokens,
                      stream ? (hipStream_t)stream : plan->pool->stream);
}

} // extern "C"
long long nvk::plan_generation(const nvk_plan *plan) { return plan ? plan->generation : -1; }
extern "C" {

int nvk_plan_info(const nvk_plan *plan, int *out4)
{
    if (!plan || !out4 || !plan->built)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_plan_info: null or unbuilt plan");
    out4[0] = plan->shape.chunk_tokens, out4[1] = plan->shape.max_chunks, out4[2] = plan->item_bound;
    out4[3] = plan->shape.waves;
    return NVK_OK;
}

int nvk_decode_planned(nvk_pool *pool, int layer, const nvk_plan *plan, const float *q_dev, const int *block_tables_dev,
                       int max_blocks, float *out_dev, void *stream)
{
    int rc = check_pool_layer(pool, layer, "nvk_decode_planned");
    if (rc != NVK_OK)
        return rc;
    if (!plan || max_blocks <= 0 || (plan->num_seqs > 0 && (!q_dev || !block_tables_dev || !out_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_planned: bad arguments");
    NVK_SCOPE(pool->device);
    return enqueue_planned(pool, layer, plan, q_dev, block_tables_dev, max_blocks, out_dev,
                           stream ? (hipStream_t)stream : pool->stream, nullptr, nullptr);
}

int nvk_decode_device(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                      const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx, float *out_dev,
                      int flags, int chunk_tokens, void *stream)
{
    int rc = check_pool_layer(pool, layer, "nvk_decode_device");
    if (rc != NVK_OK)
        return rc;
    if (num_seqs < 0 || max_blocks <= 0 || max_ctx < 0
        || (num_seqs > 0 && (!q_dev || !block_tables_dev || (!ctx_lens_dev && !(flags & NVK_DECODE_UNIFORM)) || !out_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_device: bad arguments");
    if (!decode_flags_ok(flags))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_device: flags 0x%x", flags);
    NVK_SCOPE(pool->device);
    return enqueue_decode(pool, layer, num_seqs, n_heads, q_dev, block_tables_dev, max_blocks, ctx_lens_dev, max_ctx,
                          out_dev, flags, chunk_tokens, stream ? (hipStream_t)stream : pool->stream);
}

} // extern "C"

// ---- the entry point that allocates nothing ------------------------------------------------------
// nvk_decode_device for callers that cannot call hipMalloc (a stream under graph capture, a tight serving loop,
// a harness that forbids it).  Everything nvk_decode_device would keep in buffers it allocates or grows -- the work
// list, the split-K partials, the arrival counters -- lives in ONE array that is part of the code object, so it
// exists on every GPU the library runs on from load to exit.  `nvk_noalloc_scratch` below is the host-side
// stand-in for that array, NOT a device address: its address on a device comes from hipGetSymbolAddress with
// that device current (each GPU holds its own copy), is resolved once per device and is what every use below goes
// through.  Calls on one GPU share the array and its last contents: what must start at zero (the arrival
// counters) is cleared on the stream at each call, and two calls that overlap in time on one GPU race.
constexpr size_t kNoAllocScratchBytes = (size_t)24 << 20;
static __device__ __align__(256) unsigned char nvk_noalloc_scratch[kNoAllocScratchBytes];

namespace {
constexpr int  kMaxDevices = 64;
std::mutex     g_noalloc_mutex;
unsigned char *g_noalloc_ptr[kMaxDevices] = {}; // per device: the array's address there
} // namespace

// The calling thread's current device must be `device`.
int nvk::noalloc_scratch_on(int device, unsigned char **out)
{
    if (device < 0 || device >= kMaxDevices)
        return nvk::fail(NVK_ERR_INVALID_ARG, "device %d outside [0,%d)", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[device]) {
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(nvk_noalloc_scratch)));
        if (!ptr)
            return nvk::fail(NVK_ERR_HIP, "hipGetSymbolAddress gave a null address for the scratch array on device %d",
                             device);
        g_noalloc_ptr[device] = static_cast<unsigned char *>(ptr);
    }
    *out = g_noalloc_ptr[device];
    return NVK_OK;
}

extern "C" size_t nvk_noalloc_scratch_bytes(void) { return kNoAllocScratchBytes; }

extern "C" int nvk_decode_device_noalloc(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                                         const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev,
                                         int max_ctx, float *out_dev, int flags, int chunk_tokens, void *stream)
{
    int rc = check_pool_layer(pool, layer, "nvk_decode_device_noalloc");
    if (rc != NVK_OK)
        return rc;
    if (num_seqs < 0 || max_blocks <= 0 || max_ctx < 0
        || (num_seqs > 0 && (!q_dev || !block_tables_dev || (!ctx_lens_dev && !(flags & NVK_DECODE_UNIFORM)) || !out_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_device_noalloc: bad arguments");
    if (!decode_flags_ok(flags))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_device_noalloc: flags 0x%x", flags);
    if (num_seqs == 0)
        return NVK_OK;
    nvk::DeviceScope scope(pool->device);
    if (!scope.ok)
        return nvk::fail(NVK_ERR_HIP, "nvk_decode_device_noalloc: hipSetDevice(%d) failed", pool->device);
    unsigned char *base = nullptr;
    rc                  = nvk::noalloc_scratch_on(pool->device, &base);
    if (rc != NVK_OK)
        return rc;
    hipStream_t st = stream ? (hipStream_t)stream : pool->stream;

    nvk_plan plan; // lives for this call; its device arrays are slices of the scratch array
    plan.pool = pool;
    rc        = plan_prepare(&plan, num_seqs, n_heads, ctx_lens_dev, max_ctx, flags, chunk_tokens);
    if (rc != NVK_OK)
        return rc;
    const LaunchShape &sh = plan.shape;
    // slices, each a multiple of 256 B: arrival counters | plan counters | plan items | (max, sum) | partial outputs
    auto         al       = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t rows     = (size_t)num_seqs * sh.max_chunks;
    const size_t arr_b    = sh.fuse_merge ? al((size_t)num_seqs * sh.qgroups * sizeof(int)) : 0;
    const size_t cnt_b    = plan.direct ? 0 : al((size_t)(num_seqs + 1) * sizeof(int));
    const size_t items_b  = plan.direct ? 0 : al(rows * sizeof(nvk::WorkItem));
    const size_t ml_b     = sh.max_chunks > 1 ? al(rows * n_heads * 2 * sizeof(float)) : 0;
    const size_t o_b      = sh.max_chunks > 1 ? al(rows * n_heads * pool->head_dim * sizeof(float)) : 0;
    const size_t need     = arr_b + cnt_b + items_b + ml_b + o_b;
    if (need > kNoAllocScratchBytes)
        return nvk::fail(NVK_ERR_INVALID_ARG,
                         "nvk_decode_device_noalloc: this batch needs %zu bytes of scratch (%d rows x %d chunks of %d "
                         "tokens), the static array holds %zu: use fewer or longer chunks, or nvk_decode_device",
                         need, num_seqs, sh.max_chunks, sh.chunk_tokens, kNoAllocScratchBytes);
    Workspace ws;
    ws.arrivals     = reinterpret_cast<int *>(base);
    plan.counters   = reinterpret_cast<int *>(base + arr_b);
    plan.items      = reinterpret_cast<nvk::WorkItem *>(base + arr_b + cnt_b);
    ws.ws_ml        = reinterpret_cast<float *>(base + arr_b + cnt_b + items_b);
    ws.ws_o         = reinterpret_cast<float *>(base + arr_b + cnt_b + items_b + ml_b);
    ws.arrival_ints = arr_b / sizeof(int), ws.ml_floats = ml_b / sizeof(float), ws.o_floats = o_b / sizeof(float);
    if (arr_b)
        HIP_TRY(hipMemsetAsync(ws.arrivals, 0, arr_b, st)); // whatever the array's last user left there
    rc = plan_launch(&plan, ctx_lens_dev, st);
    if (rc == NVK_OK)
        rc = enqueue_planned(pool, layer, &plan, q_dev, block_tables_dev, max_blocks, out_dev, st, nullptr, nullptr, nullptr,
                             &ws);
    plan.items = nullptr, plan.counters = nullptr; // not this plan's to free
    return rc;
}

extern "C" {

int nvk_decode(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q, const int *block_tables,
               int max_blocks, const int *ctx_lens, float *out, int flags, int chunk_tokens)
{
    int rc = check_pool_layer(pool, layer, "nvk_decode");
    if (rc != NVK_OK)
        return rc;
    if (num_seqs < 0 || max_blocks <= 0 || (num_seqs > 0 && (!q || !block_tables || !ctx_lens || !out)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode: bad arguments");
    if (!decode_flags_ok(flags))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode: flags 0x%x", flags);
    if (num_seqs == 0)
        return NVK_OK;

    // Host-side validation of everything the kernel will dereference through.
    int max_ctx = 0;
    for (int s = 0; s < num_seqs; s++) {
        const int ctx = ctx_lens[s];
        if (ctx < 0)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode: ctx_lens[%d]=%d is negative", s, ctx);
        const int nblk = (ctx + pool->block_size - 1) / pool->block_size;
        if (nblk > max_blocks)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode: sequence %d needs %d blocks, table rows hold %d", s, nblk,
                             max_blocks);
        for (int i = 0; i < nblk; i++) {
            const int pb = block_tables[(size_t)s * max_blocks + i];
            if (pb < 0 || pb >= pool->num_blocks)
                return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_decode: block_tables[%d][%d]=%d outside [0,%d)", s, i, pb,
                                 pool->num_blocks);
        }
        max_ctx = std::max(max_ctx, ctx);
    }
    // the plan modifiers speak about lengths the library cannot see; here it can: decide for the caller
    flags &= ~(NVK_DECODE_UNIFORM | NVK_DECODE_NO_PLAN);
    if (std::all_of(ctx_lens, ctx_lens + num_seqs, [&](int c) { return c == max_ctx; }))
        flags |= NVK_DECODE_UNIFORM; // every row holds max_ctx tokens: the plan's order is the direct order

    NVK_SCOPE(pool->device);
    const size_t q_b   = (size_t)num_seqs * n_heads * pool->head_dim * sizeof(float);
    const size_t q_a   = (q_b + 255) / 256 * 256;
    const size_t tab_b = (size_t)num_seqs * max_blocks * sizeof(int);
    const size_t tab_a = (tab_b + 255) / 256 * 256;
    const size_t ctx_b = (size_t)num_seqs * sizeof(int);
    const size_t ctx_a = (ctx_b + 255) / 256 * 256;
    // device staging: q | tables | lengths | plan | out.  Small batches go through the pinned twin in one copy each
    // way, and their plan (the lengths are host data here) is built on the host and travels in the same copy instead
    // of costing a kernel: a single-sequence call is bound by latency, not by bandwidth.
    LaunchShape sh;
    rc = make_shape(pool, num_seqs, n_heads, max_ctx, flags, chunk_tokens, &sh);
    if (rc != NVK_OK)
        return rc;
    const long long bound   = (long long)num_seqs * sh.max_chunks;
    const size_t    cnt_a   = ((size_t)(num_seqs + 1) * sizeof(int) + 255) / 256 * 256;
    const size_t    items_a = ((size_t)bound * sizeof(nvk::WorkItem) + 255) / 256 * 256;
    const bool      pinned  = bound <= 0x7fffffffLL && q_a + tab_a + ctx_a + cnt_a + items_a + q_a <= ((size_t)4 << 20);
    if (!pinned) { // beyond a few MB the extra host memcpy is not worth it: separate copies, plan kernel
        rc = ensure_stage(pool, 2 * q_a + tab_a + ctx_a);
        if (rc != NVK_OK)
            return rc;
        float *dq   = (float *)pool->stage;
        float *dout = (float *)(pool->stage + q_a);
        int   *dtab = (int *)(pool->stage + 2 * q_a);
        int   *dctx = (int *)(pool->stage + 2 * q_a + tab_a);
        HIP_TRY(hipMemcpyAsync(dq, q, q_b, hipMemcpyHostToDevice, pool->stream));
        HIP_TRY(hipMemcpyAsync(dtab, block_tables, tab_b, hipMemcpyHostToDevice, pool->stream));
        HIP_TRY(hipMemcpyAsync(dctx, ctx_lens, ctx_b, hipMemcpyHostToDevice, pool->stream));
        rc = enqueue_decode(pool, layer, num_seqs, n_heads, dq, dtab, max_blocks, dctx, max_ctx, dout, flags,
                            chunk_tokens, pool->stream);
        if (rc != NVK_OK)
            return rc;
        HIP_TRY(hipMemcpyAsync(out, dout, q_b, hipMemcpyDeviceToHost, pool->stream));
        HIP_TRY(hipStreamSynchronize(pool->stream));
        return NVK_OK;
    }
    const size_t off_tab = q_a, off_ctx = off_tab + tab_a, off_cnt = off_ctx + ctx_a, off_items = off_cnt + cnt_a;
    const size_t in_bytes = off_items + items_a;
    rc = ensure_stage(pool, in_bytes + q_a);
    if (rc == NVK_OK)
        rc = ensure_pin(pool, in_bytes + q_a);
    if (rc == NVK_OK && sh.max_chunks > 1)
        rc = ensure_workspace(pool, (size_t)bound * n_heads * pool->head_dim, (size_t)bound * n_heads * 2,
                              sh.fuse_merge ? (size_t)num_seqs * sh.qgroups : 0, pool->stream);
    if (rc != NVK_OK)
        return rc;
    std::memcpy(pool->pin, q, q_b);
    std::memcpy(pool->pin + off_tab, block_tables, tab_b);
    std::memcpy(pool->pin + off_ctx, ctx_lens, ctx_b);
    {
        // the plan nvk_plan_kernel would build: full chunks sequence-major, then the ragged last chunks by decreasing
        // length (empty sequences last); the order only shapes the grid's tail, never a result
        int           *cnt   = reinterpret_cast<int *>(pool->pin + off_cnt);
        nvk::WorkItem *items = reinterpret_cast<nvk::WorkItem *>(pool->pin + off_items);
        const int      ct    = sh.chunk_tokens;
        int            n     = 0;
        for (int s2 = 0; s2 < num_seqs; s2++) {
            const int ctx = std::min(ctx_lens[s2], sh.max_chunks * ct), nfull = ctx / ct, rem = ctx - nfull * ct;
            const int nch = nfull + (rem > 0 ? 1 : 0);
            cnt[1 + s2]   = nch;
            for (int c = 0; c < nfull; c++, n++) {
                const int2 t = nvk::skewed_chunk(n, c, nfull, ct, sh.chunk_skew);
                items[n]     = nvk::WorkItem{s2, c | (nch << 16), t.x, t.y};
            }
        }
        const int first_partial = n;
        for (int s2 = 0; s2 < num_seqs; s2++) {
            const int ctx = std::min(ctx_lens[s2], sh.max_chunks * ct), nfull = ctx / ct, rem = ctx - nfull * ct;
            if (rem > 0 || ctx == 0)
                items[n++] = nvk::WorkItem{s2, nfull | ((nfull + (rem > 0 ? 1 : 0)) << 16), nfull * ct, ctx};
        }
        std::stable_sort(items + first_partial, items + n, [](const nvk::WorkItem &x, const nvk::WorkItem &y) {
            return x.t1 - x.t0 > y.t1 - y.t0;
        });
        cnt[0] = n;
    }
    HIP_TRY(hipMemcpyAsync(pool->stage, pool->pin, in_bytes, hipMemcpyHostToDevice, pool->stream));
    nvk_plan host_plan; // lives for this call only; its device arrays are slices of the staging buffer
    host_plan.pool = pool, host_plan.num_seqs = num_seqs, host_plan.n_heads = n_heads, host_plan.max_ctx = max_ctx;
    host_plan.shape = sh, host_plan.built = true, host_plan.item_bound = (int)bound;
    host_plan.items    = reinterpret_cast<nvk::WorkItem *>(pool->stage + off_items);
    host_plan.counters = reinterpret_cast<int *>(pool->stage + off_cnt);
    float *dout = (float *)(pool->stage + in_bytes);
    rc = enqueue_planned(pool, layer, &host_plan, (const float *)pool->stage, (const int *)(pool->stage + off_tab), max_blocks,
                         dout, pool->stream, nullptr, nullptr);
    if (rc != NVK_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(pool->pin + in_bytes, dout, q_b, hipMemcpyDeviceToHost, pool->stream));
    HIP_TRY(hipStreamSynchronize(pool->stream));
    std::memcpy(out, pool->pin + in_bytes, q_b);
    return NVK_OK;
}

int nvk_decode_bench(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                     const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx, float *out_dev,
                     int flags, int chunk_tokens, int replan_every, int warmup, int iters, float *total_ms,
                     float *attn_ms)
{
    return nvk_decode_bench_layers(pool, layer, 1, num_seqs, n_heads, q_dev, block_tables_dev, max_blocks, ctx_lens_dev,
                                   max_ctx, out_dev, flags, chunk_tokens, replan_every, warmup, iters, total_ms, attn_ms);
}

int nvk_decode_kernel_name(nvk_pool *pool, int num_seqs, int n_heads, int max_ctx, int flags, int chunk_tokens, char *buf,
                           int cap)
{
    if (!pool || !buf || cap <= 0 || !decode_flags_ok(flags))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_kernel_name: bad arguments");
    LaunchShape sh;
    int         rc = make_shape(pool, num_seqs, n_heads, max_ctx, flags, chunk_tokens, &sh);
    if (rc != NVK_OK)
        return rc;
    if (sh.mfma)
        snprintf(buf, (size_t)cap, "nvk::nvk_decode_mfma_kernel<%d>", 64 * sh.vpl);
    else if (sh.row_pairs)
        snprintf(buf, (size_t)cap, "nvk::nvk_decode_rowpairs_kernel<%d>", sh.qpk);
    else if (sh.kpw == 2)
        snprintf(buf, (size_t)cap, "nvk::nvk_decode_pairs_kernel<%d>", sh.qpk);
    else if (sh.wide)
        snprintf(buf, (size_t)cap, "nvk::nvk_decode_wide_kernel<%d>", sh.qpk);
    else
        snprintf(buf, (size_t)cap, "nvk::nvk_decode_%skernel<%d, %d, %s>", free_running(sh) ? "free_" : "", sh.vpl, sh.qpk,
                 sh.full ? "true" : "false");
    return NVK_OK;
}

int nvk_decode_bench_layers(nvk_pool *pool, int first_layer, int n_rotate, int num_seqs, int n_heads, const float *q_dev,
                            const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                            float *out_dev, int flags, int chunk_tokens, int replan_every, int warmup, int iters,
                            float *total_ms, float *attn_ms)
{
    int rc = check_pool_layer(pool, first_layer, "nvk_decode_bench");
    if (rc != NVK_OK)
        return rc;
    if (n_rotate < 1 || first_layer + n_rotate > pool->n_layers)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_bench: layers [%d, %d) outside the pool's %d", first_layer,
                         first_layer + n_rotate, pool->n_layers);
    if (iters < 0 || warmup < 0 || num_seqs <= 0 || iters > 100000 || replan_every < 1)
        return nvk::fail(NVK_ERR_INVALID_ARG,
                         "nvk_decode_bench: need 0 <= iters <= 100000, warmup >= 0, num_seqs > 0, replan_every >= 1");
    NVK_SCOPE(pool->device);
    hipStream_t st   = pool->stream;
    nvk_plan   *plan = nullptr;
    rc               = own_plan_of(pool, &plan);
    if (rc != NVK_OK)
        return rc;
    // layer-call i rebuilds the plan when i % replan_every == 0 (1: every call, like nvk_decode_device;
    // L: once per L calls, like the L layers of one model step sharing their batch's plan)
    auto one_call = [&](int i, hipEvent_t a, hipEvent_t b) -> int {
        if (i % replan_every == 0) {
            int r = build_plan(plan, num_seqs, n_heads, ctx_lens_dev, max_ctx, flags, chunk_tokens, st);
            if (r != NVK_OK)
                return r;
        }
        // call i reads layer first_layer + i % n_rotate: with n_rotate slabs larger together than the 256 MiB
        // Infinity Cache no call finds any of its bytes there
        return enqueue_planned(pool, first_layer + i % n_rotate, plan, q_dev, block_tables_dev, max_blocks, out_dev, st, a,
                               b);
    };
    for (int i = 0; i < warmup; i++) {
        rc = one_call(i, nullptr, nullptr);
        if (rc != NVK_OK)
            return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (total_ms)
        *total_ms = 0.f;
    if (attn_ms)
        *attn_ms = 0.f;
    if (iters == 0)
        return NVK_OK;

    // One timed region: ev_a .. ev_b brackets all `iters` layer-calls (plan rebuilds included).
    // Inside it, when attn_ms is wanted, every kAttnSampleEvery-th attention-kernel launch has its
    // own event pair (an event pair costs the stream about 6 us, so not every launch carries one).
    // (every launch when the region is short: the driver's default run is 20 layer-calls)
    const int               kAttnSampleEvery = iters <= 32 ? 1 : 8;
    const int               n_samples = attn_ms ? (iters + kAttnSampleEvery - 1) / kAttnSampleEvery : 0;
    std::vector<hipEvent_t> ev((size_t)2 * n_samples, nullptr);
    hipError_t              e = hipSuccess;
    for (auto &x : ev)
        if (e == hipSuccess)
            e = hipEventCreate(&x);
    if (e == hipSuccess)
        e = hipEventRecord(pool->ev_a, st);
    for (int i = 0; i < iters && e == hipSuccess && rc == NVK_OK; i++) {
        const bool sampled = attn_ms && i % kAttnSampleEvery == 0;
        rc = one_call(i, sampled ? ev[2 * (i / kAttnSampleEvery)] : nullptr,
                      sampled ? ev[2 * (i / kAttnSampleEvery) + 1] : nullptr);
    }
    if (e == hipSuccess)
        e = hipEventRecord(pool->ev_b, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    float whole = 0.f;
    if (e == hipSuccess && rc == NVK_OK)
        e = hipEventElapsedTime(&whole, pool->ev_a, pool->ev_b);
    double kernels = 0.0;
    if (attn_ms && e == hipSuccess && rc == NVK_OK)
        for (int i = 0; i < n_samples && e == hipSuccess; i++) {
            float one = 0.f;
            e         = hipEventElapsedTime(&one, ev[2 * i], ev[2 * i + 1]);
            kernels += one;
        }
    for (auto &x : ev)
        if (x)
            (void)hipEventDestroy(x);
    if (rc != NVK_OK)
        return rc;
    if (e != hipSuccess)
        return nvk::fail(NVK_ERR_HIP, "nvk_decode_bench: %s", hipGetErrorString(e));
    if (total_ms)
        *total_ms = whole / iters;
    if (attn_ms)
        *attn_ms = (float)(kernels / n_samples);
    return NVK_OK;
}

// A model step's worth of layer-calls as ONE hipGraph: the plan is built once (outside the graph, as nvk_plan_build
// per model step), `layers_per_graph` layer-calls are recorded from the pool's stream into a graph, and the graph is
// replayed.  What the replay removes is the host's share of every launch; the device-side hand-over from one
// layer-call's kernel to the next stays.  total_ms = region / (iters x layers_per_graph).
int nvk_decode_bench_graph(nvk_pool *pool, int first_layer, int n_rotate, int num_seqs, int n_heads, const float *q_dev,
                           const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                           float *out_dev, int flags, int chunk_tokens, int layers_per_graph, int warmup, int iters,
                           float *total_ms)
{
    int rc = check_pool_layer(pool, first_layer, "nvk_decode_bench_graph");
    if (rc != NVK_OK)
        return rc;
    if (n_rotate < 1 || first_layer + n_rotate > pool->n_layers)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_bench_graph: layers [%d, %d) outside the pool's %d", first_layer,
                         first_layer + n_rotate, pool->n_layers);
    if (iters < 1 || warmup < 0 || num_seqs <= 0 || iters > 100000 || layers_per_graph < 1 || layers_per_graph > 1024
        || !total_ms)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_decode_bench_graph: bad arguments");
    NVK_SCOPE(pool->device);
    hipStream_t st   = pool->stream;
    nvk_plan   *plan = nullptr;
    rc               = own_plan_of(pool, &plan);
    if (rc != NVK_OK)
        return rc;
    // the plan and one direct layer-call first: every buffer is sized before the capture starts
    rc = build_plan(plan, num_seqs, n_heads, ctx_lens_dev, max_ctx, flags, chunk_tokens, st);
    if (rc == NVK_OK)
        rc = enqueue_planned(pool, first_layer, plan, q_dev, block_tables_dev, max_blocks, out_dev, st, nullptr, nullptr);
    if (rc != NVK_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(st));
    hipGraph_t     graph = nullptr;
    hipGraphExec_t exec  = nullptr;
    hipError_t     e     = hipSuccess;
    {
        nvk::RecordAlone alone; // (nvk_device_scope.h: no other call of the library while a stream of the process records)
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < layers_per_graph && rc == NVK_OK; i++)
            rc = enqueue_planned(pool, first_layer + i % n_rotate, plan, q_dev, block_tables_dev, max_blocks, out_dev, st,
                                 nullptr, nullptr);
        e = hipStreamEndCapture(st, &graph);
    }
    if (rc != NVK_OK) {
        if (graph)
            (void)hipGraphDestroy(graph);
        return rc;
    }
    if (e == hipSuccess)
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    for (int i = 0; i < warmup && e == hipSuccess; i++)
        e = hipGraphLaunch(exec, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e == hipSuccess)
        e = hipEventRecord(pool->ev_a, st);
    for (int i = 0; i < iters && e == hipSuccess; i++)
        e = hipGraphLaunch(exec, st);
    if (e == hipSuccess)
        e = hipEventRecord(pool->ev_b, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    float whole = 0.f;
    if (e == hipSuccess)
        e = hipEventElapsedTime(&whole, pool->ev_a, pool->ev_b);
    if (exec)
        (void)hipGraphExecDestroy(exec);
    if (graph)
        (void)hipGraphDestroy(graph);
    if (e != hipSuccess)
        return nvk::fail(NVK_ERR_HIP, "nvk_decode_bench_graph: %s", hipGetErrorString(e));
    *total_ms = whole / ((float)iters * (float)layers_per_graph);
    return NVK_OK;
}

int nvk_pool_rope_append_chunks(nvk_pool *pool, int layer, int num_seqs, int n_heads, float *q, const float *k_rows,
                                const float *v_rows, const int *block_tables, int max_blocks, const int *chunk_starts,
                                const int *chunk_lens, float rope_theta, int ptr_kind, void *stream)
{
    int rc = check_pool_layer(pool, layer, "nvk_pool_rope_append_chunks");
    if (rc != NVK_OK)
        return rc;
    const bool rotate = rope_theta > 0.f;
    if (num_seqs < 0 || max_blocks <= 0 || n_heads <= 0 || !(rope_theta >= 0.f)
        || (num_seqs > 0 && (!chunk_starts || !chunk_lens || !block_tables || !k_rows || !v_rows || (rotate && !q))))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_rope_append_chunks: bad arguments");
    if (ptr_kind != NVK_PTR_HOST && ptr_kind != NVK_PTR_DEVICE)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_pool_rope_append_chunks: ptr_kind %d", ptr_kind);
    if (num_seqs == 0)
        return NVK_OK;
    NVK_SCOPE(pool->device);
    hipStream_t st = (ptr_kind == NVK_PTR_DEVICE && stream) ? (hipStream_t)stream : pool->stream;
    if (ptr_kind == NVK_PTR_HOST) {
        rc = validate_chunk_tables(pool, num_seqs, block_tables, max_blocks, chunk_starts, chunk_lens,
                                   "nvk_pool_rope_append_chunks");
        if (rc != NVK_OK)
            return rc;
    }
    ChunkRows rows;
    rc = stage_chunk_rows(pool, num_seqs, chunk_starts, chunk_lens, max_blocks, "nvk_pool_rope_append_chunks", st, &rows);
    if (rc != NVK_OK || rows.total_q == 0)
        return rc;
    const int hd = pool->head_dim, half = hd / 2, row = pool->row_len();
    if (rotate) {
        rc = ensure_rope_table(pool, rope_theta);
        if (rc != NVK_OK)
            return rc;
    }
    float       *dq = q;
    const float *dk = k_rows, *dv = v_rows;
    const int   *dt = block_tables;
    const size_t q_b = (size_t)rows.total_q * n_heads * hd * sizeof(float);
    if (ptr_kind == NVK_PTR_HOST) {
        const size_t rows_b = (size_t)rows.total_q * row * sizeof(float);
        const size_t tab_b  = (size_t)num_seqs * max_blocks * sizeof(int);
        auto         al     = [](size_t b) { return (b + 255) / 256 * 256; };
        rc = ensure_stage(pool, al(q_b) + 2 * al(rows_b) + al(tab_b));
        if (rc != NVK_OK)
            return rc;
        char *base = pool->stage;
        dq         = (float *)base;
        dk         = (const float *)(base + al(q_b));
        dv         = (const float *)(base + al(q_b) + al(rows_b));
        dt         = (const int *)(base + al(q_b) + 2 * al(rows_b));
        if (rotate)
            HIP_TRY(hipMemcpyAsync(dq, q, q_b, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)dk, k_rows, rows_b, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)dv, v_rows, rows_b, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)dt, block_tables, tab_b, hipMemcpyHostToDevice, st));
    }
    // one workgroup per new token: batch_rope (batch_ops.hpp:25-56) and the append loop (model_chunked.hpp:82-94)
    hipLaunchKernelGGL(nvk::nvk_rope_append_kernel, dim3((unsigned)rows.total_q), dim3(256), (size_t)half * sizeof(float2),
                       st, pool->k_layer(layer), pool->v_layer(layer), dq, dk, dv, dt, rows.row_seq, max_blocks,
                       rows.row_pos, rotate ? pool->rope_inv_freq : (const float *)nullptr, pool->num_blocks,
                       pool->block_size, pool->bs_magic, n_heads, pool->n_kv_heads, hd);
    HIP_TRY(hipGetLastError());
    if (ptr_kind == NVK_PTR_HOST) {
        if (rotate)
            HIP_TRY(hipMemcpyAsync(q, dq, q_b, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return NVK_OK;
}

int nvk_prefill_device(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                       const int *block_tables_dev, int max_blocks, const int *chunk_starts, const int *chunk_lens,
                       float *out_dev, int flags, void *stream)
{
    int rc = check_pool_layer(pool, layer, "nvk_prefill_device");
    if (rc != NVK_OK)
        return rc;
    if (num_seqs < 0 || max_blocks <= 0 || n_heads <= 0
        || (num_seqs > 0 && (!chunk_starts || !chunk_lens || !block_tables_dev || !q_dev || !out_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_prefill_device: bad arguments");
    if (flags != NVK_PREFILL_AUTO && flags != NVK_PREFILL_ROWWISE && flags != NVK_PREFILL_TILED)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_prefill_device: flags %d", flags);
    if (num_seqs == 0)
        return NVK_OK;
    NVK_SCOPE(pool->device);
    hipStream_t st = stream ? (hipStream_t)stream : pool->stream;
    if (n_heads % pool->n_kv_heads != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "n_heads=%d is not a positive multiple of n_kv_heads=%d", n_heads,
                         pool->n_kv_heads);

    // TILED (fp32 MFMA): K/V read once per tile of queries.  Needs head_dim <= 128 and contexts whose table
    // slice fits a workgroup's LDS; AUTO takes it when some chunk is long enough to fill tiles.
    int max_chunk = 0, top = 0;
    for (int s = 0; s < num_seqs; s++) {
        max_chunk = std::max(max_chunk, chunk_lens[s]);
        if (chunk_lens[s] > 0)
            top = std::max(top, chunk_starts[s] + chunk_lens[s]);
    }
    const bool tiled_ok = pool->head_dim <= 128 && pool->head_dim % 4 == 0
                          && (top + pool->block_size - 1) / pool->block_size <= kPrefillMaxTableEntries;
    if (flags == NVK_PREFILL_TILED && !tiled_ok)
        return nvk::fail(NVK_ERR_INVALID_ARG,
                         "nvk_prefill_device: NVK_PREFILL_TILED needs head_dim <= 128 and at most %d blocks per sequence",
                         kPrefillMaxTableEntries);
    if (flags == NVK_PREFILL_TILED || (flags == NVK_PREFILL_AUTO && tiled_ok && max_chunk >= 8)) {
        const int kv_mul = n_heads / pool->n_kv_heads;
        const int hpw    = kv_mul % 4 == 0 ? 4 : (kv_mul % 2 == 0 ? 2 : 1);
        const int    grid_y = pool->n_kv_heads * (kv_mul / hpw);
        PrefillWork  work;
        rc = stage_prefill_items(pool, num_seqs, chunk_starts, chunk_lens, max_blocks, 4 / hpw, grid_y, st, &work);
        if (rc != NVK_OK || work.n_items == 0)
            return rc;
        if (work.max_parts > 1) {
            rc = ensure_workspace(pool, (size_t)work.total_q * work.max_parts * n_heads * pool->head_dim,
                                  (size_t)work.total_q * work.max_parts * n_heads * 2, 0, st);
            if (rc != NVK_OK)
                return rc;
        }
        nvk::PrefillParams pp;
        pp.k_layer = pool->k_layer(layer), pp.v_layer = pool->v_layer(layer), pp.q = q_dev;
        pp.block_tables = block_tables_dev, pp.items = work.items_dev, pp.out = out_dev;
        pp.ws_o = pool->ws_o, pp.ws_ml = pool->ws_ml, pp.max_parts = work.max_parts;
        pp.max_blocks = max_blocks, pp.num_blocks = pool->num_blocks, pp.block_size = pool->block_size;
        pp.bs_magic = pool->bs_magic, pp.n_kv_heads = pool->n_kv_heads, pp.n_heads = n_heads, pp.kv_mul = kv_mul;
        pp.head_dim = pool->head_dim, pp.hpw = hpw;
        pp.q_scale  = 1.44269504088896340736f / sqrtf((float)pool->head_dim);
        const int    hdp  = pool->head_dim <= 64 ? 64 : 128;
        const int    nblk = (work.max_ctx + pool->block_size - 1) / pool->block_size;
        const size_t lds  = ((size_t)32 * (hdp + 4) + (size_t)32 * (hdp + 8)) * sizeof(float) + (size_t)(nblk + 1) * sizeof(int);
        if ((long long)work.n_items * grid_y > 0x7fffffffLL)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_prefill_device: %d tiles x %d head groups exceed the grid",
                             work.n_items, grid_y);
        const dim3   grid((unsigned)(work.n_items * grid_y));
        if (hdp == 64)
            hipLaunchKernelGGL((nvk::nvk_prefill_kernel<64>), grid, dim3(256), lds, st, pp);
        else
            hipLaunchKernelGGL((nvk::nvk_prefill_kernel<128>), grid, dim3(256), lds, st, pp);
        HIP_TRY(hipGetLastError());
        if (work.max_parts > 1) { // combine the key-range partials per query row (rows with one part are done)
            const int  vecs = n_heads * (pool->head_dim / 4);
            const dim3 mgrid((unsigned)work.total_q, (unsigned)((vecs + 255) / 256));
            hipLaunchKernelGGL(nvk::nvk_merge_kernel, mgrid, dim3(256), 0, st, pool->ws_o, pool->ws_ml, work.row_parts_dev,
                               (const int *)nullptr, 0, 1, out_dev, n_heads, pool->head_dim, work.max_parts);
            HIP_TRY(hipGetLastError());
        }
        return NVK_OK;
    }

    ChunkRows rows;
    rc = stage_chunk_rows(pool, num_seqs, chunk_starts, chunk_lens, max_blocks, "nvk_prefill_device", st, &rows);
    if (rc != NVK_OK || rows.total_q == 0)
        return rc;
    // ROWWISE: every query token is a decode row of its own (ctx = pos + 1) that reads its sequence's table
    // row; K/V are streamed once per query token, so this is the HBM-bound form, right for short chunks.
    nvk_plan *plan = nullptr;
    rc             = own_plan_of(pool, &plan);
    if (rc != NVK_OK)
        return rc;
    rc = build_plan(plan, rows.total_q, n_heads, rows.row_ctx, rows.max_ctx, NVK_DECODE_AUTO, 0, st);
    if (rc != NVK_OK)
        return rc;
    return enqueue_planned(pool, layer, plan, q_dev, block_tables_dev, max_blocks, out_dev, st, nullptr, nullptr,
                           rows.row_seq);
}

int nvk_prefill(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q, const int *block_tables,
                int max_blocks, const int *chunk_starts, const int *chunk_lens, float *out, int flags)
{
    int rc = check_pool_layer(pool, layer, "nvk_prefill");
    if (rc != NVK_OK)
        return rc;
    if (num_seqs < 0 || max_blocks <= 0 || n_heads <= 0
        || (num_seqs > 0 && (!chunk_starts || !chunk_lens || !block_tables || !q || !out)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_prefill: bad arguments");
    if (num_seqs == 0)
        return NVK_OK;
    long long total = 0;
    for (int s = 0; s < num_seqs; s++) {
        if (chunk_starts[s] < 0 || chunk_lens[s] < 0
            || (long long)chunk_starts[s] + chunk_lens[s] > (long long)max_blocks * pool->block_size)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_prefill: chunk %d (start %d, length %d) does not fit its table row",
                             s, chunk_starts[s], chunk_lens[s]);
        total += chunk_lens[s];
    }
    rc = validate_chunk_tables(pool, num_seqs, block_tables, max_blocks, chunk_starts, chunk_lens, "nvk_prefill");
    if (rc != NVK_OK)
        return rc;
    if (total == 0)
        return NVK_OK;
    NVK_SCOPE(pool->device);
    const size_t q_b   = (size_t)total * n_heads * pool->head_dim * sizeof(float);
    const size_t tab_b = (size_t)num_seqs * max_blocks * sizeof(int);
    auto         al    = [](size_t b) { return (b + 255) / 256 * 256; };
    rc                 = ensure_stage(pool, 2 * al(q_b) + al(tab_b));
    if (rc != NVK_OK)
        return rc;
    float *dq   = (float *)pool->stage;
    float *dout = (float *)(pool->stage + al(q_b));
    int   *dtab = (int *)(pool->stage + 2 * al(q_b));
    HIP_TRY(hipMemcpyAsync(dq, q, q_b, hipMemcpyHostToDevice, pool->stream));
    HIP_TRY(hipMemcpyAsync(dtab, block_tables, tab_b, hipMemcpyHostToDevice, pool->stream));
    rc = nvk_prefill_device(pool, layer, num_seqs, n_heads, dq, dtab, max_blocks, chunk_starts, chunk_lens, dout, flags,
                            (void *)pool->stream);
    if (rc != NVK_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, q_b, hipMemcpyDeviceToHost, pool->stream));
    HIP_TRY(hipStreamSynchronize(pool->stream));
    return NVK_OK;
}

int nvk_bind_pool(nvk_pool *pool, int layer)
{
    if (!pool) {
        t_bound_pool = nullptr;
        return NVK_OK;
    }
    int rc = check_pool_layer(pool, layer, "nvk_bind_pool");
    if (rc != NVK_OK)
        return rc;
    t_bound_pool  = pool;
    t_bound_layer = layer;
    return NVK_OK;
}

int nvk_paged_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                        const int *block_table, float *att_scores, int num_tokens, int block_size, int head_dim,
                        int n_heads, int n_kv_heads)
{
    (void)att_scores; // the reference's [n_heads, num_tokens] scratch has no device counterpart
    if (!out || !q || num_tokens < 0 || block_size <= 0 || head_dim <= 0 || n_heads <= 0 || n_kv_heads <= 0
        || (num_tokens > 0 && !block_table))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_paged_attention: bad arguments");
    const int nblk = std::max(1, (num_tokens + block_size - 1) / block_size);

    if (t_bound_pool) {
        nvk_pool *pool = t_bound_pool;
        if (pool->block_size != block_size || pool->head_dim != head_dim || pool->n_kv_heads != n_kv_heads)
            return nvk::fail(NVK_ERR_INVALID_ARG,
                             "nvk_paged_attention: bound pool is (bs=%d, hd=%d, kv=%d), call says (%d, %d, %d)",
                             pool->block_size, pool->head_dim, pool->n_kv_heads, block_size, head_dim, n_kv_heads);
        if (num_tokens == 0) {
            std::memset(out, 0, (size_t)n_heads * head_dim * sizeof(float)); // attention.hpp:98, nothing to launch
            return NVK_OK;
        }
        return nvk_decode(pool, t_bound_layer, 1, n_heads, q, block_table, nblk, &num_tokens, out, NVK_DECODE_AUTO, 0);
    }

    // Unbound: key_cache/value_cache are host slabs of one layer.  Stage the blocks
    // this sequence names into a one-layer scratch pool and run the same kernels.
    if (!key_cache || !value_cache)
        return nvk::fail(NVK_ERR_UNBOUND, "nvk_paged_attention: no pool bound and no host cache given");
    if (num_tokens == 0) {
        std::memset(out, 0, (size_t)n_heads * head_dim * sizeof(float));
        return NVK_OK;
    }
    for (int i = 0; i < nblk; i++)
        if (block_table[i] < 0)
            return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_paged_attention: block_table[%d]=%d is negative", i, block_table[i]);
    const size_t     row = (size_t)n_kv_heads * head_dim;
    const size_t     blk = (size_t)block_size * row;
    std::vector<int> local((size_t)nblk);

    if (t_host_policy == NVK_HOST_CACHE_APPEND_ONLY) {
        HostMirror *m  = nullptr;
        int         rc = mirror_for(key_cache, value_cache, block_size, n_kv_heads, head_dim, nblk, false, &m);
        if (rc != NVK_OK)
            return rc;
        nvk_pool   *mp = m->pool;
        NVK_SCOPE(mp->device);
        // not "the previous call plus one token", or the slab no longer holds what was staged last: trust nothing staged
        const bool same_sequence = !NVK_MIRROR_TABLE_CHECK
                                   || ((size_t)nblk >= m->last_table.size() && (size_t)nblk <= m->last_table.size() + 1
                                       && std::equal(m->last_table.begin(), m->last_table.end(), block_table));
        if (num_tokens != m->last_tokens + 1 || !same_sequence || !witness_holds(*m))
            std::fill(m->rows_staged.begin(), m->rows_staged.end(), 0);
        for (int i = 0; i < nblk; i++) {
            auto it = m->local_of.find(block_table[i]);
            if (it == m->local_of.end()) {
                if ((int)m->rows_staged.size() >= mp->num_blocks) { // more distinct blocks than the mirror holds
                    m->local_of.clear();
                    m->rows_staged.clear();
                    m->last_table.clear();
                    m->last_tokens = -1;
                    return nvk_paged_attention(out, q, key_cache, value_cache, block_table, att_scores, num_tokens,
                                               block_size, head_dim, n_heads, n_kv_heads); // remapped from scratch
                }
                it = m->local_of.emplace(block_table[i], (int)m->rows_staged.size()).first;
                m->rows_staged.push_back(0);
            }
            const int lb     = it->second;
            const int filled = std::min(block_size, num_tokens - i * block_size);
            int      &have   = m->rows_staged[(size_t)lb];
            if (have < filled) {
                const size_t at = (size_t)have * row, n = (size_t)(filled - have) * row;
                HIP_TRY(hipMemcpyAsync(mp->k_layer(0) + (size_t)lb * blk + at, key_cache + (size_t)block_table[i] * blk + at,
                                       n * sizeof(float), hipMemcpyHostToDevice, mp->stream));
                HIP_TRY(hipMemcpyAsync(mp->v_layer(0) + (size_t)lb * blk + at,
                                       value_cache + (size_t)block_table[i] * blk + at, n * sizeof(float),
                                       hipMemcpyHostToDevice, mp->stream));
                t_staged_bytes += (long long)(2 * n * sizeof(float));
                have = filled;
            }
            local[(size_t)i] = lb;
        }
        m->last_tokens = num_tokens;
        m->last_table.assign(block_table, block_table + nblk);
        keep_witness(*m, (size_t)block_table[(num_tokens - 1) / block_size] * blk + (size_t)((num_tokens - 1) % block_size) * row,
                     row);
        return nvk_decode(mp, 0, 1, n_heads, q, local.data(), nblk, &num_tokens, out, NVK_DECODE_AUTO, 0);
    }

    nvk_pool *sp = nullptr;
    int       rc = scratch_pool_for(block_size, n_kv_heads, head_dim, nblk, &sp);
    if (rc != NVK_OK)
        return rc;
    NVK_SCOPE(sp->device);
    for (int i = 0; i < nblk; i++) {
        // only the rows of the block that the sequence has reached (the last block is partly filled)
        const size_t rows = (size_t)std::min(block_size, num_tokens - i * block_size) * row;
        HIP_TRY(hipMemcpyAsync(sp->k_layer(0) + (size_t)i * blk, key_cache + (size_t)block_table[i] * blk,
                               rows * sizeof(float), hipMemcpyHostToDevice, sp->stream));
        HIP_TRY(hipMemcpyAsync(sp->v_layer(0) + (size_t)i * blk, value_cache + (size_t)block_table[i] * blk,
                               rows * sizeof(float), hipMemcpyHostToDevice, sp->stream));
        t_staged_bytes += (long long)(2 * rows * sizeof(float));
        local[(size_t)i] = i;
    }
    return nvk_decode(sp, 0, 1, n_heads, q, local.data(), nblk, &num_tokens, out, NVK_DECODE_AUTO, 0);
}

int nvk_standard_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                           float *att_scores, int pos, int head_dim, int n_heads, int n_kv_heads, int max_seq_len)
{
    (void)att_scores;
    if (!out || !q || !key_cache || !value_cache || head_dim <= 0 || n_heads <= 0 || n_kv_heads <= 0
        || max_seq_len <= 0 || pos >= max_seq_len)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_standard_attention: bad arguments (pos=%d, max_seq_len=%d)", pos,
                         max_seq_len);
    if (pos < 0) { // the reference's loops over t <= pos do nothing and `out` stays zeroed (attention.hpp:29,39)
        std::memset(out, 0, (size_t)n_heads * head_dim * sizeof(float));
        return NVK_OK;
    }
    // A contiguous [max_seq_len, n_kv_heads, head_dim] cache is a paged pool whose
    // table is the identity: rows 0..pos go to blocks 0.. of the staging pool in one copy each.
    constexpr int kBlock     = 16;
    const int     num_tokens = pos + 1;
    const int     nblk       = (num_tokens + kBlock - 1) / kBlock;
    std::vector<int> ident((size_t)nblk);
    for (int i = 0; i < nblk; i++)
        ident[(size_t)i] = i;
    const size_t row = (size_t)n_kv_heads * head_dim;

    if (t_host_policy == NVK_HOST_CACHE_APPEND_ONLY) {
        // the contiguous cache only grows at its end (model.hpp:211-217): keep rows [0, staged) on the device
        HostMirror *m  = nullptr;
        int         rc = mirror_for(key_cache, value_cache, kBlock, n_kv_heads, head_dim, nblk, true, &m);
        if (rc != NVK_OK)
            return rc;
        NVK_SCOPE(m->pool->device);
        if (m->rows_staged.empty())
            m->rows_staged.push_back(0); // one counter: rows of the contiguous cache already on the device
        int &have = m->rows_staged[0];
        if (num_tokens != m->last_tokens + 1 || !witness_holds(*m))
            have = 0;
        if (have < num_tokens) {
            const size_t at = (size_t)have * row, n = (size_t)(num_tokens - have) * row;
            HIP_TRY(hipMemcpyAsync(m->pool->k_layer(0) + at, key_cache + at, n * sizeof(float), hipMemcpyHostToDevice,
                                   m->pool->stream));
            HIP_TRY(hipMemcpyAsync(m->pool->v_layer(0) + at, value_cache + at, n * sizeof(float), hipMemcpyHostToDevice,
                                   m->pool->stream));
            t_staged_bytes += (long long)(2 * n * sizeof(float));
            have = num_tokens;
        }
        m->last_tokens = num_tokens;
        keep_witness(*m, (size_t)(num_tokens - 1) * row, row);
        return nvk_decode(m->pool, 0, 1, n_heads, q, ident.data(), nblk, &num_tokens, out, NVK_DECODE_AUTO, 0);
    }

    nvk_pool *sp = nullptr;
    int       rc = scratch_pool_for(kBlock, n_kv_heads, head_dim, nblk, &sp);
    if (rc != NVK_OK)
        return rc;
    NVK_SCOPE(sp->device);
    const size_t bytes = (size_t)num_tokens * row * sizeof(float);
    HIP_TRY(hipMemcpyAsync(sp->k_layer(0), key_cache, bytes, hipMemcpyHostToDevice, sp->stream));
    HIP_TRY(hipMemcpyAsync(sp->v_layer(0), value_cache, bytes, hipMemcpyHostToDevice, sp->stream));
    t_staged_bytes += (long long)(2 * bytes);
    return nvk_decode(sp, 0, 1, n_heads, q, ident.data(), nblk, &num_tokens, out, NVK_DECODE_AUTO, 0);
}

} // extern "C"
