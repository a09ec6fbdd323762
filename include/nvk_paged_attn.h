/* nvk_paged_attn.h -- C ABI of libnvk_paged_attn.so
 *
 * MI355X (gfx950) paged-attention decode behind nano-vllm's attention
 * interface.  Plain pointers and sizes only; no C++ or torch types cross this
 * boundary and no exception ever leaves the library.  Every function returns
 * an int status (NVK_OK == 0, negative on failure) unless noted; the message
 * for the calling thread's last failure is nvk_last_error().
 *
 * "ref:" comments name the reference interface each entry point replaces
 * (paths relative to the reference tree, aerlabsAI/nano-vllm @ 2026-01-30).
 *
 * Threading: a pool is thread-compatible (external synchronisation per pool);
 * distinct pools (one per GPU) may be driven from different host threads.
 * Streams: the split-K partials of decode and prefill calls on one pool live
 * in ONE workspace owned by the pool -- and so do the arrival counters of the
 * in-kernel combine, which every launch finds at zero and leaves at zero -- so
 * launches on a pool must be ordered with respect to each other (one stream,
 * or events between streams): two decode launches of one pool that overlap in
 * time would share partials and counters, and rows of `out` could stay
 * unwritten.  Pools are independent of each other.
 * nvk_bm_* is safe exactly where the reference's BlockManager is: the
 * *_for_request / *_safe entry points lock, the plain ones do not.
 */
#ifndef NVK_PAGED_ATTN_H
#define NVK_PAGED_ATTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
#define NVK_OK 0
#define NVK_ERR_INVALID_ARG (-1) /* null pointer, negative size, geometry the pool does not have */
#define NVK_ERR_OOM (-2)         /* device allocation failed, or no free KV blocks               */
#define NVK_ERR_HIP (-3)         /* a HIP runtime call failed; text in nvk_last_error()          */
#define NVK_ERR_BAD_BLOCK (-4)   /* block id outside [0, num_blocks)                             */
#define NVK_ERR_NO_DEVICE (-5)   /* no gfx950 device visible                                     */
#define NVK_ERR_UNBOUND (-6)     /* nvk_paged_attention() called with no pool bound/stageable    */

/* decode flags */
#define NVK_DECODE_AUTO 0        /* library picks chunking from the batch shape      */
#define NVK_DECODE_SINGLE_PASS 1 /* one chunk per sequence, no workspace, no merge   */
#define NVK_DECODE_SPLIT_K 2     /* flash-decoding: chunk_tokens per workgroup+merge */
/* modifiers, OR-ed onto the mode above (default: the library decides).
 * Where a sequence is cut into chunks, the chunks' partial (max, sum, output) triples are combined as the
 * reference's softmax would have (include/core/attention.hpp:126-140).  Up to 16 chunks per sequence that
 * happens inside the decode kernel -- the workgroup that finishes a sequence's last outstanding chunk merges
 * them -- beyond that by a merge kernel that spreads a sequence's chunks over many CUs.  Same bits either way
 * for the chunk-ordered form (MERGE_SEPARATE), which the tests hold the in-kernel form to. */
#define NVK_DECODE_MERGE_IN_KERNEL 0x100 /* combine inside the decode kernel whatever the chunk count      */
#define NVK_DECODE_MERGE_SEPARATE 0x200  /* combine by nvk_merge_kernel (one thread per output float4)     */
/* The work list (plan) only ORDERS the chunks, longest first; for a batch whose sequences all have the same
 * length that order is the natural one and the plan kernel is skipped: */
#define NVK_DECODE_UNIFORM 0x400 /* every sequence holds exactly max_ctx tokens; ctx_lens_dev may be NULL   */
#define NVK_DECODE_NO_PLAN 0x800 /* ragged lengths, read by the decode kernel itself; natural chunk order   */
/* A plan built with NVK_DECODE_NO_PLAN holds no copy of the lengths: it keeps the caller's ctx_lens_dev POINTER, and
 * every nvk_decode_planned (and combine kernel) of that plan reads the array again.  The array must stay allocated and
 * unchanged for as long as the plan is used -- hipGraph replays of such launches included.  (A plan built without the
 * modifier is self-contained: its work list was computed from the lengths at build time.) */
/* GQA groups of 5..16 q-heads over head_dim 64 / 128 run their dot products on fp32 MFMA tiles (exact fp32, another
 * order of sums than the vector kernel).  This modifier keeps such a launch on the vector kernel (A/B, tests). */
#define NVK_DECODE_VECTOR_ONLY 0x1000
/* Rows of 64 or 192 B (head_dim 16 / 48: the reference's own stories15M geometry, include/core/attention.hpp:114-115
 * fixes the row at head_dim floats) share their 128-B lines with the neighbour kv-head; with an even number of
 * kv-heads one wave streams a PAIR of heads -- whole lines, each read once.  Same arithmetic per head (tokens meet
 * in another order: same tolerance, not the same bits).  This modifier keeps such a launch on one head per wave
 * (A/B, tests). */
#define NVK_DECODE_NO_PAIRS 0x2000
/* The vector kernel keeps the waves of a workgroup -- the kv-heads (or token parts) of the same token rows -- in step
 * with a barrier per double-step, and walks chunks of 1024 tokens and more in 16-token steps; both are about WHEN rows
 * are asked for, not about what is computed.  This modifier launches the free-running kernel with 8-token steps instead
 * (the code path of pools with more kv-heads than a workgroup has waves), where it exists: same bits for 8-token-step
 * shapes, same tolerance otherwise (A/B, tests). */
#define NVK_DECODE_FREE_RUNNING 0x4000
/* Whole 128-float rows of up to 8 kv-heads (an even number, up to 4 q-heads each) as a PAIR of kv-heads per wave in
 * 4-wave workgroups, two of which fit a CU: a workgroup's successor then starts beside a running one instead of behind
 * it.  For launches made of many short sequences (generations of workgroups, each paying its start and its end).
 * Same arithmetic per head; tokens meet in another order (same tolerance, not the same bits). */
#define NVK_DECODE_ROW_PAIRS 0x8000

/* prefill flags */
#define NVK_PREFILL_AUTO 0    /* library picks by chunk length                                      */
#define NVK_PREFILL_ROWWISE 1 /* every query token is a decode row of its own: HBM-bound, short chunks */
#define NVK_PREFILL_TILED 2   /* 32-query x 32-key fp32 MFMA tiles: K/V read once per tile, long chunks */

/* pointer kinds for the *_ex entry points */
#define NVK_PTR_HOST 0
#define NVK_PTR_DEVICE 1

typedef struct nvk_pool          nvk_pool;          /* device-resident paged KV pool */
typedef struct nvk_block_manager nvk_block_manager; /* host-side block allocator     */
typedef struct nvk_plan          nvk_plan;          /* a batch's decode work list    */

const char *nvk_version(void);
/* thread-local, never NULL; valid until the thread's next failing call */
const char *nvk_last_error(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int nvk_device_count(void);

/* ------------------------------------------------------------ KV pool ----
 * ref: Config paged fields        include/core/model.hpp:33-36
 *      RunState::paged_*_cache    include/core/model.hpp:94-97
 *      initialize_paged_attention include/core/model.hpp:249-277
 * Inside a layer's K slab and V slab the layout is the reference's, unchanged, so block tables built by a
 * reference-equivalent BlockManager index it directly:
 *   one slab = [num_blocks, block_size, n_kv_heads, head_dim] fp32.
 * Where the slabs lie is the library's business (one allocation, a layer's V slab right behind its K slab: the two
 * streams of a layer-call run measurably faster side by side than half a pool apart); callers that need addresses
 * ask nvk_pool_layer_ptrs per layer.  Zero-initialised like the reference's std::vector::resize. */
int nvk_pool_create(int n_layers, int num_blocks, int block_size, int n_kv_heads, int head_dim, int device,
                    nvk_pool **out_pool);
int nvk_pool_destroy(nvk_pool *pool);
/* out6 = {n_layers, num_blocks, block_size, n_kv_heads, head_dim, device} */
int nvk_pool_geometry(const nvk_pool *pool, int *out6);
/* device base addresses of one layer's K and V slabs (for callers that fill
 * the pool with their own kernels, e.g. a fused RoPE+append) */
int nvk_pool_layer_ptrs(const nvk_pool *pool, int layer, void **k_dev, void **v_dev);
/* the pool's own HIP stream (hipStream_t; non-blocking).  It changes only when a model that owns the pool had a graph
 * recording invalidated from outside the library and took a fresh one (Threads): ask again then. */
int nvk_pool_stream(const nvk_pool *pool, void **stream);
/* Counts the (re)allocations of the split-K workspace: launches recorded in a hipGraph hold its address. */
long long nvk_pool_workspace_generation(const nvk_pool *pool);
int nvk_pool_synchronize(nvk_pool *pool);

/* KV append of one token row, host pointers.
 * ref: the two memcpys in LlamaModel::forward, include/core/model.hpp:190-207.
 * k_row / v_row: [n_kv_heads * head_dim] floats. */
int nvk_pool_append(nvk_pool *pool, int layer, int physical_block, int block_offset, const float *k_row,
                    const float *v_row);

/* Batched KV append: one new token per sequence, scattered by a HIP kernel.
 * Row s goes to block_tables[s][positions[s] / block_size], offset
 * positions[s] % block_size (ref: include/core/model.hpp:190-199, applied per
 * sequence).  k_rows/v_rows: [num_seqs, n_kv_heads*head_dim]; block_tables:
 * row-padded [num_seqs, max_blocks]; ptr_kind says where all four live. */
int nvk_pool_append_batch(nvk_pool *pool, int layer, int num_seqs, const float *k_rows, const float *v_rows,
                          const int *block_tables, int max_blocks, const int *positions, int ptr_kind);

/* RoPE fused with the batched KV append: for each sequence s, with
 * pos = positions[s], rotate q[s] IN PLACE and rotate k_rows[s] on its way into the pool slot
 * of token pos; v_rows[s] is copied there unchanged.  k_rows/v_rows are only read.
 * ref: Ops::apply_rope, include/ops/positional.hpp:13-39, called at include/core/model.hpp:169-175,
 *      followed by the append at include/core/model.hpp:190-207.
 *   q       [num_seqs, n_heads, head_dim]       k_rows, v_rows [num_seqs, n_kv_heads*head_dim]
 *   rope_theta  the reference's config.rope_theta (include/core/model.hpp:40: 10000);
 *               0 = no rotation: a plain batched append on `stream` (q is not touched, may be NULL)
 * The frequencies 1/powf(theta, i/head_dim) come from the host libm (a 1-ulp difference there is
 * amplified by pos); cos/sin are evaluated in fp64 on the device; the rotation is unfused fp32.
 * ptr_kind NVK_PTR_HOST: validates, stages, runs, copies q back, synchronises.
 * NVK_PTR_DEVICE: asynchronous on `stream` (NULL = the pool's stream). */
int nvk_pool_rope_append_batch(nvk_pool *pool, int layer, int num_seqs, int n_heads, float *q, const float *k_rows,
                               const float *v_rows, const int *block_tables, int max_blocks, const int *positions,
                               float rope_theta, int ptr_kind, void *stream);

/* Prebuilds the frequency table for `rope_theta` (a pool is created with the table for 10000, the
 * reference's default).  The RoPE entry points rebuild it themselves when they meet another theta, which
 * synchronises the device and is not allowed under stream capture: call this first in that case. */
int nvk_pool_set_rope_theta(nvk_pool *pool, float rope_theta);

/* The same for CHUNKS of new tokens: sequence s brings chunk_lens[s] tokens at
 * positions chunk_starts[s] + b; rows of q / k_rows / v_rows are packed back to back in sequence order
 * (sum of chunk_lens rows in all).  chunk_starts / chunk_lens are always HOST arrays; ptr_kind says
 * where q, k_rows, v_rows and block_tables live.
 * ref: BatchOps::batch_rope, experimental/chunked_prefill/batch_ops.hpp:25-56, and the append loop of
 *      LlamaModelChunked::forward_chunk, experimental/chunked_prefill/model_chunked.hpp:79-94 (contiguous
 *      cache there; through the block table here, as include/core/model.hpp:190-207 does per token). */
int nvk_pool_rope_append_chunks(nvk_pool *pool, int layer, int num_seqs, int n_heads, float *q, const float *k_rows,
                                const float *v_rows, const int *block_tables, int max_blocks, const int *chunk_starts,
                                const int *chunk_lens, float rope_theta, int ptr_kind, void *stream);

/* ------------------------------------------- producing q / k / v on the device --
 * SURVEY.md 8f row f1, second half: the rows the RoPE + append + attention entry points consume are made
 * on the GPU, so a layer's attention input never crosses PCIe.  All pointers are DEVICE pointers on
 * `device`; asynchronous on `stream` (NULL = that device's null stream); nothing is allocated.  fp32
 * throughout; sums are associated differently from the reference's serial loops (tolerance, not bits).
 *
 * nvk_batch_rms_norm_device   out[b][i] = in[b][i] * rms_b * weight[i], rms_b = 1/sqrtf(sum_i in[b][i]^2/dim + eps)
 *   ref: BatchOps::batch_rms_norm, experimental/chunked_prefill/batch_ops.hpp:58-76 = Ops::rms_norm per row,
 *        include/ops/normalization.hpp:13-23 (eps 1e-5f there).  out may alias in.
 * nvk_batch_matmul_device     out[b][i] = dot(in[b], weight[i]); weight [out_dim, in_dim] row-major
 *   ref: BatchOps::batch_matmul, batch_ops.hpp:8-23 = Ops::matmul per row, include/ops/linear.hpp:12-22.
 *        in_dim a multiple of 4, in/weight 16-byte aligned.  Up to 96 rows the weights are streamed once
 *        (HBM-bound); more: 64x64 or 128x128 tiles on fp32 MFMA.  Where those tiles are fewer than the CUs the
 *        contraction is cut into slices (a workgroup per tile and slice, added in slice order: the same bits
 *        every time; or the k-steps of all tiles are dealt evenly over the CUs, stream-K); the partial tiles live
 *        in a buffer the library keeps per (device, stream) -- 16 MiB on
 *        256 CUs, allocated the first time a stream needs it and never moved (a captured graph may hold its
 *        address) -- not under stream capture, where a stream without one takes the uncut path.
 * nvk_qkv_project_device      the head of a layer, include/core/model.hpp:161-166 (forward_chunk:
 *        model_chunked.hpp:57-77): xb = rms_norm(x, rms_weight); q = wq.xb, k = wk.xb, v = wv.xb in ONE
 *        launch.  x [batch, dim]; xb [batch, dim] scratch (rms_weight NULL: x is used as it is and xb may be
 *        NULL); wq [q_dim, dim], wk / wv [kv_dim, dim]; q [batch, q_dim], k / v [batch, kv_dim] -- the
 *        layouts nvk_pool_rope_append_batch / _chunks take. */
int nvk_batch_rms_norm_device(float *out_dev, const float *in_dev, const float *weight_dev, int batch_size, int dim,
                              float eps, int device, void *stream);
/* x += addend (the residual add, include/core/model.hpp:227-228, 238-239), then out = rms_norm(x): the add that
 * precedes every norm but a model's first, fused into it.  out must not alias x. */
int nvk_batch_add_rms_norm_device(float *out_dev, float *x_dev, const float *addend_dev, const float *weight_dev,
                                  int batch_size, int dim, float eps, int device, void *stream);
int nvk_batch_matmul_device(float *out_dev, const float *in_dev, const float *weight_dev, int batch_size, int in_dim,
                            int out_dim, int device, void *stream);
/* The gated half of the feed-forward block, model.hpp:232-234: hb = silu(w_gate . in) * (w_up . in), silu(g) =
 * g / (1 + expf(-g)) (Ops::swiglu, include/ops/activation.hpp:34-40).  hb_dev [batch_size, hidden_dim] receives the
 * result; hb2_dev (same shape, distinct) is scratch whose contents are unspecified afterwards.  With one or two rows
 * SwiGLU runs in the epilogue of the GEMV (one launch; the bits of the two products followed by SwiGLU); with more
 * rows it is nvk_batch_matmul2_device followed by the element-wise kernel. */
int nvk_batch_swiglu_ffn_device(float *hb_dev, float *hb2_dev, const float *in_dev, const float *w_gate_dev,
                                const float *w_up_dev, int batch_size, int in_dim, int hidden_dim, int device, void *stream);
/* two products over the same activations in one launch (w_gate and w_up, include/core/model.hpp:232-233) */
int nvk_batch_matmul2_device(float *out_a_dev, float *out_b_dev, const float *in_dev, const float *weight_a_dev,
                             const float *weight_b_dev, int batch_size, int in_dim, int out_dim_a, int out_dim_b,
                             int device, void *stream);
int nvk_qkv_project_device(float *q_dev, float *k_dev, float *v_dev, const float *x_dev, float *xb_dev,
                           const float *rms_weight_dev, const float *wq_dev, const float *wk_dev, const float *wv_dev,
                           int batch_size, int dim, int q_dim, int kv_dim, float rms_eps, int device, void *stream);

/* Whole-block transfers, host pointers, [block_size, n_kv_heads, head_dim] floats each. */
int nvk_pool_upload_block(nvk_pool *pool, int layer, int physical_block, const float *k_block,
                          const float *v_block);
int nvk_pool_download_block(nvk_pool *pool, int layer, int physical_block, float *k_block, float *v_block);

/* The KV gather on its own (bit-exactness check of the addressing at
 * include/core/attention.hpp:109-115,145-151): copies, with the same HIP
 * address arithmetic the decode kernel uses, the num_tokens rows of kv_head
 * that block_table addresses into k_out/v_out [num_tokens, head_dim] (host). */
int nvk_pool_gather(nvk_pool *pool, int layer, const int *block_table, int num_tokens, int kv_head, float *k_out,
                    float *v_out);

/* Synthetic fill on the device: element i of the layer's K slab becomes
 * value(seed_k, i), of the V slab value(seed_v, i), value() being the
 * counter-based generator of SURVEY.md 8d / oracle_fill_uniform(). */
int nvk_pool_fill_uniform(nvk_pool *pool, int layer, uint64_t seed_k, uint64_t seed_v);

/* ------------------------------------------------------ decode attention --
 * Batched paged-attention decode: one new token for each of num_seqs
 * sequences in one layer (a "layer-call").  Per sequence the result equals
 * ref: Attention::paged_attention, include/core/attention.hpp:82-159, called
 * as in LlamaModel::attention, include/core/model.hpp:393-416.
 *   q, out        [num_seqs, n_heads, head_dim] fp32
 *   block_tables  [num_seqs, max_blocks] int32, row-padded; entries beyond
 *                 ceil(ctx_lens[s]/block_size) are ignored
 *   ctx_lens      [num_seqs] tokens to attend to (pos + 1); 0 gives a zero row
 *   flags         NVK_DECODE_*; chunk_tokens only for NVK_DECODE_SPLIT_K
 *                 (rounded to the kernel's step; 0 = library default)
 * Host-pointer form: validates every used table entry against num_blocks
 * (NVK_ERR_BAD_BLOCK), stages inputs, runs, copies out back, synchronises. */
int nvk_decode(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q, const int *block_tables,
               int max_blocks, const int *ctx_lens, float *out, int flags, int chunk_tokens);

/* Device-pointer form: asynchronous on `stream` (hipStream_t; NULL = the
 * pool's stream).  max_ctx >= every ctx_lens[s] sizes the launch (the lengths
 * themselves are read on the device).  Table entries are clamped to
 * [0, num_blocks) on the device rather than validated. */
int nvk_decode_device(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                      const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                      float *out_dev, int flags, int chunk_tokens, void *stream);

/* nvk_decode_device for callers that cannot allocate.  No counterpart in the reference (its attention function
 * takes a caller-owned scratch array, include/core/attention.hpp:86, and owns nothing): this one is for callers
 * that must not reach hipMalloc / hipFree -- a stream under hipGraph capture, a tight serving loop, a harness that
 * forbids allocation.  Same arguments, same kernels and same results as nvk_decode_device; the work list, the
 * split-K partials and the arrival counters live in a static device array of nvk_noalloc_scratch_bytes()
 * (24 MiB, held on every GPU the library runs on until the process ends) instead of buffers grown on demand.
 * A batch that needs more (rows x chunks x n_heads x (head_dim + 2) floats, plus the work list) is refused with
 * NVK_ERR_INVALID_ARG and a message saying how much it needs.  Calls on one GPU share the array: two calls that
 * overlap in time on one GPU (different streams, different threads, different pools) race; calls ordered on one
 * stream, or separated by a wait, do not. */
int nvk_decode_device_noalloc(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                              const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                              float *out_dev, int flags, int chunk_tokens, void *stream);
size_t nvk_noalloc_scratch_bytes(void);

/* Paged chunked-prefill attention: causal attention of each sequence's chunk of
 * new tokens over the paged pool.  Row (s, b) of out equals Attention::paged_attention
 * (include/core/attention.hpp:82-159) with num_tokens = chunk_starts[s] + b + 1 over table row s --
 * ref: LlamaModelChunked::chunked_attention, experimental/chunked_prefill/model_chunked.hpp:167-224
 * (query b attends to t <= start_pos + b, :186), which only knows the contiguous cache.
 * The chunk's own K/V rows must already be in the pool (nvk_pool_rope_append_chunks).
 *   q, out  [sum(chunk_lens), n_heads, head_dim], packed in sequence order
 *   chunk_starts, chunk_lens  HOST arrays [num_seqs] (the scheduler knows them: Request::current_pos,
 *                             include/scheduler/request.hpp:70)
 * nvk_prefill: host pointers, validates the tables, synchronises.  nvk_prefill_device: q / tables / out
 * are device pointers, asynchronous on `stream` (NULL = the pool's stream). */
int nvk_prefill(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q, const int *block_tables,
                int max_blocks, const int *chunk_starts, const int *chunk_lens, float *out, int flags);
int nvk_prefill_device(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                       const int *block_tables_dev, int max_blocks, const int *chunk_starts, const int *chunk_lens,
                       float *out_dev, int flags, void *stream);

/* Pre-size the split-K workspace and the pool's own plan so later
 * nvk_decode_device calls of up to this shape do not allocate. */
int nvk_pool_reserve_workspace(nvk_pool *pool, int num_seqs, int n_heads, int max_ctx, int chunk_tokens);

/* Decode plans.  nvk_decode_device first turns the batch's context lengths
 * into a work list on the device (one small kernel: chunks of every sequence,
 * longest first, so ragged batches drain evenly over the CUs) and then runs
 * it.  The list depends on the lengths only, not on the layer: an engine
 * builds it once per model step with nvk_plan_build and runs every layer with
 * nvk_decode_planned -- the role the reference's per-step bookkeeping
 * (positions and block tables shared by all layers, include/core/model.hpp:157-221)
 * plays on the CPU.  Build and use a plan on the same stream (or order the
 * streams yourself).  nvk_plan_build may allocate; nvk_decode_planned never does.
 * With NVK_DECODE_NO_PLAN in `flags` the plan refers to ctx_lens_dev instead of
 * copying it: see the lifetime note at that flag.
 * ctx_lens_dev is a device pointer.  out4 = {chunk_tokens, max_chunks,
 * workgroup bound, kv-heads per workgroup}. */
int nvk_plan_create(nvk_pool *pool, nvk_plan **out_plan);
int nvk_plan_destroy(nvk_plan *plan);
int nvk_plan_build(nvk_plan *plan, int num_seqs, int n_heads, const int *ctx_lens_dev, int max_ctx, int flags,
                   int chunk_tokens, void *stream);
int nvk_plan_info(const nvk_plan *plan, int *out4);
int nvk_decode_planned(nvk_pool *pool, int layer, const nvk_plan *plan, const float *q_dev,
                       const int *block_tables_dev, int max_blocks, float *out_dev, void *stream);

/* HIP-event timing of the device form, on the pool's stream (the stream the
 * kernels are launched on): `warmup` untimed layer-calls, a stream
 * synchronise, then ONE timed region of `iters` layer-calls.  Call i rebuilds
 * the plan when i % replan_every == 0 (1 = every call, exactly
 * nvk_decode_device; L = once per L calls, as for the L layers of a model step).
 *   total_ms  region time / iters (plan + attention + merge kernels + gaps)
 *   attn_ms   mean over the same region of the attention kernel's own
 *             duration, from an event pair around each of its launches
 *             (around every 8th when iters > 32)
 * Either may be NULL; iters == 0 only warms up.  Returns after the region has
 * finished on the device. */
int nvk_decode_bench(nvk_pool *pool, int layer, int num_seqs, int n_heads, const float *q_dev,
                     const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                     float *out_dev, int flags, int chunk_tokens, int replan_every, int warmup, int iters,
                     float *total_ms, float *attn_ms);

/* The same with call i reading layer first_layer + i % n_rotate of the pool (identical batches over n_rotate
 * different slabs): no call can find its K/V in the 256 MiB Infinity Cache left there by the call before. */
int nvk_decode_bench_layers(nvk_pool *pool, int first_layer, int n_rotate, int num_seqs, int n_heads,
                            const float *q_dev, const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev,
                            int max_ctx, float *out_dev, int flags, int chunk_tokens, int replan_every, int warmup,
                            int iters, float *total_ms, float *attn_ms);
/* The same layer-calls as a model step would issue them under hipGraph replay: the plan is built once, then
 * layers_per_graph layer-calls (call i reading layer first_layer + i % n_rotate) are captured from the pool's stream
 * into ONE graph, which is launched `warmup` times untimed and `iters` times between a pair of HIP events.
 * total_ms = region / (iters x layers_per_graph): the STEP of a layer-call inside a replayed model step. */
int nvk_decode_bench_graph(nvk_pool *pool, int first_layer, int n_rotate, int num_seqs, int n_heads, const float *q_dev,
                           const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                           float *out_dev, int flags, int chunk_tokens, int layers_per_graph, int warmup, int iters,
                           float *total_ms);
/* Name of the attention kernel instantiation a launch of this shape runs (as rocprofv3 prints it), for matching
 * profiles to launches. */
int nvk_decode_kernel_name(nvk_pool *pool, int num_seqs, int n_heads, int max_ctx, int flags, int chunk_tokens,
                           char *buf, int cap);

/* Drop-in for the reference function, same eleven arguments, host pointers.
 * ref: Attention::paged_attention, include/core/attention.hpp:82-92.
 *  - With a pool bound on this thread (nvk_bind_pool), key_cache/value_cache
 *    are not dereferenced: the bound layer of the device pool is the cache
 *    (its geometry must match block_size/head_dim/n_kv_heads).
 *  - With no pool bound, key_cache/value_cache are the reference's host slabs
 *    for one layer: the blocks block_table names are staged to the device for
 *    this call (correct, PCIe-bound; for tests and first bring-up).
 *  att_scores is accepted and ignored (may be NULL).  Returns a status where
 *  the reference returns void. */
int nvk_paged_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                        const int *block_table, float *att_scores, int num_tokens, int block_size, int head_dim,
                        int n_heads, int n_kv_heads);
/* Drop-in for the reference's contiguous-cache twin, same ten arguments, host pointers.
 * ref: Attention::standard_attention, include/core/attention.hpp:16-25 (body :26-76),
 * called from LlamaModel::attention, include/core/model.hpp:419-430.
 * key_cache/value_cache: one layer's [max_seq_len, n_kv_heads, head_dim] fp32 slab;
 * tokens 0..pos are attended to (pos < 0 gives a zero row, as the reference's
 * empty loops do).  Rows 0..pos are staged to the device for this call and run
 * through the same kernels behind an identity block table.  att_scores ignored. */
int nvk_standard_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                           float *att_scores, int pos, int head_dim, int n_heads, int n_kv_heads, int max_seq_len);
/* How the host-slab forms of the two drop-in entry points bring K/V to the device (per thread).
 *  NVK_HOST_CACHE_RESTAGE (default): every call copies the rows it reads; always correct.
 *  NVK_HOST_CACHE_APPEND_ONLY: the caller promises what the reference engine does -- between two calls
 *    for the same slab, rows are only appended (include/core/model.hpp:190-207,211-217) and num_tokens
 *    grows by one (:397).  A device mirror per slab (keyed by the key_cache pointer) then keeps what it
 *    has, and only new rows cross PCIe: 2 rows per call instead of 2 * num_tokens.  A call that is not
 *    "the previous call on this slab plus one token" -- the same blocks in the same order (at most one
 *    more) and num_tokens one higher -- restages everything, so a new request needs no notice, also
 *    when it takes blocks another has just freed; rewriting rows in place between such calls is what
 *    the promise forbids.
 * Switching back to RESTAGE frees the mirrors.  nvk_host_cache_staged_bytes: K/V bytes the drop-in
 * entry points have copied host-to-device on this thread (reset != 0 clears the counter). */
#define NVK_HOST_CACHE_RESTAGE 0
#define NVK_HOST_CACHE_APPEND_ONLY 1
int       nvk_set_host_cache_policy(int policy);
long long nvk_host_cache_staged_bytes(int reset);
/* Bind (pool, layer) as this thread's cache for nvk_paged_attention; pool NULL unbinds.
 * ref: the layer slicing in LlamaModel::attention, include/core/model.hpp:400-403. */
int nvk_bind_pool(nvk_pool *pool, int layer);

/* ------------------------------------------------------- head-sharded decode --
 * The decode layer-call split over the G GPUs of one node by kv-head group.  The reference has one
 * process and no collective; heads are independent in it (q-head loop outermost, include/core/attention.hpp:102,
 * kv-head h / kv_mul, :105), so rank g holds kv-heads [g*KV/G, (g+1)*KV/G) of every sequence in its own pool of
 * [num_blocks, block_size, KV/G, head_dim] (same block tables on every rank), computes its q-heads' slice of out,
 * and ONE all-gather per layer-call joins the slices: gathered [G, num_seqs, H/G, head_dim], transpose-free.
 * The collective is RCCL's ncclAllGather over xGMI, called directly (librccl.so is opened by the first of these
 * calls).  One process per GPU: rank 0 asks nvk_shard_unique_id for 128 bytes and hands them to the other ranks by
 * whatever channel the engine has, then every rank calls nvk_shard_create (collective: returns when all have).
 * The sequence-sharded split needs none of this: its ranks share nothing.
 *   nvk_shard_head_range    out4 = {local q-heads, local kv-heads, first q-head, first kv-head}; host only
 *   nvk_shard_gather_bytes  bytes one rank contributes to the gather (x G = what it ends up holding); host only
 *   nvk_shard_info          out6 = {rank, world, local q-heads, local kv-heads, first q-head, first kv-head}
 *   nvk_shard_all_gather_out  the collective alone, on `stream` (NULL: the shard's own communication stream)
 *   nvk_shard_decode_gather   nvk_decode_device on `compute_stream` (NULL: the pool's), then the gather on the
 *       shard's communication stream behind an event: it runs under whatever `compute_stream` does next (the next
 *       layer-call's attention).  `buffer` (0 or 1) names which of two (out_local, gathered) buffer pairs the call
 *       uses; a pair is reused only after its gather has finished (the call orders that itself).
 *       IN PLACE: out_local_dev may be the rank's own slot of the result, gathered_dev + rank * (num_seqs x H/G x
 *       head_dim) floats (ncclAllGather's in-place form): the attention then writes its rows where the gather wants
 *       them and no copy of the rank's own rows is made -- with one rank nothing moves at all.
 *   nvk_shard_wait_gather     makes `stream` (NULL: the calling thread) wait until buffer's gather has landed --
 *       before the consumer of `gathered` (the wo projection, include/core/model.hpp:224) reads it.
 *   nvk_shard_set_gather_mode where nvk_shard_decode_gather runs the collective.  OVERLAPPED: on the shard's own
 *       stream, as described above (the hand-over costs the compute stream 8-9 us per layer-call: worth it wherever
 *       a real collective is hidden).  INLINE: on the compute stream right behind the attention (no hand-over,
 *       nothing hidden).  AUTO (default): OVERLAPPED with more than one rank, INLINE with one, where the "gather" is
 *       a device copy.  Same protocol for the caller either way (buffers, nvk_shard_wait_gather).
 *       nvk_shard_gather_mode says which of the two a call would use now. */
#define NVK_SHARD_GATHER_AUTO 0
#define NVK_SHARD_GATHER_OVERLAPPED 1
#define NVK_SHARD_GATHER_INLINE 2
typedef struct nvk_shard nvk_shard;
int       nvk_shard_head_range(int rank, int world, int n_heads, int n_kv_heads, int *out4);
long long nvk_shard_gather_bytes(int num_seqs, int n_heads, int head_dim, int world);
int       nvk_shard_unique_id(void *id128);
int       nvk_shard_create(int rank, int world, const void *id128, int device, int n_heads, int n_kv_heads,
                           nvk_shard **out_shard);
int       nvk_shard_destroy(nvk_shard *shard);
int       nvk_shard_info(const nvk_shard *shard, int *out6);
int       nvk_shard_all_gather_out(nvk_shard *shard, const float *out_local_dev, float *gathered_dev, int num_seqs,
                                   int head_dim, void *stream);
int       nvk_shard_decode_gather(nvk_shard *shard, nvk_pool *pool, int layer, int num_seqs, const float *q_local_dev,
                                  const int *block_tables_dev, int max_blocks, const int *ctx_lens_dev, int max_ctx,
                                  float *out_local_dev, float *gathered_dev, int buffer, int flags, int chunk_tokens,
                                  void *compute_stream);
int       nvk_shard_wait_gather(nvk_shard *shard, int buffer, void *stream);
int       nvk_shard_set_gather_mode(nvk_shard *shard, int mode);
int       nvk_shard_gather_mode(const nvk_shard *shard);

/* ------------------------------------------------------------ decode batch --
 * Per-request block tables kept on the device, and one append + attention call
 * per layer for all DECODING requests.
 * ref: ScheduledBatch::decode_requests  include/scheduler/scheduler.hpp:24-43,66-77 (formed, never executed:
 *      include/scheduler/batched_runner.hpp:84-85), Request::block_ids include/scheduler/request.hpp:74,
 *      the per-request allocator API include/scheduler/block_manager.hpp:145-171, and the per-token
 *      bookkeeping of LlamaModel::forward include/core/model.hpp:181-207,221 applied per request.
 * A block id names the same slot in every layer's slab, so one table and one plan serve all layers of a
 * step.  Not thread-safe; one batch per pool stream.  Step protocol:
 *   nvk_batch_begin_step(ids)      every listed request takes one token: a block is drawn for those at a
 *                                  block boundary (NVK_ERR_OOM, text "Out of memory: no free blocks", with
 *                                  nothing changed when the pool cannot cover them); tables, positions and
 *                                  context lengths of the step reach HBM (new entries only when the list is
 *                                  the previous step's); the decode plan is built.  Asynchronous on `stream`.
 *   nvk_batch_layer(layer, ...)    q [n, n_heads, hd] (rotated in place when rope_theta > 0), k_rows/v_rows
 *                                  [n, n_kv_heads*hd], out [n, n_heads, hd], all device pointers in the
 *                                  order of the step's list: RoPE + append, then attention over pos+1 tokens.
 *   nvk_batch_end_step()           lengths advance by one.
 * nvk_batch_finish_request waits for the device, returns the request's blocks to the manager
 * (BlockManager::free_request) and frees its slot. */
typedef struct nvk_batch nvk_batch;
int nvk_batch_create(nvk_pool *pool, nvk_block_manager *bm, int max_seqs, int max_blocks_per_seq,
                     nvk_batch **out_batch);
int nvk_batch_destroy(nvk_batch *batch);
int nvk_batch_add_request(nvk_batch *batch, int request_id);
int nvk_batch_finish_request(nvk_batch *batch, int request_id);
int nvk_batch_num_requests(const nvk_batch *batch);
int nvk_batch_request_len(const nvk_batch *batch, int request_id);                       /* tokens in the pool */
int nvk_batch_request_blocks(const nvk_batch *batch, int request_id, int *ids, int cap); /* count; = block_ids */
int nvk_batch_begin_step(nvk_batch *batch, const int *request_ids, int n, int n_heads, void *stream);
/* A step that also brings CHUNKS (chunked prefill beside decode, what ScheduledBatch's two lists describe,
 * include/scheduler/scheduler.hpp:24-43): request i takes chunk_lens[i] >= 1 tokens.  List the one-token
 * (decode) requests first; rows of q / k_rows / v_rows / out for nvk_batch_layer are then one per decode
 * request followed by the chunks' rows packed in list order (nvk_batch_step_rows in all).  Per layer the
 * decode part runs as above and the chunks run RoPE + append of all their tokens, then chunked-prefill
 * attention (experimental/chunked_prefill/model_chunked.hpp:79-96).  With every chunk_lens[i] == 1 this is
 * nvk_batch_begin_step. */
int nvk_batch_begin_step_chunks(nvk_batch *batch, const int *request_ids, const int *chunk_lens, int n, int n_heads,
                                void *stream);
int nvk_batch_step_rows(const nvk_batch *batch);
int nvk_batch_layer(nvk_batch *batch, int layer, float *q_dev, const float *k_rows_dev, const float *v_rows_dev,
                    float *out_dev, float rope_theta, void *stream);
/* The same layer from the hidden state (include/core/model.hpp:161-221 in one call): x [rows, dim] ->
 * rms_norm (rms_weight NULL: x is already normalised) -> q / k / v projections (nvk_qkv_project_device) ->
 * RoPE + append + attention as nvk_batch_layer.  q / k / v live in scratch rows the batch keeps (grown, with a
 * device synchronisation, the first time a step needs more).  A step whose rows all decode, up to 16 of them (up
 * to 96 for most shapes), runs the projections, the rotation and the append as ONE launch: the epilogue of the projection
 * kernel rotates q and k in their pairs and writes k and v straight to the rows' cache slots -- the arithmetic of the
 * separate kernels, so the same bits in `out` and in the pool. */
int nvk_batch_layer_hidden(nvk_batch *batch, int layer, const float *x_dev, int dim, const float *rms_weight_dev,
                           const float *wq_dev, const float *wk_dev, const float *wv_dev, float rms_eps, float *out_dev,
                           float rope_theta, void *stream);
int nvk_batch_end_step(nvk_batch *batch);
/* For callers that capture a step's nvk_batch_layer / nvk_batch_layer_hidden calls in a hipGraph and replay it on later
 * steps: out8 describes everything the captured launches bake in besides the caller's own pointers (rows, kind of
 * step, the plan's launch shape, the addresses of the library's scratch).  A graph captured under one signature may
 * be replayed in any later step whose signature is equal (pure decode steps keep theirs for up to 256 steps). */
int nvk_batch_graph_signature(const nvk_batch *batch, long long *out8);
/* Abandons the step in flight (a later call of the step failed): no request advances; blocks drawn for the step
 * stay with their requests. */
int nvk_batch_abort_step(nvk_batch *batch);
/* out2 = {steps that uploaded whole tables, steps that only patched new entries} */
int nvk_batch_stats(const nvk_batch *batch, long long *out2);

/* ------------------------------------------------------------- the tokenizer --
 * tokenizer.bin and the reference's BPE, host-only, same ids bit for bit.
 * ref: class Tokenizer, include/core/tokenizer.hpp -- load :31-61, decode :63-75, encode :77-131 (BOS = 1,
 * dummy-prefix " ", one lookup per byte with unknown bytes dropped, best-score pair merges, EOS = 2). */
typedef struct nvk_tokenizer nvk_tokenizer;
int  nvk_tokenizer_load(const char *path, int vocab_size, nvk_tokenizer **out_tokenizer);
void nvk_tokenizer_destroy(nvk_tokenizer *tokenizer);
int  nvk_tokenizer_vocab_size(const nvk_tokenizer *tokenizer);
int  nvk_tokenizer_max_token_length(const nvk_tokenizer *tokenizer);
/* Bytes of the piece (NUL-terminated copy of at most cap-1 of them in out); "" for ids outside the vocabulary. */
int  nvk_tokenizer_decode(const nvk_tokenizer *tokenizer, int token, char *out, int cap);
/* Number of ids of the encoding (at most cap written).  text_len < 0: strlen(text). */
int  nvk_tokenizer_encode(const nvk_tokenizer *tokenizer, const char *text, int text_len, int bos, int eos, int *ids,
                          int cap);

/* ------------------------------------------------------ the model on the device --
 * SURVEY.md 8f row f4 (checkpoint format) and the caller of the whole path: a llama2.c checkpoint resident in HBM
 * and LlamaModel::forward (include/core/model.hpp:150-247) as batched steps, nothing but token ids and results
 * crossing PCIe.  ref: LlamaModel::load / read_weights, include/core/model.hpp:118-148, 316-365 (7 x int32 header;
 * tensors grouped by parameter type; lm_head shared with the embedding when the file ends early, :349-364);
 * initialize_paged_attention, :249-277 (block_size / num_blocks: Config, :35-36); Sampler greedy branch,
 * include/core/sampler.hpp:27-29.  Sampling with temperature is not offered: the reference seeds it from the clock
 * (include/scheduler/request_processor.hpp:31-34), so only greedy streams are reproducible. */
typedef struct nvk_model nvk_model;
int nvk_model_load(const char *checkpoint_path, int block_size, int num_blocks, int max_seqs, int device,
                   nvk_model **out_model);
int nvk_model_destroy(nvk_model *model);
/* out8 = {dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, max_seq_len, shared_classifier} */
int nvk_model_config(const nvk_model *model, int *out8);
int nvk_model_add_request(nvk_model *model, int request_id);
int nvk_model_finish_request(nvk_model *model, int request_id); /* frees its KV blocks */
int nvk_model_request_len(const nvk_model *model, int request_id);
int nvk_model_free_blocks(const nvk_model *model);
/* Pure decode steps replay a hipGraph of their ~12 launches per layer (captured again whenever the batch, the launch
 * shape or a scratch address changes); on by default, results are the same bits either way.  0 switches it off. */
int nvk_model_use_graphs(nvk_model *model, int on);
/* A step of up to 16 decoding rows (the reference's own case is one: bound by launch latency, not by HBM) carries its norms between
 * the GEMVs instead of running them as kernels: the wo / w_down GEMVs add their product to the residual stream and
 * leave x * gain and partial sums of squares, the GEMV that follows multiplies its product by 1 / rms (6 launches per
 * layer instead of 8).  1 / rms after the dot product instead of before it: a few ulp from the separate norm, inside
 * the tolerance; everything else keeps its bits.  OFF by default (the bits of the separate norm kernels); 1 opts in. */
int nvk_model_use_linked_norms(nvk_model *model, int on);
/* One model step: request i takes the next chunk_lens[i] >= 1 tokens of `tokens` (packed in list order; a prompt
 * chunk, or 1 = decode).  next_tokens [n] (host, may be NULL) = greedy argmax after each request's last token;
 * logits [n, vocab_size] (host, may be NULL) = the classifier's output for those rows.  Synchronous. */
int nvk_model_step_chunks(nvk_model *model, const int *request_ids, const int *chunk_lens, const int *tokens, int n,
                          int *next_tokens, float *logits);
/* every chunk_lens[i] == 1 */
int nvk_model_step(nvk_model *model, const int *request_ids, const int *tokens, int n, int *next_tokens, float *logits);
/* LlamaModel::forward(token, pos) for ONE sequence (request 0, added on first use); pos must be its length */
int nvk_model_forward(nvk_model *model, int token, int pos, float *logits);

/* ------------------------------------------- the model on several GPUs: sequence-sharded --
 * SURVEY.md 8e, primary mode, at engine level: one nvk_model per entry of `devices` (its own copy of the weights, its
 * own pool, allocator and batch), each driven by a host thread of its own.  A request is placed on ONE member when it
 * is admitted -- the member whose running requests promise the fewest tokens, lowest index on ties -- and stays there:
 * the members share nothing, no collective exists on the data path, and every request's token stream is exactly the one
 * a single nvk_model produces for it.
 * ref: what is split is the scheduler's running list, include/scheduler/scheduler.hpp:66-103, with
 *      Request::block_ids (include/scheduler/request.hpp:74) private to each request; the attention itself touches one
 *      sequence at a time (include/core/attention.hpp:82-159).  The reference has one process and one device.
 * `devices` may name a device more than once (two members on one GPU: a rehearsal of two GPUs on a one-GPU box).
 *   nvk_model_group_add_request   expected_tokens = what the request promises (prompt + max_tokens): the balance;
 *                                 returns the member index (>= 0) or a negative status (NVK_ERR_OOM: every member full)
 *   nvk_model_group_step_chunks   nvk_model_step_chunks for a list that may span members: each member runs its share
 *                                 (in list order) on its own thread, all at once; results come back in list order. */
typedef struct nvk_model_group nvk_model_group;
int nvk_model_group_load(const char *checkpoint_path, int block_size, int num_blocks, int max_seqs_per_member,
                         int n_members, const int *devices, nvk_model_group **out_group);
int nvk_model_group_destroy(nvk_model_group *group);
int nvk_model_group_size(const nvk_model_group *group);
int nvk_model_group_config(const nvk_model_group *group, int *out8);
int nvk_model_group_member_device(const nvk_model_group *group, int member);
int nvk_model_group_member_of(const nvk_model_group *group, int request_id); /* -1: not running */
int nvk_model_group_add_request(nvk_model_group *group, int request_id, int expected_tokens);
int nvk_model_group_finish_request(nvk_model_group *group, int request_id);
int nvk_model_group_request_len(const nvk_model_group *group, int request_id);
int nvk_model_group_step_chunks(nvk_model_group *group, const int *request_ids, const int *chunk_lens, const int *tokens,
                                int n, int *next_tokens, float *logits);
int nvk_model_group_use_graphs(nvk_model_group *group, int on);

/* ---------------------------------------------------------- BlockManager --
 * Host-side allocator returning the same ids in the same order as
 * ref: BlockManager, include/scheduler/block_manager.hpp:14-283
 * (lowest free index first), in O(log64 n) instead of a linear scan.
 * Where the reference throws std::runtime_error the status is returned
 * instead and the state is left as the reference leaves it. */
int  nvk_bm_create(int num_blocks, int block_size, nvk_block_manager **out_bm); /* ref :17-24  */
void nvk_bm_destroy(nvk_block_manager *bm);
int  nvk_bm_allocate_block(nvk_block_manager *bm);                /* id, or -1 when full. ref :28-44 */
int  nvk_bm_free_block(nvk_block_manager *bm, int block_id);      /* NVK_ERR_BAD_BLOCK where ref throws. ref :48-62 */
/* count written to ids (capacity cap), or NVK_ERR_OOM where ref throws "Out of memory". ref :67-92 */
int  nvk_bm_allocate_sequence(nvk_block_manager *bm, int num_tokens, int *ids, int cap);
int  nvk_bm_free_sequence(nvk_block_manager *bm, const int *ids, int n); /* ref :96-101 */
int  nvk_bm_num_free_blocks(const nvk_block_manager *bm);               /* ref :107 */
int  nvk_bm_num_blocks(const nvk_block_manager *bm);                    /* ref :110 */
int  nvk_bm_block_size(const nvk_block_manager *bm);                    /* ref :113 */
int  nvk_bm_is_free(const nvk_block_manager *bm, int block_id);         /* ref :116-121 */
float nvk_bm_utilization(const nvk_block_manager *bm);                  /* ref :124 */
/* per-request, mutex-guarded, non-throwing. ref :132-216 */
int  nvk_bm_allocate_for_request(nvk_block_manager *bm, int request_id, int num_tokens, int *ids, int cap);
int  nvk_bm_allocate_block_for_request(nvk_block_manager *bm, int request_id);
void nvk_bm_free_request(nvk_block_manager *bm, int request_id);
/* Gives ONE block of a request back (no counterpart in the reference, whose per-request API only frees whole
 * requests, :157-171): the block leaves the request's list AND the used set in one locked step, so a caller that
 * has to undo part of a multi-block draw never leaves an id that a later free_request would free under another
 * owner.  NVK_ERR_BAD_BLOCK when the request does not hold it. */
int  nvk_bm_release_block_for_request(nvk_block_manager *bm, int request_id, int block_id);
int  nvk_bm_get_request_blocks(nvk_block_manager *bm, int request_id, int *ids, int cap);
int  nvk_bm_get_request_block_count(nvk_block_manager *bm, int request_id);
int  nvk_bm_num_active_requests(nvk_block_manager *bm);
int  nvk_bm_allocate_block_safe(nvk_block_manager *bm);
void nvk_bm_free_block_safe(nvk_block_manager *bm, int block_id);

#ifdef __cplusplus
}
#endif
#endif /* NVK_PAGED_ATTN_H */
