// nvk_internal.h -- shared by the translation units of libnvk_paged_attn.so.
#pragma once

#include <cstdarg>
#include <cstdio>
#include <new>

#include "nvk_paged_attn.h"

namespace nvk {

// Records `msg` (printf-style) as the calling thread's last error and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// This device's address of the static scratch array behind nvk_decode_device_noalloc (hipGetSymbolAddress, once per
// device; the calling thread's current device must be `device`).
int noalloc_scratch_on(int device, unsigned char **out);

// Counts the times a plan's device arrays have moved (a captured graph that launches the plan holds their addresses).
long long plan_generation(const nvk_plan *plan);

// ---- RoPE + KV append in the epilogue of the q / k / v GEMV (one or two rows) -----------------
// Where nvk_rope_append_kernel would put a step's rows: the layer's slabs, the step's tables and positions, the
// frequency table.  Filled by nvk_paged_attn.hip, consumed by nvk_linear.hip.
struct RopeAppendTarget
{
    float       *k_layer, *v_layer;
    const int   *block_tables, *positions; // [rows, max_blocks], [rows]
    const float *inv_freq;                 // [head_dim / 2]
    int          max_blocks, num_blocks, block_size, head_dim, row_len; // row_len = n_kv_heads * head_dim
    unsigned     bs_magic;
};
// ---- a norm carried from one GEMV to the next (one or two rows) --------------------------------
// The residual add + RMSNorm between two projections without a kernel of its own.  The producer (wo, w_down) adds its
// product to the residual stream and leaves xg = x * gain and, per 4-column tile, the sum of the four new squares
// (ss_out, gain_out, xg_out set).  The consumer reads xg as its input rows and multiplies its product by 1 / rms,
// rms from those partial sums (ss_in, ss_n set).  A few ulp from the separate norm (1 / rms after the dot product
// instead of before); nothing else changes.
struct NormLink
{
    const float *ss_in  = nullptr; // consumer: [rows, ss_n] partial sums of squares of the (un-gained) input rows
    int          ss_n   = 0;
    float        eps    = 0.f;
    const float *gain_out = nullptr; // producer: the next norm's weight [out_dim]
    float       *xg_out = nullptr;   // producer: [rows, out_dim]
    float       *ss_out = nullptr;   // producer: [rows, ss_out_n], ss_out_n = out_dim / 4 (filled in by linked_gemv)
    int          ss_out_n = 0;
};
// out = in . w0^T with the link applied (producer: out += ...; consumer: ... / rms); glu: silu(in . w0^T) * (in . w1^T)
int  linked_ss_n(int rows, int dim); // how many partial sums per row a producer of `dim` columns leaves
bool linked_gemv_ok(); // false in a build whose GEMV geometry has no linked form (NVK_GEMV_COLS != 4)
int  linked_gemv(const NormLink &link, bool glu, float *out_dev, const float *in_dev, const float *w0_dev,
                const float *w1_dev, int rows, int in_dim, int out_dim, int device, void *stream,
                float *scratch_dev = nullptr); // glu with more than two rows: [rows, out_dim], contents unspecified after

// After a graph recording on the pool's stream was invalidated from outside (another thread of the application put work
// on the null stream or waited for the device meanwhile): ROCm 7.2 leaves such a stream in the state "invalidated" for
// good -- every later call on it fails, ending the recording again is refused (tools/capture_recovery_probe.hip).  This
// waits for the device, gives the pool a fresh stream and destroys the old one.  NVK_OK also when the stream is fine.
int pool_renew_stream_if_stuck(nvk_pool *pool);

constexpr int kFusedRowsMax = 16; // the GEMV (1-2 rows) and the 16-row MFMA kernel carry the fused epilogues
int pool_rope_append_target(nvk_pool *pool, int layer, float rope_theta, const int *block_tables_dev, int max_blocks,
                            const int *positions_dev, RopeAppendTarget *out);
// q_dev [rows, q_dim] = rope(wq . x); rope(wk . x) and wv . x go straight to the cache slots of the rows' positions
// (model.hpp:164-175, 190-207 in one launch; the arithmetic of the separate kernels, so the same bits).
// qkv_rope_append_ok(rows, ...) must hold (up to 16 rows always; up to 96 where every row tile has its own workgroup).
int qkv_rope_append(const RopeAppendTarget &target, float *q_dev, const float *x_dev, const float *wq_dev,
                    const float *wk_dev, const float *wv_dev, int rows, int dim, int q_dim, int device, void *stream,
                    const NormLink *link = nullptr);
// nvk_batch_layer_hidden for one or two decoding rows whose input rows are a producer's xg (link: ss_in, ss_n, eps)
int batch_layer_hidden_linked(nvk_batch *b, int layer, const float *xg_dev, int dim, const NormLink &link,
                              const float *wq_dev, const float *wk_dev, const float *wv_dev, float *out_dev,
                              float rope_theta, void *stream);
bool qkv_rope_append_ok(int rows, int dim, int q_dim, int kv_dim, int device);

} // namespace nvk
