// nvk_internal.h -- shared by the translation units of libnvk_paged_attn.so.
#pragma once

#include <cstdarg>
#include <cstdio>
#include <new>

#include "nvk_paged_attn.h"

namespace nvk {

// Records `msg` (printf-style) as the calling thread's last error and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

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
constexpr int kQkvRopeMaxRows = 2;
int pool_rope_append_target(nvk_pool *pool, int layer, float rope_theta, const int *block_tables_dev, int max_blocks,
                            const int *positions_dev, RopeAppendTarget *out);
// q_dev [rows, q_dim] = rope(wq . x); rope(wk . x) and wv . x go straight to the cache slots of the rows' positions
// (model.hpp:164-175, 190-207 in one launch; the arithmetic of the separate kernels, so the same bits).
// rows <= kQkvRopeMaxRows.
int qkv_rope_append(const RopeAppendTarget &target, float *q_dev, const float *x_dev, const float *wq_dev,
                    const float *wk_dev, const float *wv_dev, int rows, int dim, int q_dim, int device, void *stream);
bool qkv_rope_append_ok(int rows, int dim, int q_dim, int kv_dim);

} // namespace nvk
