// nvk_dropin/core/attention.hpp -- C++ drop-in for the reference's include/core/attention.hpp.
//
// Put this directory in front of the reference's include path (-I include/nvk_dropin -I <nano-vllm>/include)
// and link libnvk_paged_attn.so: the engine's only call site, LlamaModel::attention
// (include/core/model.hpp:405-415 and :421-430), compiles unchanged and now runs the
// gfx950 HIP kernels.  Same namespace, names, argument lists and argument meaning as
//   Attention::paged_attention     include/core/attention.hpp:82-92
//   Attention::standard_attention  include/core/attention.hpp:16-25
// The reference functions cannot fail; these throw std::runtime_error (the engine's own
// error type, caught in src/main.cpp:66-69,86-89) if the device call does.
#pragma once

#include <stdexcept>
#include <string>

#include "nvk_paged_attn.h"

namespace Attention {

inline void paged_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                            const int *block_table, float *att_scores, int num_tokens, int block_size, int head_dim,
                            int n_heads, int n_kv_heads)
{
    // With a device pool bound (nvk_bind_pool) key_cache/value_cache are not read; otherwise they
    // are the reference's host slabs and the blocks block_table names are staged for this call.
    if (nvk_paged_attention(out, q, key_cache, value_cache, block_table, att_scores, num_tokens, block_size,
                            head_dim, n_heads, n_kv_heads)
        != NVK_OK)
        throw std::runtime_error(std::string("nvk_paged_attention: ") + nvk_last_error());
}

inline void standard_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                               float *att_scores, int pos, int head_dim, int n_heads, int n_kv_heads,
                               int max_seq_len)
{
    if (nvk_standard_attention(out, q, key_cache, value_cache, att_scores, pos, head_dim, n_heads, n_kv_heads,
                               max_seq_len)
        != NVK_OK)
        throw std::runtime_error(std::string("nvk_standard_attention: ") + nvk_last_error());
}

} // namespace Attention
