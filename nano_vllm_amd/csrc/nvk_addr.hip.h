// nvk_addr.hip.h -- device helpers shared by the attention kernels (nvk_kernels.hip.h) and the GEMV's RoPE + append
// epilogue (nvk_linear.hip): paged-cache address arithmetic and the rotation's arithmetic.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace nvk {

// ---- address arithmetic shared by decode, gather and append ---------------
// t / block_size without a divide and without a branch: bs_magic = ceil(2^32 / block_size),
// exact for t < 2^32 / block_size (checked on the host); block_size 1 has magic 0 (2^32 does not fit).
__device__ __forceinline__ int token_block(int tok, unsigned bs_magic)
{
    const unsigned whole = bs_magic ? 0u : 0xFFFFFFFFu; // uniform; keeps the lane code branch-free
    return (int)(__umulhi((unsigned)tok, bs_magic) + ((unsigned)tok & whole));
}
__device__ __forceinline__ int token_offset(int tok, int blk, int block_size) { return tok - blk * block_size; }
// float index of element 0 of (physical block, offset, kv head) inside a layer slab
__device__ __forceinline__ int64_t row_float_index(int physical_block, int block_offset, int kv_head,
                                                   int block_size, int n_kv_heads, int head_dim)
{
    return (((int64_t)physical_block * block_size + block_offset) * n_kv_heads + kv_head) * (int64_t)head_dim;
}
// A table entry the reference would treat as undefined behaviour must not fault the GPU.
__device__ __forceinline__ int clamp_block(int id, int num_blocks)
{
    return (int)min((unsigned)id, (unsigned)(num_blocks - 1));
}

// The rotation of Ops::apply_rope (ops/positional.hpp:26-27, :35-36): four products and two sums, each rounded on
// its own (hipcc contracts a*b - c*d into an fma by default, and __fmul_rn/__fsub_rn are plain operators in this
// ROCm: the pragma is what keeps them separate).  cs = (cos, sin).
__device__ __forceinline__ float2 rope_rotate(float2 v, float2 cs)
{
#pragma clang fp contract(off)
    const float xc = v.x * cs.x, ys = v.y * cs.y, xs = v.x * cs.y, yc = v.y * cs.x;
    return make_float2(xc - ys, xs + yc);
}

} // namespace nvk
