// nvk_kernels.hip.h -- gfx950 (CDNA4, wave64) device code for paged-attention decode.
//
// What the reference computes per sequence (include/core/attention.hpp:82-159):
//   out[h,:] = sum_t softmax_t(q[h].K[t,kv(h)] / sqrt(hd)) * V[t,kv(h)],  kv(h) = h / (n_heads/n_kv_heads)
// with K/V rows addressed through the block table (:109-115, :145-151):
//   row(t) = block_table[t / block_size] * block_size + t % block_size
//   &K[t, kvh, 0] = key_cache + (row(t) * n_kv_heads + kvh) * head_dim
//
// How it is mapped to the machine:
//   * work item = (sequence, token chunk).  nvk_plan_kernel turns the batch's
//     context lengths into a list of items, longest first (all full chunks,
//     then the ragged last chunks by decreasing length), so a ragged batch
//     drains evenly over the 256 CUs; one workgroup per item.
//   * wave w of a workgroup owns kv-head (group*waves + w) and its QPK
//     q-heads, so the waves of a workgroup together stream whole, contiguous
//     token rows of the pool.
//   * inside a wave the 64 lanes are 4 token slots x 16 lanes; a slot's 16
//     lanes cover one (token, kv-head) row with VPL float4 each (lane r owns
//     dims 64*j + 4*r .. +3).  One wave load instruction therefore fetches
//     four tokens x 256 contiguous bytes = 8 full 128-B lines, every byte used.
//   * K/V go straight from HBM to VGPRs with the non-temporal hint (read once,
//     no reuse, so no LDS round trip), kStepTokens tokens per step, double
//     buffered: 8-16 KiB in flight per wave.
//   * QK^T: 4*VPL FMAs per (token, q-head) then a 16-lane all-reduce done with
//     four DPP adds (quad_perm, quad_perm, row_half_mirror, row_mirror); no
//     LDS, no cross-row traffic in the loop.
//   * online softmax state (m, l) and the PV accumulator live per token slot
//     in registers, in the exp2 domain (q is pre-scaled by log2(e)/sqrt(hd)).
//   * the chunk's slice of the block table is staged once per workgroup in LDS.
//   * at the end the four slots are merged with two xor-shuffles, and either
//     the normalised row is written (sequence with one chunk) or an
//     unnormalised partial (m, l, o) goes to the split-K workspace, merged by
//     nvk_merge_kernel (flash-decoding).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// ---- build-time tuning knobs (defaults are the measured best) ----
#ifndef NVK_STEP_TOKENS
#define NVK_STEP_TOKENS 8 // tokens per wave per pipeline step: 4 token slots x (NVK_STEP_TOKENS/4) each
#endif
#ifndef NVK_MIN_WAVES_PER_EU
#define NVK_MIN_WAVES_PER_EU 1 // __launch_bounds__ second argument (caps VGPRs: 2 -> 256, 3 -> 168, 4 -> 128)
#endif
#ifndef NVK_NT_LOADS
#define NVK_NT_LOADS 1 // 1: K/V loads carry the non-temporal hint (read-once stream)
#endif
#ifndef NVK_MFMA_LOCKSTEP
#define NVK_MFMA_LOCKSTEP 1 // 1: the same in nvk_decode_mfma_kernel where a workgroup's waves are 8 kv-heads; 2: token parts too; 0 for A/B
#endif
#ifndef NVK_LOCKSTEP
#define NVK_LOCKSTEP 1 // 1: the waves of a workgroup are kept on the same token rows (see decode_body); 0 for A/B
#endif

#include "nvk_addr.hip.h" // token_block, token_offset, row_float_index, clamp_block, rope_rotate

namespace nvk {

// A pipeline step is kStepTokens tokens per wave: 4 token slots x kTokPerLane tokens each.
static_assert(NVK_STEP_TOKENS % 4 == 0 && NVK_STEP_TOKENS >= 4, "a step is whole token quads");
#ifndef NVK_TPL_VPL1
#define NVK_TPL_VPL1 4 // tokens per lane per step for rows of <= 64 floats (twice the others: same bytes in flight)
#endif
// (one or two q-heads over rows of <= 64 floats leave registers for twice that again: 32-token steps, +7 %)
constexpr int tok_per_lane(int vpl, int qpk)
{
    return vpl == 1 ? (qpk <= 2 ? 2 * NVK_TPL_VPL1 : NVK_TPL_VPL1) : NVK_STEP_TOKENS / 4;
}
constexpr int step_tokens(int vpl, int qpk) { return 4 * tok_per_lane(vpl, qpk); }
// Waves per workgroup: 8 (two per SIMD, <= 256 VGPRs each) where q, the accumulators and two tiles fit 256
// registers; 4 (one per SIMD, the whole register file: values the VALU is not working on sit in AGPRs) where
// they do not -- 256-float rows, and 128-float rows with 7 or 8 q-heads per wave (6 when head_dim < 128 adds
// the load predicates).  Cheaper than reading K/V once per q-head sub-group, and no scratch.
constexpr int wg_waves(int vpl, int qpk, bool full) { return vpl >= 4 || (vpl == 2 && qpk >= (full ? 7 : 6)) ? 4 : 8; }
constexpr float kNegBig     = -1.0e30f; // running-max seed: finite, so exp2(m - m) never sees inf - inf
// The reference seeds its max with -1e10 (attention.hpp:126); in the exp2
// domain that floor is -1e10 * log2(e).
constexpr float kRefMaxFloorLog2 = -1.0e10f * 1.44269504088896340736f;
constexpr int   kPlanThreads     = 1024;
constexpr int   kPlanBuckets     = 1024;

// One unit of work for a workgroup: tokens [t0, t1) of sequence `seq`.
// w = chunk index | (chunks of the sequence << 16).
struct __attribute__((aligned(16))) WorkItem
{
    int seq, w, t0, t1;
};

struct DecodeParams
{
    const float    *k_layer; // one layer of the pool: [num_blocks, block_size, n_kv_heads, head_dim]
    const float    *v_layer;
    const float    *q;            // [num_seqs, n_heads, head_dim]
    const int      *block_tables; // [num_seqs, max_blocks]
    const int      *table_rows;   // null, or [num_seqs]: row s of q/out reads table row table_rows[s]
                                  // (chunked prefill: the query tokens of one sequence share its table)
    const WorkItem *items;        // [*n_items], written by nvk_plan_kernel
    const int      *n_items;
    float          *out;   // [num_seqs, n_heads, head_dim]
    float          *ws_o;  // [num_seqs, max_chunks, n_heads, head_dim]  unnormalised partial outputs
    float          *ws_ml; // [num_seqs, max_chunks, n_heads, 2]         (running max, running sum)
    int             max_blocks;
    int             num_blocks;
    int             block_size;
    unsigned        bs_magic; // ceil(2^32 / block_size); 0 for block_size 1 (see token_block)
    int             n_kv_heads;
    int             n_heads;
    int             kv_mul; // n_heads / n_kv_heads
    int             head_dim;
    int             max_chunks;
    int             kv_waves;  // kv-heads (= waves along the kv axis) per workgroup; the workgroup's other
                               // blockDim/64/kv_waves waves split the chunk's tokens among themselves
    int             merge_off; // int offset of the token-split merge area behind the table slice in LDS
    int             items_padded; // item slots per (kv-head group, q sub-group): the item bound, a multiple of 8 when qsplit > 1
    float           q_scale;   // log2(e) / sqrt(head_dim)
    // Direct indexing (no plan): item i is chunk i % max_chunks of sequence i / max_chunks, its length read from
    // ctx_lens on the spot (null: every sequence holds uniform_ctx tokens).  For batches whose lengths are all
    // equal the plan's order IS this order, so nvk_plan_kernel has nothing to add.
    int             direct;
    int             num_seqs;
    int             chunk_tokens;
    int             uniform_ctx;
    const int      *ctx_lens;
    // Split-K combine inside this kernel: one arrival counter per (sequence, kv-head group x q sub-group), zero
    // between launches; the workgroup whose arrival completes a sequence's chunks merges them (null: the partials
    // are left for nvk_merge_kernel / nvk_merge_wide_kernel).
    int            *arrivals;
    // Uniform batches: workgroup i takes item (i % 8) * (n / 8) + i / 8 instead of item i, so that the workgroups of
    // one XCD (ids go round-robin over the 8 XCDs) stream one contiguous eighth of the batch -- see xcd_range_item.
    // 2 or 4: the XCDs share ranges in pairs or fours (xcd_group_item).
    int             xcd_ranges;
    // The even XCDs of an MI355X stream 3-6 % slower than the odd ones (profiles/r03_xcd_balance.log), and a workgroup
    // never changes XCD.  Not 0 (only with xcd_ranges = 2, an even number of full chunks per sequence, where a sequence's
    // even chunks run on even XCDs and its odd chunks on the odd XCD beside it): every even chunk hands its last
    // chunk_skew tokens to the odd chunk that follows.
    int             chunk_skew;
#ifdef NVK_STAMPS // diagnostic build: sixteen 100 MHz timestamps per workgroup (tools/wg_timeline.py)
    unsigned long long *stamps;
#endif
};
#ifdef NVK_STAMPS
#define NVK_STAMP(i)                                                                                                   \
    if (p.stamps && threadIdx.x == 0)                                                                                  \
        p.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
#define NVK_STAMP_WAVE()                                                                                               \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                                           \
        p.stamps[(size_t)blockIdx.x * 16 + 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
#else
#define NVK_STAMP_WAVE()
#define NVK_STAMP(i)
#endif

// ---- wave-level helpers ---------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_move(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
// Sum over the 16 lanes of a DPP row; every lane of the row gets the same bits.
__device__ __forceinline__ float row16_allsum(float x)
{
    x += dpp_move<0xB1>(x);  // quad_perm [1,0,3,2]
    x += dpp_move<0x4E>(x);  // quad_perm [2,3,0,1]
    x += dpp_move<0x141>(x); // row_half_mirror
    x += dpp_move<0x140>(x); // row_mirror
    return x;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b, float acc)
{
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

template <int VPL, int TPL> struct KVTile
{
    float4 k[TPL][VPL];
    float4 v[TPL][VPL];
};

// NT: the non-temporal hint, for row pieces made of whole 128-B lines (a read-once stream: +7-17 %).  A 192-B
// piece (head_dim 48) shares a line with its neighbour kv-head, which another wave reads a little later: with
// the hint that line is gone from L2 by then (measured: G-tiny 5b 0.65 -> 0.74 of peak without it).
template <bool NT> __device__ __forceinline__ float4 load_kv(const float *p)
{
#if NVK_NT_LOADS
    if constexpr (NT) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
#endif
    return *reinterpret_cast<const float4 *>(p);
}

// ---- which item a workgroup of a uniform batch takes ------------------------------------------------------
// Workgroup ids go round-robin over the 8 XCDs.  With item = id, every XCD walks the whole batch at a stride of 8
// items; behind contiguous block tables that is 8 address streams 8 chunks apart per XCD, and how fast they run
// then depends on where the pool happens to lie: config 4 at 0.75 or 0.86 of the HBM peak, 5b at 0.80 or 0.87,
// by the distance between a layer's K slab and V slab alone (profiles/r02_sweep_pool_skew.log).  Dealing each XCD
// one contiguous eighth of the items instead takes the lottery away (0.84-0.86 and 0.86-0.87 at every distance
// tried, profiles/r02_sweep_item_order.log).  A bijection of [0, n) for any n: XCD x owns n / 8 (+1 for the
// first n % 8) consecutive items.  Only for batches of equal lengths: ragged ones are dealt longest-first by the
// plan, round-robin, which is what balances the XCDs.  Every item is computed as before: same bits.
__device__ __forceinline__ int xcd_range_item(int id, int n, int enabled)
{
    if (!enabled)
        return id;
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return x * q + min(x, r) + k;
}

// The same with the XCDs taken g = 2 or 4 at a time (n a multiple of 8): XCDs (g i .. g i + g - 1) own the i-th
// contiguous g/8 of the items and take turns in it, so XCD x streams every g-th item of its group's range -- with an
// even number of chunks per sequence and g = 2, the even chunks run on even XCDs and the odd chunks on the odd XCD
// beside it, which is what DecodeParams::chunk_skew needs.  A bijection of [0, n).
__device__ __forceinline__ int xcd_group_item(int id, int n, int g)
{
    const int x = id & 7, k = id >> 3;
    return (x / g) * ((n >> 3) * g) + g * k + x % g;
}

// ---- the plan kernel: context lengths -> work items, longest first ----------
// One workgroup.  Full chunks (all the same length) come first, sequence-major;
// the ragged last chunks follow by decreasing length (counting sort on the
// length), so the grid's tail is made of the shortest items.  A sequence with no
// context gets one empty item: its workgroup zeroes the output row, as the
// reference does first of all (attention.hpp:98).  Which of two equally long
// partial chunks comes first is arbitrary and does not affect any result: every
// item is computed independently and merged per sequence in chunk order.
// chunk_skew (long chunks only, see DecodeParams::chunk_skew): item = workgroup id here, so the items at even
// positions run on even XCDs -- the slower ones.  Where two neighbours in the list (2j, 2j + 1) are consecutive full
// chunks of one sequence, the first hands its last chunk_skew tokens to the second.
__host__ __device__ __forceinline__ int2 skewed_chunk(int pos, int c, int nfull, int chunk_tokens, int chunk_skew)
{
    int2 t = make_int2(c * chunk_tokens, (c + 1) * chunk_tokens);
    if (!(pos & 1) && c + 1 < nfull)
        t.y -= chunk_skew;
    else if ((pos & 1) && c >= 1)
        t.x -= chunk_skew;
    return t;
}

// Inclusive scan of one int per thread over the whole workgroup (kPlanThreads = 16 waves):
// shuffle scan inside each wave, wave totals through LDS, two barriers in all.
__device__ __forceinline__ int block_inclusive_scan(int v, int *s_wave_totals)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(v, d);
        if (lane >= d)
            v += up;
    }
    __syncthreads(); // previous readers of s_wave_totals are done
    if (lane == 63)
        s_wave_totals[wave] = v;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < kPlanThreads / 64; w++)
        before += w < wave ? s_wave_totals[w] : 0;
    return v + before;
}

__global__ void __launch_bounds__(kPlanThreads)
    nvk_plan_kernel(const int *__restrict__ ctx_lens, int num_seqs, int chunk_tokens, int max_chunks,
                    WorkItem *__restrict__ items, int *__restrict__ n_items, int *__restrict__ seq_chunks, int chunk_skew)
{
    static_assert(kPlanBuckets == kPlanThreads, "one bucket per thread in the bucket scan");
    __shared__ int s_totals[kPlanThreads / 64];
    __shared__ int s_hist[kPlanBuckets];
    __shared__ int s_start[kPlanBuckets];

    const int tid = threadIdx.x;
    s_hist[tid]   = 0;
    __syncthreads();

    auto bucket_of = [&](int rem) { // rem in [0, chunk_tokens): longer remainders get higher buckets
        return (int)(((int64_t)rem * kPlanBuckets) / chunk_tokens);
    };
    const int ctx_cap = max_chunks * chunk_tokens;

    // pass 1: full chunks, sequence-major; histogram of the ragged remainders
    int n_full = 0;
    for (int base = 0; base < num_seqs; base += kPlanThreads) {
        const int s     = base + tid;
        const int ctx   = s < num_seqs ? min(max(ctx_lens[s], 0), ctx_cap) : 0;
        const int nfull = ctx / chunk_tokens;
        const int rem   = ctx - nfull * chunk_tokens;
        const int nch   = nfull + (rem > 0 ? 1 : 0);
        const int incl  = block_inclusive_scan(nfull, s_totals);
        const int first = n_full + incl - nfull;
        for (int c = 0; c < nfull; c++) {
            int2 t = skewed_chunk(first + c, c, nfull, chunk_tokens, chunk_skew);
            items[first + c] = WorkItem{s, c | (nch << 16), t.x, t.y};
        }
        if (s < num_seqs) {
            seq_chunks[s] = nch;
            if (rem > 0 || ctx == 0)
                atomicAdd(&s_hist[bucket_of(rem)], 1);
        }
#pragma unroll
        for (int w = 0; w < kPlanThreads / 64; w++)
            n_full += s_totals[w]; // every thread keeps the running total itself
    }
    __syncthreads(); // histogram complete

    // descending exclusive scan over the buckets: s_start[b] = items in buckets above b
    const int b_mine = kPlanBuckets - 1 - tid;
    const int h_mine = s_hist[b_mine];
    const int incl   = block_inclusive_scan(h_mine, s_totals);
    s_start[b_mine]  = incl - h_mine;
    int n_part       = 0;
#pragma unroll
    for (int w = 0; w < kPlanThreads / 64; w++)
        n_part += s_totals[w];
    __syncthreads(); // everyone has read its s_hist entry
    s_hist[tid] = 0; // reused as per-bucket cursors
    __syncthreads();

    // pass 2: scatter the ragged last chunks (and the empty sequences) behind the full ones
    for (int s = tid; s < num_seqs; s += kPlanThreads) {
        const int ctx   = min(max(ctx_lens[s], 0), ctx_cap);
        const int nfull = ctx / chunk_tokens;
        const int rem   = ctx - nfull * chunk_tokens;
        if (rem > 0 || ctx == 0) {
            const int b   = bucket_of(rem);
            const int pos = s_start[b] + atomicAdd(&s_hist[b], 1);
            const int nch = nfull + (rem > 0 ? 1 : 0);
            items[n_full + pos] = WorkItem{s, nfull | (nch << 16), nfull * chunk_tokens, ctx};
        }
    }
    if (tid == 0)
        *n_items = n_full + n_part;
}

// ---- split-K combine of one (row, q-head, float4) --------------------------------
// Log-sum-exp merge of a row's chunk partials in chunk order, then the reference's normalisation
// (attention.hpp:126-140 with its -1e10 floor on the maximum).  The ONE definition of the combine's arithmetic:
// nvk_merge_kernel runs it a thread per output float4, nvk_decode_kernel runs it in the workgroup that completes
// a sequence, and the two give the same bits.  Loads go four chunks ahead of the (serial) sums.
__device__ __forceinline__ void merge_chunks(const float *ws_o, const float *ws_ml, float *out, int64_t row, int head,
                                             int vec, int nch, int n_heads, int head_dim, int max_chunks)
{
    const float  *ml = ws_ml + (row * max_chunks * n_heads + head) * 2;
    const float  *po = ws_o + (row * max_chunks * n_heads + head) * head_dim + 4 * vec;
    const int64_t ml_step = (int64_t)n_heads * 2, o_step = (int64_t)n_heads * head_dim;

    // the first four chunks' partials are requested before anything waits on them (most launches have no more than
    // four chunks per sequence: one load latency for the whole combine instead of one for the maxima and one for
    // the sums); same arithmetic in the same order as a plain two-pass walk
    float2 s0[4];
    float4 a0[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        s0[u] = make_float2(kRefMaxFloorLog2, 0.f);
        a0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < nch) {
            s0[u] = *reinterpret_cast<const float2 *>(ml + u * ml_step);
            a0[u] = *reinterpret_cast<const float4 *>(po + u * o_step);
        }
    }
    float top = kRefMaxFloorLog2; // the reference's -1e10 floor, see nvk_decode_kernel
#pragma unroll
    for (int u = 0; u < 4; u++)
        top = fmaxf(top, s0[u].x); // (slots past nch hold the floor itself)
    for (int c = 4; c < nch; c++)
        top = fmaxf(top, ml[c * ml_step]);
    float  denom = 0.f;
    float4 acc   = make_float4(0.f, 0.f, 0.f, 0.f);
    auto   add   = [&](const float2 &s, const float4 &a) {
        const float f = fast_exp2(s.x - top);
        denom         = fmaf(s.y, f, denom);
        acc.x         = fmaf(a.x, f, acc.x);
        acc.y         = fmaf(a.y, f, acc.y);
        acc.z         = fmaf(a.z, f, acc.z);
        acc.w         = fmaf(a.w, f, acc.w);
    };
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (u < nch)
            add(s0[u], a0[u]);
    int c = 4;
    for (; c + 4 <= nch; c += 4) {
        float2 s[4];
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s[u] = *reinterpret_cast<const float2 *>(ml + (c + u) * ml_step);
            a[u] = *reinterpret_cast<const float4 *>(po + (c + u) * o_step);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            add(s[u], a[u]);
    }
    for (; c < nch; c++)
        add(*reinterpret_cast<const float2 *>(ml + c * ml_step), *reinterpret_cast<const float4 *>(po + c * o_step));
    const float inv = 1.0f / denom;
    *reinterpret_cast<float4 *>(out + (row * n_heads + head) * head_dim + 4 * vec) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// ---- split-K combine by the workgroup that arrives last (cdna_hip_programming.md, Guideline 16) ----
// The ordering below is spelled in gfx950's own terms: partials stored with the sc1 bit (aux 16 = write-through to
// memory on gfx940+; the same bit means something else on gfx90a), drained with s_waitcnt, a relaxed agent-scope
// arrival, one acquire fence on the last arriver.  On another target it would silently lose its meaning:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "nvk_kernels.hip.h: the in-kernel split-K combine is written for gfx950 (sc1 write-through stores); build with --offload-arch=gfx950"
#endif
// Called by EVERY wave of a workgroup whose chunk's partial has just been stored write-through (sc1).  Every
// storing wave drains its stores, the workgroup meets, ONE lane adds the arrival (agent scope).  Whoever draws
// nch - 1 knows every chunk of the sequence has been written through: one agent-scope acquire (drops this CU's
// stale lines), its wait, a barrier, then plain loads.  Nobody waits for anybody: no spin, no residency
// assumption, any placement of the chunks over XCDs.  qpk = q-heads per kv-head wave of this launch.
__device__ __forceinline__ void combine_if_last(const DecodeParams &p, int *s_flag, int seq, int nch, int kvg, int zq,
                                                int qsplit, int qpk, int kpw = 1)
{
    NVK_STAMP(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    NVK_STAMP(4)
    const int qgroups = gridDim.x / p.items_padded; // (kv-head group, q sub-group) pairs = arrival counters per sequence
    if (threadIdx.x == 0) {
        int *cnt       = p.arrivals + (int64_t)seq * qgroups + (kvg * qsplit + zq);
        const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == nch - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // zero again for the next launch
        }
        s_flag[0] = last; // (the table slice this word belonged to is dead: every wave is past its loop)
    }
    __syncthreads();
    NVK_STAMP(5)
    if (!s_flag[0])
        return;
    const int hd = p.head_dim, vec_row = hd >> 2;
    const int kv_wg  = p.kv_waves * kpw;      // kv-heads of this workgroup's group
    const int n_mine = kv_wg * qpk * vec_row; // (q-head, float4) pairs of this workgroup's heads
    for (int idx = threadIdx.x; idx < n_mine; idx += blockDim.x) {
        const int hh = idx / vec_row, vec = idx - hh * vec_row;
        const int w = hh / qpk, kvh_w = kvg * kv_wg + w;
        if (kvh_w < p.n_kv_heads)
            merge_chunks(p.ws_o, p.ws_ml, p.out, seq, kvh_w * p.kv_mul + zq * qpk + (hh - w * qpk), vec, nch, p.n_heads,
                         hd, p.max_chunks);
    }
    NVK_STAMP(6)
}

// ---- the decode kernel ----------------------------------------------------
// VPL   float4 per lane per row (head_dim <= 64*VPL)
// QPK   q-heads handled per kv-head by one wave (divides kv_mul; blockIdx.y also walks kv_mul/QPK)
// FULL  head_dim == 64*VPL, so no lane is idle and no load needs a predicate
// KPW   kv-heads per wave.  2 only for rows that are an odd number of 64-B half-lines (head_dim 16 / 48: a 192-B row
//       shares a 128-B line with its neighbour kv-head): the wave's four 16-lane rows are then 2 tokens x 2 ADJACENT
//       kv-heads, so one load instruction covers 2 * head_dim * 4 contiguous bytes = whole lines per token, every line
//       is requested once (by one wave), and the loads can carry the read-once hint again.
// grid: 1-D over (work item, kv-head group x q sub-group); workgroups past *n_items leave
// Waves: wave w owns kv-head (w % kv_waves) of the group; when the pool has fewer than 8 kv-heads
// (a head-sharded rank, small models; fewer than 4 for 256-float rows) the remaining waves split the
// chunk's tokens ("token-split"):
// wave w takes sub-range w / kv_waves, and the sub-ranges are merged through LDS at the end, so a
// workgroup keeps 8 waves of loads in flight whatever the head count.
// LOCK  the workgroup holds EVERY kv-head of its token rows (one kv-head group): its waves are kept in step, see below
// WIDE  16-token steps instead of 8 (whole 128-float rows, up to 4 q-heads): a step of the workgroup is then one whole
//       64-KiB block of K and one of V at block_size 16.  For long chunks: config 3 0.84 -> 0.86, config 4 +2 %.
template <int VPL, int QPK, bool FULL, int KPW, bool LOCK, bool WIDE = false>
__device__ __forceinline__ void decode_body(const DecodeParams &p)
{
    static_assert(!WIDE || (VPL == 2 && QPK <= 4 && FULL && KPW == 1), "16-token steps: registers were counted for these");
    static_assert(KPW == 1 || (KPW == 2 && ((VPL == 1 && !FULL) || (VPL == 2 && FULL && QPK <= 4))),
                  "kv-head pairs: short odd rows, or whole 128-float rows in 4-wave workgroups (NVK_DECODE_ROW_PAIRS)");
    constexpr int  kTokPerLane = WIDE ? 4 : tok_per_lane(VPL, QPK); // tokens each lane holds per step
    constexpr int  kSlots      = 4 / KPW;                      // token slots of a wave (16-lane rows per kv-head)
    constexpr int  kStepTokens = 4 * kTokPerLane / KPW;        // tokens per wave per pipeline step
    constexpr bool kStream     = FULL || KPW == 2;             // row pieces are whole lines: read-once hint
    using Tile                = KVTile<VPL, kTokPerLane>;
    extern __shared__ __attribute__((aligned(16))) int s_tab[]; // this chunk's block-table slice (+ merge area)

    // 1-D grid: kv-head group outermost, then (item, q sub-group).  The q sub-groups of an item (only when a GQA
    // group has more q-heads than one wave carries) read the very same K/V bytes: ids are dealt in blocks of
    // 8 * qsplit so that id and id + 8 are the same item, and -- workgroup ids going round-robin over the 8 XCDs --
    // they run on one XCD at about the same time and meet in its L2.
    NVK_STAMP(0)
    const int qsplit = p.kv_mul / QPK;
    const int inner  = p.items_padded * qsplit; // workgroups per kv-head group
    const int kvg    = blockIdx.x / inner;
    int       item_i = blockIdx.x - kvg * inner, zq = 0;
    if (qsplit > 1) {
        const int span = 8 * qsplit, rem = item_i % span;
        item_i         = item_i / span * 8 + rem % 8;
        zq             = rem / 8;
    }
    int seq, chunk, nch = 0, t0, t1 = 0, ctx_raw = 0; // nch: chunks of this sequence; 1 -> this workgroup writes `out` itself
    if (p.direct) {
        if (item_i >= p.num_seqs * p.max_chunks)
            return;
        item_i  = p.xcd_ranges > 1 ? xcd_group_item(item_i, p.num_seqs * p.max_chunks, p.xcd_ranges)
                                   : xcd_range_item(item_i, p.num_seqs * p.max_chunks, p.xcd_ranges);
        seq     = item_i / p.max_chunks;
        chunk   = item_i - seq * p.max_chunks;
        t0      = chunk * p.chunk_tokens - (chunk & 1) * p.chunk_skew;
        ctx_raw = p.ctx_lens ? p.ctx_lens[seq] : p.uniform_ctx; // used only after the q and table loads are on their way
    }
    else {
        if (item_i >= *p.n_items)
            return;
        const WorkItem item = p.items[item_i];
        seq = item.seq, chunk = item.w & 0xFFFF, nch = item.w >> 16, t0 = item.t0, t1 = item.t1;
    }
    // wave-uniform by construction (functions of blockIdx and of scalar loads); say so, for the buffer
    // descriptors and the branches below
    seq = __builtin_amdgcn_readfirstlane(seq), chunk = __builtin_amdgcn_readfirstlane(chunk);
    t0  = __builtin_amdgcn_readfirstlane(t0);

    const int lane   = threadIdx.x & 63;
    const int wave   = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int slot   = (lane >> 4) / KPW; // which token of a step's quad (pair)
    const int hsub   = (lane >> 4) % KPW; // which kv-head of the wave's pair
    const int r      = lane & 15;         // which 16 B of a 256-B row piece

    // waves enumerate the kv-heads of the group (and, with fewer kv-heads than waves, the token parts)
    const int  parts  = nwaves / p.kv_waves; // token-split factor (1 when the group has 8 kv-heads)
    const int  part   = wave / p.kv_waves;
    const int  kvh    = (kvg * p.kv_waves + wave % p.kv_waves) * KPW + hsub;
    const bool live   = kvh < p.n_kv_heads; // (KPW == 2: per 16-lane row; an odd head count leaves the last row dead)
    const int  qh0    = kvh * p.kv_mul + zq * QPK; // first q-head of this wave

    const int hd = p.head_dim;
    bool      act[VPL];
#pragma unroll
    for (int j = 0; j < VPL; j++)
        act[j] = FULL || (64 * j + 4 * r < hd);

    // q for this wave's q-heads: issued before the table staging so both latencies overlap
    float4 qv[QPK][VPL];
#pragma unroll
    for (int h = 0; h < QPK; h++)
#pragma unroll
        for (int j = 0; j < VPL; j++) {
            qv[h][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live && act[j])
                qv[h][j] = *reinterpret_cast<const float4 *>(p.q + ((int64_t)seq * p.n_heads + qh0 + h) * hd
                                                             + 64 * j + 4 * r);
        }

    // the chunk's slice of the block table -> LDS.  With a plan the slice ends at the chunk's last token; without
    // one the length is still on its way, so the whole chunk's range is staged (entries past a sequence's end are
    // padding of its row: readable, clamped, never used).
    const int blk0 = token_block(t0, p.bs_magic);
    const int chunk_len = p.chunk_tokens + ((chunk & 1) ? p.chunk_skew : -p.chunk_skew); // (direct launches)
    const int blk_last = p.direct ? min(token_block(t0 + chunk_len - 1, p.bs_magic), p.max_blocks - 1)
                                  : (t1 > t0 ? token_block(t1 - 1, p.bs_magic) : blk0 - 1);
#ifdef NVK_NO_TABLE_ROWS // A/B build: the decode kernel without the prefill indirection
    const int trow = seq;
#else
    const int trow = p.table_rows ? p.table_rows[seq] : seq;
#endif
    for (int i = threadIdx.x; i <= blk_last - blk0; i += blockDim.x)
        s_tab[i] = clamp_block(p.block_tables[(int64_t)trow * p.max_blocks + blk0 + i], p.num_blocks);
    if (p.direct) {
        const int ctx = min(max(ctx_raw, 0), p.max_chunks * p.chunk_tokens);
        nch           = (ctx + p.chunk_tokens - 1) / p.chunk_tokens;
        t1            = min(ctx, t0 + chunk_len);
        if (chunk >= max(nch, 1))
            return; // the sequence ends before this chunk (an empty sequence keeps chunk 0: it zeroes the row)
    }
    nch = __builtin_amdgcn_readfirstlane(nch), t1 = __builtin_amdgcn_readfirstlane(t1);
    __syncthreads();
    NVK_STAMP(1)

    if (t1 <= t0) {
        // a sequence with no context: the reference zeroes `out` and does nothing else (attention.hpp:98)
        if (live && slot == 0 && part == 0) {
#pragma unroll
            for (int h = 0; h < QPK; h++)
#pragma unroll
                for (int j = 0; j < VPL; j++)
                    if (act[j])
                        *reinterpret_cast<float4 *>(p.out + ((int64_t)seq * p.n_heads + qh0 + h) * hd + 64 * j
                                                    + 4 * r) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    // split-K combine in this kernel: every wave of the workgroup stays to the end (the arrival protocol has
    // two barriers); otherwise waves without a kv-head leave now (they exist only when parts == 1: no barrier follows)
    const bool fused = p.arrivals != nullptr && nch > 1;
    if (KPW == 1 && !live && !fused)
        return;

    // this wave's share of the chunk: whole steps
    // (steps dealt round-robin over the token-split parts: at any moment the workgroup's waves stream neighbouring
    // rows, the access pattern of a workgroup whose waves are the kv-heads of one row.  Of the two waves a SIMD
    // hosts the one launched first streams 1.1-1.35 x faster -- profiles/r02_wave_timeline.log -- but dealing it
    // more steps, 5:4 or 4:3, measured no gain: whoever finishes first leaves its bandwidth to the other.)
    const int w1 = t1, w0 = t0 + part * kStepTokens, stride = parts * kStepTokens;
    int       nsteps = (t1 - t0 + kStepTokens - 1) / kStepTokens;
    nsteps           = nsteps > part ? (nsteps - part + parts - 1) / parts : 0;
    const bool has_tokens = live && nsteps > 0;

    // pre-scale q into the exp2 domain
#pragma unroll
    for (int h = 0; h < QPK; h++)
#pragma unroll
        for (int j = 0; j < VPL; j++) {
            qv[h][j].x *= p.q_scale, qv[h][j].y *= p.q_scale, qv[h][j].z *= p.q_scale, qv[h][j].w *= p.q_scale;
        }

    float  m[QPK], l[QPK];
    float4 o[QPK][VPL];
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        m[h] = kNegBig;
        l[h] = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; j++)
            o[h][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int lane_col = kvh * hd + 4 * r; // float offset of this lane inside a token row
    const int row_len  = p.n_kv_heads * hd;

    auto load_tile = [&](Tile &T, int ts) {
        int64_t at[kTokPerLane];
#pragma unroll
        for (int i = 0; i < kTokPerLane; i++) {
            // lanes past the chunk's end (last step only) re-read its last row; consume_tile masks their scores
            const int tok = min(ts + kSlots * i + slot, w1 - 1);
            const int blk = token_block(tok, p.bs_magic);
            const int off = token_offset(tok, blk, p.block_size);
            const int pb  = s_tab[blk - blk0];
            at[i]         = ((int64_t)pb * p.block_size + off) * row_len + lane_col;
        }
#pragma unroll
        for (int i = 0; i < kTokPerLane; i++)
#pragma unroll
            for (int j = 0; j < VPL; j++) {
                T.k[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (act[j])
                    T.k[i][j] = load_kv<kStream>(p.k_layer + at[i] + 64 * j);
            }
#pragma unroll
        for (int i = 0; i < kTokPerLane; i++)
#pragma unroll
            for (int j = 0; j < VPL; j++) {
                T.v[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (act[j])
                    T.v[i][j] = load_kv<kStream>(p.v_layer + at[i] + 64 * j);
            }
    };

    // TAIL = false: every token of the step is inside the chunk, no masking needed
    auto consume_tile = [&](const Tile &T, int ts, auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        float          s[kTokPerLane][QPK];
#pragma unroll
        for (int i = 0; i < kTokPerLane; i++) {
            const bool in_range = !TAIL || (ts + kSlots * i + slot < w1);
#pragma unroll
            for (int h = 0; h < QPK; h++) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < VPL; j++)
                    acc = dot4(qv[h][j], T.k[i][j], acc);
                acc     = row16_allsum(acc);
                s[i][h] = in_range ? acc : -INFINITY;
            }
        }
#pragma unroll
        for (int h = 0; h < QPK; h++) {
            float mn = m[h];
#pragma unroll
            for (int i = 0; i < kTokPerLane; i++)
                mn = fmaxf(mn, s[i][h]);
            const float alpha = fast_exp2(m[h] - mn);
            float       pw[kTokPerLane];
            float       psum = 0.f;
#pragma unroll
            for (int i = 0; i < kTokPerLane; i++) {
                pw[i] = fast_exp2(s[i][h] - mn);
                psum += pw[i];
            }
            m[h] = mn;
            l[h] = fmaf(l[h], alpha, psum);
#pragma unroll
            for (int j = 0; j < VPL; j++) {
                float4 a = o[h][j];
                a.x *= alpha, a.y *= alpha, a.z *= alpha, a.w *= alpha;
#pragma unroll
                for (int i = 0; i < kTokPerLane; i++) {
                    a.x = fmaf(pw[i], T.v[i][j].x, a.x);
                    a.y = fmaf(pw[i], T.v[i][j].y, a.y);
                    a.z = fmaf(pw[i], T.v[i][j].z, a.z);
                    a.w = fmaf(pw[i], T.v[i][j].w, a.w);
                }
                o[h][j] = a;
            }
        }
    };

    // Software pipeline over steps of kStepTokens tokens: two named tiles, hand-unrolled by two
    // (no register copies, no runtime-indexed arrays).  The steady-state loop prefetches
    // unconditionally, so every path into a consume has the same number of loads in flight and
    // hipcc emits counted vmcnt waits that leave the whole next tile outstanding; the last one
    // or two steps (the only ones that can run past the chunk's end) are peeled off.
    const std::integral_constant<bool, false> body;
    const std::integral_constant<bool, true>  tail;
    // Lockstep: the waves of a workgroup are the kv-heads (and token parts) of the SAME token rows, and the stream
    // runs fastest when they ask for those rows together -- left alone, the older wave of each SIMD runs up to 10 %
    // of the chunk ahead of the younger one (profiles/r02_wave_timeline.log) and the workgroup's requests spread
    // over more DRAM pages.  A bare s_barrier (no wait on the loads in flight) at the top of every double-step keeps
    // them within one: config 2 0.82 -> 0.85 of the HBM peak, 5b 0.87 -> 0.89 (profiles/r02_variants_lockstep.log).
    // Only where the workgroup holds every kv-head of the row (several kv-head groups, e.g. MHA's 32 heads, are
    // different workgroups that no barrier joins: measured 2 % slower there; those launch nvk_decode_free_kernel,
    // this code without the barrier) and every wave is live; the barriers are counted from the steps EVERY wave
    // runs (token parts differ by at most one step).  Not for 256-float rows (one wave per SIMD: nothing drifts;
    // measured 3 % slower), nor for 8 q-heads over whole 64-float rows (the MFMA kernel's shape unless
    // NVK_DECODE_VECTOR_ONLY forces this one: the branch costs the instantiation its last registers).
    constexpr bool kLock = LOCK && NVK_LOCKSTEP && (KPW == 1 || VPL == 2) && VPL < 4 && !(VPL == 1 && QPK == 8 && FULL);
    const int locked_upto = kLock && p.n_kv_heads == p.kv_waves * KPW
                                ? (t1 - t0 + kStepTokens - 1) / kStepTokens / parts // steps of the part with the fewest
                                : 0;
    if (has_tokens) {
        Tile        ta, tb;
        int         k = 0;
        load_tile(ta, w0);
        for (; k + 2 < nsteps; k += 2) {
            if constexpr (kLock)
                if (k + 2 < locked_upto)
                    __builtin_amdgcn_s_barrier();
            load_tile(tb, w0 + (k + 1) * stride);
            consume_tile(ta, w0 + k * stride, body);
            load_tile(ta, w0 + (k + 2) * stride);
            consume_tile(tb, w0 + (k + 1) * stride, body);
        }
        if (nsteps - k == 2) {
            load_tile(tb, w0 + (k + 1) * stride);
            consume_tile(ta, w0 + k * stride, body);
            consume_tile(tb, w0 + (k + 1) * stride, tail);
        }
        else {
            consume_tile(ta, w0 + k * stride, tail);
        }
    }

    NVK_STAMP(2)
    NVK_STAMP_WAVE()
    // merge the token slots of the wave (lanes 16 and 32 apart; with a kv-head pair per wave the rows 16 apart are
    // the other head's, and only the rows 32 apart meet)
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        float mw = m[h];
        if (KPW == 1)
            mw = fmaxf(mw, __shfl_xor(mw, 16));
        mw            = fmaxf(mw, __shfl_xor(mw, 32));
        const float f = fast_exp2(m[h] - mw);
        float       lw = l[h] * f;
        if (KPW == 1)
            lw += __shfl_xor(lw, 16);
        lw += __shfl_xor(lw, 32);
        m[h] = mw;
        l[h] = lw;
#pragma unroll
        for (int j = 0; j < VPL; j++) {
            float4 a = o[h][j];
            a.x *= f, a.y *= f, a.z *= f, a.w *= f;
            if (KPW == 1)
                a.x += __shfl_xor(a.x, 16), a.y += __shfl_xor(a.y, 16), a.z += __shfl_xor(a.z, 16),
                    a.w += __shfl_xor(a.w, 16);
            a.x += __shfl_xor(a.x, 32), a.y += __shfl_xor(a.y, 32), a.z += __shfl_xor(a.z, 32),
                a.w += __shfl_xor(a.w, 32);
            o[h][j] = a;
        }
    }

    // token-split: the workgroup's sub-ranges of one kv-head meet in LDS.  Every part leaves its (m, l, o); after the
    // barrier the merge is dealt over ALL threads of the workgroup, one output float4 each, every part's record
    // loaded at once -- one LDS latency where one wave walking the parts paid one per part and q-head (4/1/128 at
    // config-2 size: 5.5 us of a 28 us kernel, profiles/r02_config2_wg_timeline.log) -- and each thread stores its
    // float4 straight to `out` (one chunk) or to the chunk's partial.  Same sums in the same order as before: part 0
    // first, then the parts by index.
    if (parts > 1) {
        constexpr int kRec  = 4 + VPL * 64; // floats per (wave, q-head): m, l, pad, pad, then o as [VPL][16] float4
        constexpr int kVecs = VPL * 16;     // float4 per (kv-head, q-head)
        constexpr int kMaxParts = 8;
        float        *s_mrg = reinterpret_cast<float *>(s_tab + p.merge_off);
        if (slot == 0) {
#pragma unroll
            for (int h = 0; h < QPK; h++) {
                float *rec = s_mrg + ((wave * KPW + hsub) * QPK + h) * kRec;
                if (r == 0)
                    rec[0] = m[h], rec[1] = l[h];
#pragma unroll
                for (int j = 0; j < VPL; j++)
                    *reinterpret_cast<float4 *>(rec + 4 + (j * 16 + r) * 4) = o[h][j];
            }
        }
        NVK_STAMP(7) // (diagnostic build: this wave's records are in LDS)
        __syncthreads();
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        constexpr int kSc1  = 16; // aux bit 4 of the buffer instructions on gfx950: write-through, see below
        const int     total = p.kv_waves * KPW * QPK * kVecs;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const int vec = idx % kVecs, hh = idx / kVecs; // hh = (kv slot, head of the pair, q-head)
            const int h = hh % QPK, ks = hh / QPK;         // ks = kw * KPW + hsub
            const int kvh_i = kvg * p.kv_waves * KPW + ks;
            if (kvh_i >= p.n_kv_heads || 4 * vec >= hd)
                continue;
            const int    kw = ks / KPW, hs = ks - kw * KPW;
            float        pm[kMaxParts], pl[kMaxParts];
            float4       po[kMaxParts];
#pragma unroll
            for (int q = 0; q < kMaxParts; q++) {
                pm[q] = kNegBig, pl[q] = 0.f, po[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < parts) {
                    const float *rec = s_mrg + ((((q * p.kv_waves + kw) * KPW + hs) * QPK + h)) * kRec;
                    pm[q] = rec[0], pl[q] = rec[1];
                    po[q] = *reinterpret_cast<const float4 *>(rec + 4 + vec * 4);
                }
            }
            float mw = pm[0];
#pragma unroll
            for (int q = 1; q < kMaxParts; q++)
                if (q < parts)
                    mw = fmaxf(mw, pm[q]);
            const float f0 = fast_exp2(pm[0] - mw);
            float       lw = pl[0] * f0;
            float4      a  = make_float4(po[0].x * f0, po[0].y * f0, po[0].z * f0, po[0].w * f0);
#pragma unroll
            for (int q = 1; q < kMaxParts; q++)
                if (q < parts) {
                    const float f = fast_exp2(pm[q] - mw);
                    lw            = fmaf(pl[q], f, lw);
                    a.x = fmaf(po[q].x, f, a.x), a.y = fmaf(po[q].y, f, a.y);
                    a.z = fmaf(po[q].z, f, a.z), a.w = fmaf(po[q].w, f, a.w);
                }
            const int head = kvh_i * p.kv_mul + zq * QPK + h;
            if (nch == 1) {
                const float mf  = fmaxf(mw, kRefMaxFloorLog2); // the reference's -1e10 floor, as below
                const float f   = fast_exp2(mw - mf);
                const float inv = f / (lw * f);
                *reinterpret_cast<float4 *>(p.out + ((int64_t)seq * p.n_heads + head) * hd + 4 * vec) =
                    make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
            }
            else {
                const int64_t slab = (int64_t)seq * p.max_chunks + chunk;
                const __amdgpu_buffer_rsrc_t rs_o =
                    __builtin_amdgcn_make_buffer_rsrc(p.ws_o + slab * p.n_heads * hd, 0, p.n_heads * hd * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rs_o, (head * hd + 4 * vec) * 4, 0,
                                                       kSc1);
                if (vec == 0) {
                    const __amdgpu_buffer_rsrc_t rs_ml =
                        __builtin_amdgcn_make_buffer_rsrc(p.ws_ml + slab * p.n_heads * 2, 0, p.n_heads * 8, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(mw, lw)), rs_ml, head * 8,
                                                          0, kSc1);
                }
            }
        }
        if (!fused)
            return;
        combine_if_last(p, s_tab, seq, nch, kvg, zq, qsplit, QPK, KPW);
        return;
    }

    const bool writer = live && slot == 0 && part == 0; // the lanes that hold the chunk's (m, l, o)

    if (nch == 1) {
        if (!writer)
            return;
#pragma unroll
        for (int h = 0; h < QPK; h++) {
            // reference: max_val = max(-1e10, scores) (attention.hpp:126-130); a no-op unless every score is below it
            const float mf  = fmaxf(m[h], kRefMaxFloorLog2);
            const float f   = fast_exp2(m[h] - mf);
            const float inv = f / (l[h] * f);
#pragma unroll
            for (int j = 0; j < VPL; j++)
                if (act[j])
                    *reinterpret_cast<float4 *>(p.out + ((int64_t)seq * p.n_heads + qh0 + h) * hd + 64 * j + 4 * r) =
                        make_float4(o[h][j].x * inv, o[h][j].y * inv, o[h][j].z * inv, o[h][j].w * inv);
        }
        return;
    }

    // The chunk's unnormalised partial goes to its slab of the workspace, write-through (sc1): the bytes are on
    // their way to memory when the store completes, so the workgroup that combines the sequence -- on whichever
    // XCD it runs -- needs no release fence from this one, only the arrival below.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    constexpr int kSc1 = 16; // aux bit 4 of the buffer instructions on gfx950
    if (writer) {
        const int64_t slab = (int64_t)seq * p.max_chunks + chunk;
        const __amdgpu_buffer_rsrc_t rs_o =
            __builtin_amdgcn_make_buffer_rsrc(p.ws_o + slab * p.n_heads * hd, 0, p.n_heads * hd * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_ml =
            __builtin_amdgcn_make_buffer_rsrc(p.ws_ml + slab * p.n_heads * 2, 0, p.n_heads * 8, 0x00020000);
#pragma unroll
        for (int h = 0; h < QPK; h++) {
#pragma unroll
            for (int j = 0; j < VPL; j++)
                if (act[j])
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[h][j]), rs_o,
                                                           ((qh0 + h) * hd + 64 * j + 4 * r) * 4, 0, kSc1);
            if (r == 0)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(m[h], l[h])), rs_ml,
                                                      (qh0 + h) * 8, 0, kSc1);
        }
    }
    if (!fused)
        return;

    combine_if_last(p, s_tab, seq, nch, kvg, zq, qsplit, QPK, KPW);
}

template <int VPL, int QPK, bool FULL>
__global__ void __launch_bounds__(64 * wg_waves(VPL, QPK, FULL), NVK_MIN_WAVES_PER_EU)
    nvk_decode_kernel(const DecodeParams p)
{
    decode_body<VPL, QPK, FULL, 1, true>(p);
}

// whole 128-float rows in 16-token steps (WIDE), for chunks of 1024 tokens and more that no token parts split
template <int QPK>
__global__ void __launch_bounds__(64 * wg_waves(2, QPK, true), NVK_MIN_WAVES_PER_EU) nvk_decode_wide_kernel(const DecodeParams p)
{
    decode_body<2, QPK, true, 1, true, true>(p);
}

// the same for pools with more kv-heads than a workgroup has waves (several kv-head groups per token row, e.g. MHA):
// the waves run free
template <int VPL, int QPK, bool FULL>
__global__ void __launch_bounds__(64 * wg_waves(VPL, QPK, FULL), NVK_MIN_WAVES_PER_EU)
    nvk_decode_free_kernel(const DecodeParams p)
{
    decode_body<VPL, QPK, FULL, 1, false>(p);
}

// rows of 64 / 192 B (head_dim 16 / 48), an even number of kv-heads: a pair of adjacent kv-heads per wave (KPW = 2)
template <int QPK>
__global__ void __launch_bounds__(64 * wg_waves(1, QPK, false), NVK_MIN_WAVES_PER_EU)
    nvk_decode_pairs_kernel(const DecodeParams p)
{
    decode_body<1, QPK, false, 2, false>(p);
}

// whole 128-float rows, a PAIR of adjacent kv-heads per wave, 4 waves per workgroup (NVK_DECODE_ROW_PAIRS): the
// workgroup still holds every kv-head of its token rows (8 kv-heads), but two of them fit a CU, so a workgroup's
// successor starts beside a running one instead of behind it -- for batches of many short sequences, whose launch
// is generations of workgroups each paying its start and its end with the CU idle
template <int QPK>
__global__ void __launch_bounds__(256, 2) nvk_decode_rowpairs_kernel(const DecodeParams p)
{
    decode_body<2, QPK, true, 2, true>(p);
}

// ---- the decode kernel on fp32 MFMA tiles, for GQA groups of 5..16 q-heads --------------------------------
// With many q-heads per kv-head the vector kernel above stops being a stream: at 8 q-heads over 64-float rows
// it issues ~500 VALU instructions per 16 tokens per wave (8 x the dot products, 8 x the 16-lane DPP sums, the
// exponentials 16 x redundantly), the two waves of a SIMD contend for issue slots and the younger one runs 1.6 x
// slower (profiles/r02_wave_timeline.log): contraction-bound, which is where north_star asks for the matrix
// cores.  Same work items, same block-table staging, same split-K / token-split / combine as nvk_decode_kernel;
// only the inner products change:
//   * a step is a tile of 16 tokens.  S[token][head] = K . Q^T on v_mfma_f32_16x16x4_f32 (exact fp32, an fmaf
//     chain): A = K with the token on lane & 15 and four dims per lane quarter (lane >> 4), B = the wave's Q
//     fragment (head on lane & 15, heads past kv_mul zero), resident in HD / 4 registers.  HD / 4 MFMAs.
//   * the result has the head on lane & 15 and tokens 4 * (lane >> 4) + reg in four registers -- which is exactly
//     the B operand of O^T[dim][head] += V^T[dim][token] . P[token][head] with V in its natural coalesced
//     layout (token 4 * (lane >> 4) + reg, dims 4 * (lane & 15) + c [+ 64]): no transpose, no LDS.  HD / 4 MFMAs.
//   * softmax statistics per head = per lane column: 3 in-lane max + 2 exchanges (lanes 16 and 32 apart), FOUR
//     exponentials per lane per tile; the running sum stays per lane and is joined once at the end.
// K is loaded as 16 token rows x 64-B pieces per instruction (the A operand's shape), V as 4 rows x 256 B.
// Bound: HBM; the MFMA pipe is ~35 % busy (32 or 64 MFMAs of 32 cycles per 16 tokens per wave, two waves per SIMD).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int HD> struct MfmaTile
{
    float4 k[HD / 16];
    float4 v[4][HD / 64];
};

template <int HD> __global__ void __launch_bounds__(512, 1) nvk_decode_mfma_kernel(const DecodeParams p)
{
    constexpr int NM = HD / 16, NV = HD / 64, kStepTokens = 16, kRec = 4 + HD;
    using Tile = MfmaTile<HD>;
    extern __shared__ __attribute__((aligned(16))) int s_tab[];

    NVK_STAMP(0)
    const int kvg    = blockIdx.x / p.items_padded; // one wave carries the whole GQA group: no q sub-groups
    int       item_i = blockIdx.x - kvg * p.items_padded;
    int seq, chunk, nch = 0, t0, t1 = 0, ctx_raw = 0;
    if (p.direct) {
        if (item_i >= p.num_seqs * p.max_chunks)
            return;
        item_i  = p.xcd_ranges > 1 ? xcd_group_item(item_i, p.num_seqs * p.max_chunks, p.xcd_ranges)
                                   : xcd_range_item(item_i, p.num_seqs * p.max_chunks, p.xcd_ranges);
        seq     = item_i / p.max_chunks;
        chunk   = item_i - seq * p.max_chunks;
        t0      = chunk * p.chunk_tokens - (chunk & 1) * p.chunk_skew;
        ctx_raw = p.ctx_lens ? p.ctx_lens[seq] : p.uniform_ctx;
    }
    else {
        if (item_i >= *p.n_items)
            return;
        const WorkItem item = p.items[item_i];
        seq = item.seq, chunk = item.w & 0xFFFF, nch = item.w >> 16, t0 = item.t0, t1 = item.t1;
    }
    seq = __builtin_amdgcn_readfirstlane(seq), chunk = __builtin_amdgcn_readfirstlane(chunk);
    t0  = __builtin_amdgcn_readfirstlane(t0);

    const int  lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int  c16 = lane & 15, q4 = lane >> 4;
    const int  parts = nwaves / p.kv_waves, part = wave / p.kv_waves;
    const int  kvh  = kvg * p.kv_waves + wave % p.kv_waves;
    const bool live = kvh < p.n_kv_heads;
    const int  kv_mul = p.kv_mul;
    const bool head_live = live && c16 < kv_mul; // lanes whose column is a real q-head
    const int64_t head_row = (int64_t)seq * p.n_heads + kvh * kv_mul + c16; // row of q / out

    // Q fragment: Q[head c16][dims 16 m + 4 q4 .. + 3]
    float4 qf[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) {
        qf[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (head_live)
            qf[m] = *reinterpret_cast<const float4 *>(p.q + head_row * HD + 16 * m + 4 * q4);
    }

    const int blk0 = token_block(t0, p.bs_magic);
    const int chunk_len = p.chunk_tokens + ((chunk & 1) ? p.chunk_skew : -p.chunk_skew); // (direct launches)
    const int blk_last = p.direct ? min(token_block(t0 + chunk_len - 1, p.bs_magic), p.max_blocks - 1)
                                  : (t1 > t0 ? token_block(t1 - 1, p.bs_magic) : blk0 - 1);
    const int trow = p.table_rows ? p.table_rows[seq] : seq;
    for (int i = threadIdx.x; i <= blk_last - blk0; i += blockDim.x)
        s_tab[i] = clamp_block(p.block_tables[(int64_t)trow * p.max_blocks + blk0 + i], p.num_blocks);
    if (p.direct) {
        const int ctx = min(max(ctx_raw, 0), p.max_chunks * p.chunk_tokens);
        nch           = (ctx + p.chunk_tokens - 1) / p.chunk_tokens;
        t1            = min(ctx, t0 + chunk_len);
        if (chunk >= max(nch, 1))
            return;
    }
    nch = __builtin_amdgcn_readfirstlane(nch), t1 = __builtin_amdgcn_readfirstlane(t1);
    __syncthreads();
    NVK_STAMP(1)

    if (t1 <= t0) { // no context: the reference zeroes `out` and does nothing else (attention.hpp:98)
        if (head_live && part == 0) {
#pragma unroll
            for (int m = 0; m < NM; m++)
                *reinterpret_cast<float4 *>(p.out + head_row * HD + 16 * m + 4 * q4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const bool fused = p.arrivals != nullptr && nch > 1;
    if (!live && !fused)
        return;

    // (steps dealt round-robin over the token-split parts: at any moment the workgroup's waves stream neighbouring
    // rows, the access pattern of a workgroup whose waves are the kv-heads of one row.  Of the two waves a SIMD
    // hosts the one launched first streams 1.1-1.35 x faster -- profiles/r02_wave_timeline.log -- but dealing it
    // more steps, 5:4 or 4:3, measured no gain: whoever finishes first leaves its bandwidth to the other.)
    const int w1 = t1, w0 = t0 + part * kStepTokens, stride = parts * kStepTokens;
    int       nsteps = (t1 - t0 + kStepTokens - 1) / kStepTokens;
    nsteps           = nsteps > part ? (nsteps - part + parts - 1) / parts : 0;
    const bool has_tokens = live && nsteps > 0;

#pragma unroll
    for (int m = 0; m < NM; m++)
        qf[m].x *= p.q_scale, qf[m].y *= p.q_scale, qf[m].z *= p.q_scale, qf[m].w *= p.q_scale;

    float m_run = kNegBig, l_run = 0.f; // this lane's head: running max (shared by its four lanes), this lane's share of the sum
    f32x4 oacc[NV][4];                  // O[head c16][dims 64 j + 16 q4 + 4 reg + c] = oacc[j][c][reg]
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            oacc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row_len = p.n_kv_heads * HD;
    auto row_at = [&](int tok) -> int64_t {
        const int blk = token_block(tok, p.bs_magic);
        const int off = token_offset(tok, blk, p.block_size);
        return ((int64_t)s_tab[blk - blk0] * p.block_size + off) * row_len + kvh * HD;
    };
    auto load_tile = [&](Tile &T, int ts) {
        // lanes past the chunk's end (last step only) re-read its last row; consume_tile masks their scores
        const int64_t atk = row_at(min(ts + c16, w1 - 1)) + 4 * q4;
        int64_t       atv[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            atv[i] = row_at(min(ts + 4 * q4 + i, w1 - 1)) + 4 * c16;
#ifdef NVK_MFMA_K_AS_V // TIMING ONLY (wrong results): K fetched in V's shape, 4 rows x 256 B per instruction
#pragma unroll
        for (int m = 0; m < NM; m++)
            T.k[m] = load_kv<true>(p.k_layer + atv[m & 3] + 64 * (m >> 2));
#else
#pragma unroll
        for (int m = 0; m < NM; m++)
#ifdef NVK_MFMA_K_NT // A/B: the read-once hint on K too
            T.k[m] = load_kv<true>(p.k_layer + atk + 16 * m);
#else
            // K arrives as 64-B half-lines (the A operand's shape): the instruction for the other half follows at
            // once and should find the line in L1, which the read-once hint would have dropped.  Measured
            // (profiles/r02_variants_mfma_k_hint.log): GQA-8 64 x 2K 0.72 -> 0.78 of peak, 512 x 2K 0.84 -> 0.85;
            // head_dim 64: 0.70 -> 0.72 and 0.856 -> 0.846.  V (whole lines per instruction) keeps the hint.
            T.k[m] = load_kv<false>(p.k_layer + atk + 16 * m);
#endif
#endif
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < NV; j++)
                T.v[i][j] = load_kv<true>(p.v_layer + atv[i] + 64 * j);
    };
    auto consume_tile = [&](const Tile &T, int ts, auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f}; // two chains: the pipe never waits on one
#pragma unroll
        for (int m = 0; m < NM; m += 2) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m].x, qf[m].x, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m + 1].x, qf[m + 1].x, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m].y, qf[m].y, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m + 1].y, qf[m + 1].y, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m].z, qf[m].z, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m + 1].z, qf[m + 1].z, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m].w, qf[m].w, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(T.k[m + 1].w, qf[m + 1].w, s1, 0, 0, 0);
        }
        float s[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            s[r] = s0[r] + s1[r]; // token ts + 4 q4 + r, head c16
            if (TAIL && ts + 4 * q4 + r >= w1)
                s[r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        mx       = fmaxf(mx, __shfl_xor(mx, 16));
        mx       = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn    = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - mn);
        float       pw[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            pw[r] = fast_exp2(s[r] - mn);
        m_run = mn;
        l_run = fmaf(l_run, alpha, (pw[0] + pw[1]) + (pw[2] + pw[3]));
        // the running max settles after the first tiles: skip the rescale whenever no head saw a new maximum
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int j = 0; j < NV; j++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    oacc[j][c] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < NV; j++) {
                oacc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(T.v[r][j].x, pw[r], oacc[j][0], 0, 0, 0);
                oacc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(T.v[r][j].y, pw[r], oacc[j][1], 0, 0, 0);
                oacc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(T.v[r][j].z, pw[r], oacc[j][2], 0, 0, 0);
                oacc[j][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(T.v[r][j].w, pw[r], oacc[j][3], 0, 0, 0);
            }
    };

    const std::integral_constant<bool, false> body;
    const std::integral_constant<bool, true>  tail;
    // Lockstep, as in decode_body: left alone the older wave of each SIMD runs 1.4 x faster than the younger one here
    // (2048-token chunks of 64/8/128: main loops of 434 and 602 us, profiles/r03_mfma_lockstep.log), and once the older
    // four have ended the CU streams with half its loads in flight.  A bare s_barrier at the top of every double-step
    // keeps the eight within one: 64/8/128 at config-2 size 0.745 -> 0.792 of the HBM peak, at 16 x 32K 0.824 -> 0.832.
    // Only where the eight waves are the kv-heads of the same rows: with token parts (1-4 kv-heads) it measured +1.5 % at
    // config-2 size and -1 ... -2 % on long chunks and on 5a (NVK_MFMA_LOCKSTEP=2 locks those too).
    const int locked_upto = NVK_MFMA_LOCKSTEP && p.n_kv_heads == p.kv_waves && (parts == 1 || NVK_MFMA_LOCKSTEP == 2)
                                ? (t1 - t0 + kStepTokens - 1) / kStepTokens / parts // steps of the part with the fewest
                                : 0;
    if (has_tokens) {
        Tile ta, tb;
        int  k = 0;
        load_tile(ta, w0);
        for (; k + 2 < nsteps; k += 2) {
            if (NVK_MFMA_LOCKSTEP && k + 2 < locked_upto)
                __builtin_amdgcn_s_barrier();
            load_tile(tb, w0 + (k + 1) * stride);
            consume_tile(ta, w0 + k * stride, body);
            load_tile(ta, w0 + (k + 2) * stride);
            consume_tile(tb, w0 + (k + 1) * stride, body);
        }
        if (nsteps - k == 2) {
            load_tile(tb, w0 + (k + 1) * stride);
            consume_tile(ta, w0 + k * stride, body);
            consume_tile(tb, w0 + (k + 1) * stride, tail);
        }
        else {
            consume_tile(ta, w0 + k * stride, tail);
        }
    }
    NVK_STAMP(2)
    NVK_STAMP_WAVE()

    // the four lanes of a head (16 and 32 apart) join their shares of the sum; the max is already common
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);

    // token-split: the parts of one kv-head meet in LDS, record = (m, l, -, -, o[HD]) per (wave, head) of parts
    // 1 .. parts - 1 (part 0 merges from its registers: no record; make_shape sizes the area for parts - 1)
    if (parts > 1) {
        float *s_mrg = reinterpret_cast<float *>(s_tab + p.merge_off);
        auto   rec_of = [&](int w) { return s_mrg + ((w - p.kv_waves) * kv_mul + c16) * kRec; }; // w >= kv_waves
        if (part > 0 && head_live) {
            float *rec = rec_of(wave);
            if (q4 == 0)
                rec[0] = m_run, rec[1] = l_run;
#pragma unroll
            for (int j = 0; j < NV; j++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    *reinterpret_cast<float4 *>(rec + 4 + 64 * j + 16 * q4 + 4 * r) =
                        make_float4(oacc[j][0][r], oacc[j][1][r], oacc[j][2][r], oacc[j][3][r]);
        }
        __syncthreads();
        if (part > 0 && !fused)
            return;
        if (part == 0 && head_live) {
            float mw = m_run;
            for (int q = 1; q < parts; q++)
                mw = fmaxf(mw, rec_of(q * p.kv_waves + wave)[0]);
            const float f0 = fast_exp2(m_run - mw);
            float       lw = l_run * f0;
#pragma unroll
            for (int j = 0; j < NV; j++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    oacc[j][c] *= f0;
            for (int q = 1; q < parts; q++) {
                const float *rec = rec_of(q * p.kv_waves + wave);
                const float  f   = fast_exp2(rec[0] - mw);
                lw               = fmaf(rec[1], f, lw);
#pragma unroll
                for (int j = 0; j < NV; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float4 b = *reinterpret_cast<const float4 *>(rec + 4 + 64 * j + 16 * q4 + 4 * r);
                        oacc[j][0][r]  = fmaf(b.x, f, oacc[j][0][r]);
                        oacc[j][1][r]  = fmaf(b.y, f, oacc[j][1][r]);
                        oacc[j][2][r]  = fmaf(b.z, f, oacc[j][2][r]);
                        oacc[j][3][r]  = fmaf(b.w, f, oacc[j][3][r]);
                    }
            }
            m_run = mw, l_run = lw;
        }
    }

    const bool writer = head_live && part == 0;
    if (nch == 1) {
        if (!writer)
            return;
        const float mf  = fmaxf(m_run, kRefMaxFloorLog2); // the reference's max_val = -1e10 floor (attention.hpp:126-130)
        const float f   = fast_exp2(m_run - mf);
        const float inv = f / (l_run * f);
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                *reinterpret_cast<float4 *>(p.out + head_row * HD + 64 * j + 16 * q4 + 4 * r) =
                    make_float4(oacc[j][0][r] * inv, oacc[j][1][r] * inv, oacc[j][2][r] * inv, oacc[j][3][r] * inv);
        return;
    }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    constexpr int kSc1 = 16;
    if (writer) { // the chunk's unnormalised partial, write-through (see nvk_decode_kernel)
        const int64_t slab = (int64_t)seq * p.max_chunks + chunk;
        const __amdgpu_buffer_rsrc_t rs_o =
            __builtin_amdgcn_make_buffer_rsrc(p.ws_o + slab * p.n_heads * HD, 0, p.n_heads * HD * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_ml =
            __builtin_amdgcn_make_buffer_rsrc(p.ws_ml + slab * p.n_heads * 2, 0, p.n_heads * 8, 0x00020000);
        const int head = kvh * kv_mul + c16;
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4, make_float4(oacc[j][0][r], oacc[j][1][r], oacc[j][2][r], oacc[j][3][r])),
                    rs_o, (head * HD + 64 * j + 16 * q4 + 4 * r) * 4, 0, kSc1);
        if (q4 == 0)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(m_run, l_run)), rs_ml, head * 8, 0,
                                                  kSc1);
    }
    if (!fused)
        return;
    combine_if_last(p, s_tab, seq, nch, kvg, 0, 1, kv_mul);
}

// ---- split-K merge (flash-decoding combine) ---------------------------------
// One thread per (sequence, q-head, float4 of head_dim): log-sum-exp merge of
// the sequence's chunk partials, in chunk order, then the reference's
// normalisation.  Sequences with at most one chunk were finished by
// nvk_decode_kernel and are skipped.
// Chunks of row `seq`: from the plan's per-row counts, or -- no plan -- from the row's length (null: uniform_ctx).
__device__ __forceinline__ int row_chunks(const int *seq_chunks, const int *ctx_lens, int uniform_ctx, int chunk_tokens,
                                          int max_chunks, int seq)
{
    if (seq_chunks)
        return min(seq_chunks[seq], max_chunks);
    const int ctx = max(ctx_lens ? ctx_lens[seq] : uniform_ctx, 0);
    return (int)min((int64_t)max_chunks, ((int64_t)ctx + chunk_tokens - 1) / chunk_tokens);
}

__global__ void __launch_bounds__(256) nvk_merge_kernel(const float *__restrict__ ws_o,
                                                        const float *__restrict__ ws_ml,
                                                        const int *__restrict__ seq_chunks,
                                                        const int *__restrict__ ctx_lens, int uniform_ctx,
                                                        int chunk_tokens, float *__restrict__ out, int n_heads,
                                                        int head_dim, int max_chunks)
{
    const int seq = blockIdx.x; // rows on x: no 65535 limit on the batch
    const int nch = row_chunks(seq_chunks, ctx_lens, uniform_ctx, chunk_tokens, max_chunks, seq);
    if (nch <= 1)
        return;
    const int vec_row = head_dim >> 2; // float4 per head
    const int idx     = blockIdx.y * blockDim.x + threadIdx.x;
    if (idx >= n_heads * vec_row)
        return;
    const int head = idx / vec_row;
    const int vec  = idx - head * vec_row;

    merge_chunks(ws_o, ws_ml, out, seq, head, vec, nch, n_heads, head_dim, max_chunks);
}

// ---- split-K merge for few rows and many chunks ------------------------------
// nvk_merge_kernel walks a row's chunks one after the other: with a handful of sequences cut into dozens of short
// chunks (the shape AUTO picks to occupy the chip with one request) that walk is a chain of dependent-latency loads
// on a nearly empty GPU.  Here one workgroup owns one (sequence, q-head): its 256 threads take one chunk's (m, l)
// each, agree on the maximum and the denominator through two block reductions, then split into 256 / (head_dim / 4)
// groups that each accumulate every groups-th chunk's partial row; the groups meet in LDS and are added in group
// order.  Needs max_chunks <= 256.  Same (m, l, o) convention as nvk_merge_kernel; the order of the sums differs.
constexpr int kMergeWideMaxChunks = 256;

__global__ void __launch_bounds__(256) nvk_merge_wide_kernel(const float *__restrict__ ws_o,
                                                             const float *__restrict__ ws_ml,
                                                             const int *__restrict__ seq_chunks,
                                                             const int *__restrict__ ctx_lens, int uniform_ctx,
                                                             int chunk_tokens, float *__restrict__ out, int n_heads,
                                                             int head_dim, int max_chunks)
{
    __shared__ float  s_f[kMergeWideMaxChunks];
    __shared__ float  s_red[8];
    __shared__ float4 s_acc[256];
    const int seq = blockIdx.x, head = blockIdx.y;
    const int nch = row_chunks(seq_chunks, ctx_lens, uniform_ctx, chunk_tokens, max_chunks, seq); // uniform over the workgroup
    if (nch <= 1)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)seq * max_chunks * n_heads + head; // chunk c's slab row = row0 + c * n_heads

    float2 s = make_float2(kRefMaxFloorLog2, 0.f);
    if (tid < nch)
        s = *reinterpret_cast<const float2 *>(ws_ml + (row0 + (int64_t)tid * n_heads) * 2);
    float top = s.x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
        top = fmaxf(top, __shfl_xor(top, d));
    if (lane == 0)
        s_red[wave] = top;
    __syncthreads();
    top = fmaxf(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])), kRefMaxFloorLog2);
    const float f = tid < nch ? fast_exp2(s.x - top) : 0.f;
    s_f[tid]      = f;
    float denom   = s.y * f;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
        denom += __shfl_xor(denom, d);
    if (lane == 0)
        s_red[4 + wave] = denom;
    __syncthreads();
    denom = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);

    const int vec_row = head_dim >> 2, groups = 256 / vec_row;
    const int g = tid / vec_row, vec = tid - g * vec_row;
    float4    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < groups) {
        const float *po = ws_o + row0 * head_dim + 4 * vec;
        const int64_t cs = (int64_t)n_heads * head_dim; // floats between chunks
        int c = g;
        for (; c + groups < nch; c += 2 * groups) { // two independent loads in flight
            const float4 a = *reinterpret_cast<const float4 *>(po + c * cs);
            const float4 b = *reinterpret_cast<const float4 *>(po + (c + groups) * cs);
            const float  fa = s_f[c], fb = s_f[c + groups];
            acc.x = fmaf(a.x, fa, acc.x), acc.y = fmaf(a.y, fa, acc.y), acc.z = fmaf(a.z, fa, acc.z), acc.w = fmaf(a.w, fa, acc.w);
            acc.x = fmaf(b.x, fb, acc.x), acc.y = fmaf(b.y, fb, acc.y), acc.z = fmaf(b.z, fb, acc.z), acc.w = fmaf(b.w, fb, acc.w);
        }
        if (c < nch) {
            const float4 a = *reinterpret_cast<const float4 *>(po + c * cs);
            const float  fa = s_f[c];
            acc.x = fmaf(a.x, fa, acc.x), acc.y = fmaf(a.y, fa, acc.y), acc.z = fmaf(a.z, fa, acc.z), acc.w = fmaf(a.w, fa, acc.w);
        }
    }
    s_acc[tid] = acc;
    __syncthreads();
    if (tid < vec_row) {
        float4 sum = s_acc[tid];
        for (int k = 1; k < groups; k++) {
            const float4 a = s_acc[k * vec_row + tid];
            sum.x += a.x, sum.y += a.y, sum.z += a.z, sum.w += a.w;
        }
        const float inv = 1.0f / denom;
        *reinterpret_cast<float4 *>(out + ((int64_t)seq * n_heads + head) * head_dim + 4 * tid) =
            make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
    }
}

// ---- KV gather on its own (addressing check) --------------------------------
// grid.x = tokens; copies K and V rows of one kv-head to [num_tokens, head_dim].
__global__ void nvk_gather_kernel(const float *__restrict__ k_layer, const float *__restrict__ v_layer,
                                  const int *__restrict__ block_table, int num_tokens, int kv_head, int num_blocks,
                                  int block_size, unsigned bs_magic, int n_kv_heads, int head_dim,
                                  float *__restrict__ k_out, float *__restrict__ v_out)
{
    const int tok = blockIdx.x;
    if (tok >= num_tokens)
        return;
    const int     blk = token_block(tok, bs_magic);
    const int     off = token_offset(tok, blk, block_size);
    const int     pb  = clamp_block(block_table[blk], num_blocks);
    const int64_t at  = row_float_index(pb, off, kv_head, block_size, n_kv_heads, head_dim);
    for (int d = threadIdx.x; d < head_dim; d += blockDim.x) {
        k_out[(int64_t)tok * head_dim + d] = k_layer[at + d];
        v_out[(int64_t)tok * head_dim + d] = v_layer[at + d];
    }
}

// ---- batched KV append (reference: model.hpp:190-207, once per sequence) -----
// grid.x = sequences; a workgroup copies one K row and one V row of
// n_kv_heads*head_dim floats to the slot positions[s] of its sequence.
__global__ void nvk_append_kernel(float *__restrict__ k_layer, float *__restrict__ v_layer,
                                  const float *__restrict__ k_rows, const float *__restrict__ v_rows,
                                  const int *__restrict__ block_tables, int max_blocks,
                                  const int *__restrict__ positions, int num_blocks, int block_size,
                                  unsigned bs_magic, int row_len)
{
    const int seq = blockIdx.x;
    const int pos = positions[seq];
    if (pos < 0)
        return;
    const int     blk = token_block(pos, bs_magic);
    const int     off = token_offset(pos, blk, block_size);
    const int     pb  = clamp_block(block_tables[(int64_t)seq * max_blocks + blk], num_blocks);
    const int64_t at  = ((int64_t)pb * block_size + off) * row_len;
    for (int i = threadIdx.x * 4; i < row_len; i += blockDim.x * 4) { // row_len % 4 == 0 (checked on the host)
        *reinterpret_cast<float4 *>(k_layer + at + i) =
            *reinterpret_cast<const float4 *>(k_rows + (int64_t)seq * row_len + i);
        *reinterpret_cast<float4 *>(v_layer + at + i) =
            *reinterpret_cast<const float4 *>(v_rows + (int64_t)seq * row_len + i);
    }
}

// ---- paged chunked prefill on fp32 MFMA tiles -----------
// Reference maths: LlamaModelChunked::chunked_attention (experimental/chunked_prefill/
// model_chunked.hpp:167-224): query b of a chunk attends to keys t <= start_pos + b.  Here K/V
// come through the block table (attention.hpp:109-115) and every K/V tile is read from HBM once
// per 32..128 queries x all grouped q-heads instead of once per (query, head).
//
//   * work item = (sequence, tile of up to 32*qsubs consecutive query tokens); blockIdx.y =
//     (kv-head, q-head sub-group).  4 waves: wave w owns q-head (w % hpw) of the group and query
//     sub-tile (w / hpw) of 32 tokens; all four share each 32-key K/V tile through LDS.
//   * S^T[key][query] = K . Q^T on v_mfma_f32_32x32x2_f32 (exact fp32, an fmaf chain): A = K tile
//     rows from LDS, B = the wave's Q fragment held in registers for the whole key loop.  The k
//     index of the instruction is re-associated: lane half h covers dims [h*HDP/2, (h+1)*HDP/2),
//     so a lane's operands are contiguous floats (ds_read_b128, float4 global loads).
//   * the result has the query on the lane and the 32 keys in 16 registers x 2 lane halves
//     (row = (r&3) + 8*(r>>2) + 4*half), so the online-softmax statistics are per lane plus one
//     exchange with lane^32, and register r is directly the B operand of the second product
//     O^T[dim][query] += V^T[dim][key] . P^T[key][query] (A = V from LDS, one float per lane):
//     no transpose, no LDS round trip for P.
//   * softmax in the exp2 domain, running max seeded finite, the reference's -1e10 floor applied
//     at the end (as in nvk_decode_kernel).  Keys past a query's position are masked to -inf
//     only in the tiles that straddle the diagonal.
//   * next K/V tile is fetched global -> registers while the current one is consumed.
//   * split-K: when a call has too few (tile, head-group) pairs to fill the chip (one long prompt fed in
//     chunks), the host cuts every tile's key range into parts; each part is an item that writes an
//     unnormalised (m, l, o) partial per query row, combined by nvk_merge_kernel exactly as decode chunks are.
// Bound: fp32 MFMA (64 FLOP/clk/SIMD); 128 MFMAs of 64 cycles per wave per 32x32 tile at HDP = 128.
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct __attribute__((aligned(16))) PrefillItem
{
    int seq, row0, pos0, nq; // table row, first packed q/out row, position of that query, queries in the tile
    int k0, k1;              // keys [k0, k1) of the context are this item's share (k0 a multiple of 32)
    int part, nparts;        // nparts > 1: the tile's key range is split over several items ("split-K"); each
                             // writes an unnormalised partial and nvk_merge_kernel combines them
};

struct PrefillParams
{
    const float       *k_layer, *v_layer, *q;
    const int         *block_tables;
    const PrefillItem *items;
    float             *out;
    float             *ws_o, *ws_ml; // split-K partials: [row, max_parts, n_heads, head_dim] and [.., 2]
    int                max_parts;
    int                max_blocks, num_blocks, block_size;
    unsigned           bs_magic;
    int                n_kv_heads, n_heads, kv_mul, head_dim;
    int                hpw; // q-heads of one kv-head per workgroup: 1, 2 or 4 (divides kv_mul)
    float              q_scale;
};

#ifndef NVK_PREFILL_MIN_WAVES
#define NVK_PREFILL_MIN_WAVES 2 // 2 waves/SIMD: one wave's softmax overlaps the other's MFMAs (+11-15 % measured)
#endif
template <int HDP>
__global__ void __launch_bounds__(256, NVK_PREFILL_MIN_WAVES) nvk_prefill_kernel(const PrefillParams p)
{
    constexpr int KS  = HDP + 4; // K tile row stride in floats: 16 lanes x b128 land in 64 different banks
    constexpr int VS  = HDP + 8; // V tile row stride: rows r and r+4 (the two lane halves) 32 banks apart
    constexpr int VPT = HDP / 32; // float4 per thread per tile (32 rows x HDP/4 vectors over 256 threads)
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float *s_k   = s_mem;
    float *s_v   = s_mem + 32 * KS;
    int   *s_tab = reinterpret_cast<int *>(s_v + 32 * VS);

    // 1-D grid, head group fastest: workgroups launched together are the head groups of ONE item, walking the
    // same token rows at the same time, so whole [n_kv_heads * head_dim] rows are consumed together (DRAM pages,
    // Infinity Cache) instead of one 512-B piece of many rows
    const int hpw    = p.hpw;
    const int groups = p.kv_mul / hpw;
    const int grid_y = p.n_kv_heads * groups;
    const int by     = blockIdx.x % grid_y;
    const PrefillItem item = p.items[blockIdx.x / grid_y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int kvh    = by / groups;
    const int head   = kvh * p.kv_mul + (by % groups) * hpw + wave % hpw;
    const int wq0    = 32 * (wave / hpw);            // first query of this wave inside the item
    const int my_nq  = min(32, item.nq - wq0);       // <= 0: the wave only helps with the loads
    const bool q_live = j < my_nq;
    const int qpos   = q_live ? item.pos0 + wq0 + j : 0x7fffffff; // dead lanes mask nothing, store nothing
    const int hd     = p.head_dim;
    const int kbeg   = item.k0;                                    // first key of this item's share
    const int kmax   = min(item.k1, item.pos0 + item.nq) - 1;      // last one (no query of the tile sees further)
    const int row_len = p.n_kv_heads * hd;

    const int blk0 = token_block(kbeg, p.bs_magic);
    const int nblk = token_block(kmax, p.bs_magic) - blk0 + 1;
    for (int i = tid; i < nblk; i += 256)
        s_tab[i] = clamp_block(p.block_tables[(int64_t)item.seq * p.max_blocks + blk0 + i], p.num_blocks);

    // this wave's Q fragment: lane (query j, half) holds dims [half*HDP/2, (half+1)*HDP/2), pre-scaled
    float qf[HDP / 2];
    {
        const float *qrow = p.q + ((int64_t)(item.row0 + wq0 + j) * p.n_heads + head) * hd;
#pragma unroll
        for (int s4 = 0; s4 < HDP / 8; s4++) {
            const int d = half * (HDP / 2) + 4 * s4;
            float4    v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q_live && d < hd)
                v = *reinterpret_cast<const float4 *>(qrow + d);
            qf[4 * s4 + 0] = v.x * p.q_scale, qf[4 * s4 + 1] = v.y * p.q_scale;
            qf[4 * s4 + 2] = v.z * p.q_scale, qf[4 * s4 + 3] = v.w * p.q_scale;
        }
    }
    __syncthreads(); // table slice staged

    float4 kreg[VPT], vreg[VPT];
    auto fetch_tile = [&](int t0) {
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            const int idx = tid + 256 * u;
            const int row = idx / (HDP / 4), c4 = idx % (HDP / 4);
            const int tok = t0 + row;
            kreg[u] = vreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tok <= kmax && 4 * c4 < hd) {
                const int     blk = token_block(tok, p.bs_magic);
                const int     off = token_offset(tok, blk, p.block_size);
                const int64_t at  = ((int64_t)s_tab[blk - blk0] * p.block_size + off) * row_len + kvh * hd + 4 * c4;
                kreg[u]           = *reinterpret_cast<const float4 *>(p.k_layer + at);
                vreg[u]           = *reinterpret_cast<const float4 *>(p.v_layer + at);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            const int idx = tid + 256 * u;
            const int row = idx / (HDP / 4), c4 = idx % (HDP / 4);
            *reinterpret_cast<float4 *>(s_k + row * KS + 4 * c4) = kreg[u];
            *reinterpret_cast<float4 *>(s_v + row * VS + 4 * c4) = vreg[u];
        }
    };

    float  m = kNegBig, l = 0.f;
    f32x16 o[HDP / 32];
#pragma unroll
    for (int c = 0; c < HDP / 32; c++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            o[c][r] = 0.f;

    const int ntiles    = (kmax - kbeg) / 32 + 1;
    const int wave_last = item.pos0 + wq0 + my_nq - 1; // last position this wave's queries reach
    fetch_tile(kbeg);
    store_tile();
    __syncthreads();
    for (int it = 0; it < ntiles; it++) {
        const int t0 = kbeg + 32 * it;
        if (it + 1 < ntiles)
            fetch_tile(t0 + 32);
        if (my_nq > 0 && t0 <= wave_last) { // wave-uniform
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; r++)
                sacc[r] = 0.f;
            const float *krow = s_k + j * KS + half * (HDP / 2); // A operand: key row = lane & 31
#pragma unroll
            for (int s4 = 0; s4 < HDP / 8; s4++) {
                const float4 kv = *reinterpret_cast<const float4 *>(krow + 4 * s4);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qf[4 * s4 + 0], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qf[4 * s4 + 1], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qf[4 * s4 + 2], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qf[4 * s4 + 3], sacc, 0, 0, 0);
            }
            // keys after the query's own position are not attended to (model_chunked.hpp:186)
            if (t0 + 31 > item.pos0 + wq0 || t0 + 31 > kmax) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = t0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key > qpos || key > kmax)
                        sacc[r] = -INFINITY;
                }
            }
            float mx = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; r++)
                mx = fmaxf(mx, sacc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn    = fmaxf(m, mx);
            const float alpha = fast_exp2(m - mn);
            float       psum  = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                sacc[r] = fast_exp2(sacc[r] - mn);
                psum += sacc[r];
            }
            m = mn;
            l = fmaf(l, alpha, psum); // this lane half's share of the row sum; joined at the end
#ifndef NVK_PREFILL_ALWAYS_RESCALE
            // the running max settles after the first tiles of a row: skip the HDP/2 multiplies per lane
            // whenever no query of the wave saw a new maximum (alpha == 1 everywhere)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0)
#endif
            {
#pragma unroll
                for (int c = 0; c < HDP / 32; c++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        o[c][r] *= alpha;
            }
#pragma unroll
            for (int c = 0; c < HDP / 32; c++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int   key = (r & 3) + 8 * (r >> 2) + 4 * half; // the key register r of this half holds
                    const float vv  = s_v[key * VS + 32 * c + j];        // A operand: dim = 32c + (lane & 31)
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sacc[r], o[c], 0, 0, 0);
                }
            }
        }
        __syncthreads(); // every wave is done with this tile
        if (it + 1 < ntiles) {
            store_tile();
            __syncthreads();
        }
    }

    if (!q_live)
        return;
    l += __shfl_xor(l, 32); // q_live is the same for lane and lane^32
    if (item.nparts > 1) {
        // split-K: the unnormalised partial of this key range; rows whose queries precede it carry (m, l, o) = (-big, 0, 0)
        const int64_t rec = (((int64_t)(item.row0 + wq0 + j) * p.max_parts + item.part) * p.n_heads + head);
        if (half == 0)
            *reinterpret_cast<float2 *>(p.ws_ml + rec * 2) = make_float2(m, l);
#pragma unroll
        for (int c = 0; c < HDP / 32; c++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int dim = 32 * c + 8 * g + 4 * half;
                if (dim < hd)
                    *reinterpret_cast<float4 *>(p.ws_o + rec * hd + dim) =
                        make_float4(o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]);
            }
        return;
    }
    const float mf  = fmaxf(m, kRefMaxFloorLog2); // the reference's max_val = -1e10 floor (model_chunked.hpp:197)
    const float f   = fast_exp2(m - mf);
    const float inv = f / (l * f);
    float      *orow = p.out + ((int64_t)(item.row0 + wq0 + j) * p.n_heads + head) * hd;
#pragma unroll
    for (int c = 0; c < HDP / 32; c++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int dim = 32 * c + 8 * g + 4 * half; // registers 4g..4g+3 are four consecutive dims
            if (dim < hd)
                *reinterpret_cast<float4 *>(orow + dim) = make_float4(o[c][4 * g] * inv, o[c][4 * g + 1] * inv,
                                                                      o[c][4 * g + 2] * inv, o[c][4 * g + 3] * inv);
        }
}

// ---- RoPE fused with the batched KV append --------------
// Reference: Ops::apply_rope (ops/positional.hpp:13-39) as called at model.hpp:169-175,
// then the append at model.hpp:190-207.  grid.x = sequences.  Per sequence:
//   angle_p = (float)pos * inv_freq[p],  inv_freq[p] = 1 / powf(theta, 2p / head_dim)  (host libm, exactly
//   the reference's expression: a 1-ulp difference in the frequency would be amplified by pos);
//   cos/sin evaluated in fp64 and rounded once, so they agree with a correctly rounded cosf/sinf;
//   the rotation itself is unfused fp32 mul/sub/add (rope_rotate), the order of positional.hpp:26-27, :35-36.
// q is rotated in place (as the reference rotates state.q); the rotated K row and the V row go
// straight to the pool slot of token `pos` (k_rows itself is left untouched).
// (hipcc contracts a*b - c*d into an fma by default, and __fmul_rn/__fsub_rn are plain operators
// in this ROCm: the pragma is what keeps the four products and the two sums separately rounded)
__global__ void __launch_bounds__(256)
    nvk_rope_append_kernel(float *__restrict__ k_layer, float *__restrict__ v_layer, float *__restrict__ q,
                           const float *__restrict__ k_rows, const float *__restrict__ v_rows,
                           const int *__restrict__ block_tables, const int *__restrict__ table_rows, int max_blocks,
                           const int *__restrict__ positions, const float *__restrict__ inv_freq, int num_blocks,
                           int block_size, unsigned bs_magic, int n_heads, int n_kv_heads, int head_dim)
{
    extern __shared__ __attribute__((aligned(16))) float2 s_cs[]; // [head_dim / 2] (cos, sin)
    const int seq = blockIdx.x;
    const int pos = positions[seq];
    if (pos < 0)
        return;
    const int  half   = head_dim >> 1;
    const bool rotate_on = inv_freq != nullptr; // null: plain append, q untouched, K copied as it is
    for (int p = threadIdx.x; p < half; p += blockDim.x) {
        float2 cs = make_float2(1.f, 0.f);
        if (rotate_on) {
            const float angle = (float)pos * inv_freq[p];
            double      sn, c;
            sincos((double)angle, &sn, &c);
            cs = make_float2((float)c, (float)sn);
        }
        s_cs[p] = cs;
    }
    __syncthreads();

    auto rotate = [](float2 v, float2 cs) { return rope_rotate(v, cs); };

    if (rotate_on) {
        float2 *q2 = reinterpret_cast<float2 *>(q + (int64_t)seq * n_heads * head_dim);
        for (int i = threadIdx.x; i < n_heads * half; i += blockDim.x)
            q2[i] = rotate(q2[i], s_cs[i % half]);
    }

    const int     row_len = n_kv_heads * head_dim;
    const int     blk     = token_block(pos, bs_magic);
    const int     off     = token_offset(pos, blk, block_size);
    const int     trow    = table_rows ? table_rows[seq] : seq; // chunked prefill: several rows per table row
    const int     pb      = clamp_block(block_tables[(int64_t)trow * max_blocks + blk], num_blocks);
    const int64_t at      = ((int64_t)pb * block_size + off) * row_len;
    const float2 *k2      = reinterpret_cast<const float2 *>(k_rows + (int64_t)seq * row_len);
    float2       *kd      = reinterpret_cast<float2 *>(k_layer + at);
    for (int i = threadIdx.x; i < n_kv_heads * half; i += blockDim.x)
        kd[i] = rotate_on ? rotate(k2[i], s_cs[i % half]) : k2[i];
    for (int i = threadIdx.x * 4; i < row_len; i += blockDim.x * 4)
        *reinterpret_cast<float4 *>(v_layer + at + i) =
            *reinterpret_cast<const float4 *>(v_rows + (int64_t)seq * row_len + i);
}

// ---- synthetic fill: the counter-based generator of SURVEY.md 8d -------------
__device__ __forceinline__ float uniform_from_counter(uint64_t seed, uint64_t index)
{
    uint64_t x = (seed ^ index) + 0x9E3779B97F4A7C15ULL;
    x          = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x          = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (float)(x >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
}

__global__ void nvk_fill_kernel(float *__restrict__ dst, uint64_t seed, uint64_t count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += stride) {
        if (i + 4 <= count) {
            *reinterpret_cast<float4 *>(dst + i) =
                make_float4(uniform_from_counter(seed, i), uniform_from_counter(seed, i + 1),
                            uniform_from_counter(seed, i + 2), uniform_from_counter(seed, i + 3));
        }
        else {
            for (uint64_t k = i; k < count; k++)
                dst[k] = uniform_from_counter(seed, k);
        }
    }
}

} // namespace nvk
