// nvk_linear.hip -- the producers of q/k/v on the device (second half):
// RMSNorm and the Q/K/V projections of a batch of token rows, so that the rows the attention path
// consumes never cross PCIe.
//
// Reference: Ops::rms_norm (include/ops/normalization.hpp:13-23) and Ops::matmul
// (include/ops/linear.hpp:12-22) as LlamaModel::forward calls them (include/core/model.hpp:161-166),
// and their batched forms BatchOps::batch_rms_norm / batch_matmul
// (experimental/chunked_prefill/batch_ops.hpp:58-76, :8-23).  Weights are [out_dim, in_dim]
// row-major and out[b][i] = dot(in[b], weight[i]): C = A . W^T with both operands contiguous along
// the contraction ("NT"), all fp32.
//
// Three kernels behind one dispatcher (thresholds measured, profiles/r01_qkv_projection.log):
//   * nvk_gemv_kernel        -- 1 or 2 rows (the reference's own case: Ops::matmul is a GEMV): weights
//                               streamed once and multiplied on the vector ALUs; bound by HBM;
//   * nvk_gemm_skinny_kernel -- up to 96 rows (decode batches of a few requests): 16 x 16 (up to 16 rows) or
//                               32 x 32 output tiles on fp32 MFMA, the contraction split over the workgroup's
//                               8 waves; weights streamed once per tile of rows; HBM / MFMA about level;
//   * nvk_gemm_nt_kernel     -- more rows: 64 x 64 or 128 x 128 tiles per workgroup on the same MFMA
//                               (exact fp32 products and sums, an fmaf chain per output); bound by the
//                               fp32 MFMA rate.
// All take up to three (weight, out) segments in one launch, so wq, wk and wv share the launch and
// the normalised activations.  gfx950 only; no CPU path.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "nvk_device_scope.h"
#include "nvk_addr.hip.h"
#include "nvk_internal.h"

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return nvk::fail(e__ == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP, "%s failed: %s (%s:%d)", #expr,  \
                             hipGetErrorString(e__), __FILE__, __LINE__);                                              \
    } while (0)

#ifndef NVK_K_ROTATE
#define NVK_K_ROTATE 1 // A/B switch for the rotated walk over the contraction dimension
#endif
#ifndef NVK_GEMM_FORCE_TILE
#define NVK_GEMM_FORCE_TILE 0 // tuning: 128 or 64 pins the tiled kernel's tile, 0 = by workgroup count
#endif
#ifndef NVK_GEMV_COLS
#define NVK_GEMV_COLS 4 // weight rows (output columns) per GEMV workgroup: 4, 8 or 16.  Measured per shape
                        // (tools/gemv_sweep.sh, profiles/r01_qkv_projection.log): 4 is the fastest for one row at every
                        // size (2048 x 2048 is 512 workgroups instead of 128, and fewer registers let more of them
                        // share a CU), and for two rows up to a few thousand columns (NVK_GEMV2_MAX_COLS)
#endif
#ifndef NVK_GEMV2_MAX_COLS
#define NVK_GEMV2_MAX_COLS 4096 // two rows: the GEMV up to this many output columns, above it the 16 x 16 MFMA kernel
                                // (measured: 2 x 5632 columns 22.6 us GEMV / 18.9 us MFMA, 32000 columns 50.2 / 44.8;
                                // 2560 columns 6.4 / 7.6, 2048 columns of 5632 floats 10.3 / 15.9)
#endif
#ifndef NVK_GEMV_DEPTH
#define NVK_GEMV_DEPTH 2 // steps of weight loads a GEMV wave keeps in flight
#endif
#ifndef NVK_GEMV_MAX_ROWS
#define NVK_GEMV_MAX_ROWS 2 // up to here the vector-ALU GEMV kernel (measured: from 3 rows on its registers allow one
                            // workgroup per CU and the MFMA kernel is as fast); at most 2, 0 switches it off
#endif
#ifndef NVK_SKINNY_TM
#define NVK_SKINNY_TM 1 // 33-96 rows: one wave multiplies 2-3 row tiles by each weight tile (0: separate workgroups per row tile)
#endif
#ifndef NVK_SKINNY16_MAX_ROWS
#define NVK_SKINNY16_MAX_ROWS 16 // up to here the weight-streaming kernel uses 16 x 16 MFMA tiles (0: always 32 x 32)
#endif
#ifndef NVK_SKINNY16_BALANCED
#define NVK_SKINNY16_BALANCED 1 // 16-row tiles: output columns dealt evenly over the CUs (nvk_gemm_skinny16_kernel); 0: 16-column tiles
#endif
#ifndef NVK_SKINNY16_BK
#define NVK_SKINNY16_BK 64 // k-step of the 16 x 16 tile kernel without an epilogue (32: half the LDS per workgroup, twice the steps)
#endif
#ifndef NVK_SKINNY16_TM2
#define NVK_SKINNY16_TM2 0 // 1: 17-32 rows on 16 x 16 tiles with both row tiles in one wave (weights cross L2 once).  Built and measured in
                           // round 3: 32.3 us against 33.0 at 6144 columns -- the kernel is not bound by the second reading -- not kept
#endif
#ifndef NVK_SKINNY16_WIDE
#define NVK_SKINNY16_WIDE 1 // 17-64 rows over few columns: 16 x 16 tiles, one workgroup per row tile (0: 32 x 32 tiles)
#endif
#ifndef NVK_SKINNY16_BALANCE_PCT
#define NVK_SKINNY16_BALANCE_PCT 25 // the balanced 16-row kernel where 16-column tiles leave the busiest CU this much above the mean
#endif
#ifndef NVK_SKINNY16_SPLIT
#define NVK_SKINNY16_SPLIT 1 // up to 16 rows over at most half as many 16-column tiles as CUs: the contraction cut into slices
#endif
#ifndef NVK_NT_SPLIT_MIN_ROWS
#define NVK_NT_SPLIT_MIN_ROWS 97 // from here a plain product whose 64 x 64 tiles leave CUs idle is cut along k (nvk_gemm_nt_kernel<.., SPLIT>)
#endif
#ifndef NVK_NT_STREAM_MIN_ROWS
#define NVK_NT_STREAM_MIN_ROWS 33 // from here a plain product runs nvk_gemm_nt_kernel<64, 64, 64, kNtStream> (stream-K) where ...
#endif
#ifdef NVK_LINEAR_ENV
#define NVK_LINEAR_ENV_ON 1
#else
#define NVK_LINEAR_ENV_ON 0
#endif
#ifndef NVK_NT_STREAM128_MIN_ROWS
#define NVK_NT_STREAM128_MIN_ROWS 100000 // from here: stream-K over 128 x 128 tiles where those are fewer than six to a CU (measured: off)
#endif
#ifndef NVK_NT_STREAM128_FILL
#define NVK_NT_STREAM128_FILL 1 // workgroups per CU of that launch
#endif
#ifndef NVK_NT_STREAM32_MIN_STEPS
#define NVK_NT_STREAM32_MIN_STEPS 20 // 17-32 rows: stream-K over 32 x 128 tiles from this many 32-steps to a CU (0: never)
#endif
#ifndef NVK_NT_STREAM_MAX_TILES_PER_CU
#define NVK_NT_STREAM_MAX_TILES_PER_CU 3 // ... its 64 x 64 tiles are fewer than this many per CU (more: plain tiles balance by themselves)
#endif
#ifndef NVK_NT_STREAM_FILL
#define NVK_NT_STREAM_FILL 2 // workgroups per CU of a stream-K launch
#endif
#ifndef NVK_NT_SPLIT_FILL
#define NVK_NT_SPLIT_FILL 2 // ... so that the launch is this many workgroups per CU (two fit)
#endif
#ifndef NVK_NT_SPLIT_MIN_K
#define NVK_NT_SPLIT_MIN_K 256 // ... into slices of at least this many contraction steps
#endif
#ifndef NVK_SKINNY_MAX_ROWS
#define NVK_SKINNY_MAX_ROWS 96 // up to here the weight-streaming MFMA kernel (32-row tiles), above it the tiled one
#endif

static_assert(NVK_GEMV_MAX_ROWS >= 0 && NVK_GEMV_MAX_ROWS <= 2, "nvk_gemv_kernel is instantiated for one and two rows");

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "nvk_linear.hip: the split-K tile combine is written for gfx950 (sc1 write-through stores); build with --offload-arch=gfx950"
#endif
namespace nvk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// weights are read once per launch: keep them out of the way of the activations in the caches
__device__ __forceinline__ float4 load_stream(const float *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

struct LinearSeg
{
    const float *w;     // [n, k] row-major
    float       *out;   // [m, ldo]
    int          n, ldo;
    int          tile0; // index of the segment's first column tile in the launch
};

struct LinearParams
{
    const float *a; // [m, lda], k floats used per row
    int          m, k, lda;
    LinearSeg    seg[3];
    int          nseg, n_tiles, m_tiles;
    int          cw; // nvk_gemm_skinny16_kernel: output columns per workgroup (a multiple of 4)
    // nvk_gemm_nt_kernel<.., SPLIT>: the contraction cut into `ksplit` slices of `kslice` (a multiple of BK), one
    // workgroup per (tile, slice); partial tiles in `part` [tile][slice][BM x BN], arrivals per tile in `arrive` (zero
    // between launches: the workgroup that completes a tile sums its slices in slice order and zeroes the counter)
    int          ksplit, kslice;
    float       *part;
    int         *arrive;
    RopeAppendTarget rope; // nvk_gemv_kernel<.., ROPE>: where the q / k / v epilogue writes
    NormLink         link; // nvk_gemv_kernel<.., SCALE / RESID>: the norm carried from one GEMV to the next
};

__device__ __forceinline__ const LinearSeg &segment_of(const LinearParams &p, int tile)
{
    int s = 0;
    if (p.nseg > 1 && tile >= p.seg[1].tile0)
        s = 1;
    if (p.nseg > 2 && tile >= p.seg[2].tile0)
        s = 2;
    return p.seg[s];
}

// ---- RMSNorm: one workgroup per row ---------------------------------------------------------
// out[i] = in[i] * rms * weight[i], rms = 1 / sqrtf(sum(in^2) / dim + eps)   (normalization.hpp:15-22)
// The sum of squares is a tree over 256 threads instead of the reference's serial loop (fp32
// association differs, inside the tolerance); the two multiplies keep the reference's order.
// ADD: first x += addend (the residual add of model.hpp:227-228 / 238-239 that precedes every norm but the first),
// written back to `in`, then the norm of the sum.
template <bool ADD>
__global__ void __launch_bounds__(256) nvk_rms_norm_kernel(float *__restrict__ out, float *in, const float *__restrict__ addend,
                                                           const float *__restrict__ weight, int dim, float eps)
{
    __shared__ float s_part[4];
    float           *x = in + (int64_t)blockIdx.x * dim;
    const float     *a = ADD ? addend + (int64_t)blockIdx.x * dim : nullptr;
    float           *y = out + (int64_t)blockIdx.x * dim;
    float            ss = 0.f;
    const int        vec = (dim % 4 == 0) ? dim / 4 : 0;
    // rows of up to 8192 floats stay in registers between the two passes (one read of x instead of two)
    constexpr int KEEP = 8;
    const bool    keep = vec <= KEEP * 256;
    float4        held[KEEP];
#pragma unroll
    for (int u = 0; u < KEEP; u++) {
        const int i = threadIdx.x + 256 * u;
        held[u]     = make_float4(0.f, 0.f, 0.f, 0.f);
        if (keep && i < vec) {
            held[u] = *reinterpret_cast<const float4 *>(x + 4 * i);
            if (ADD) {
                const float4 b = *reinterpret_cast<const float4 *>(a + 4 * i);
                held[u].x += b.x, held[u].y += b.y, held[u].z += b.z, held[u].w += b.w;
                *reinterpret_cast<float4 *>(x + 4 * i) = held[u];
            }
            ss = fmaf(held[u].x, held[u].x, ss), ss = fmaf(held[u].y, held[u].y, ss);
            ss = fmaf(held[u].z, held[u].z, ss), ss = fmaf(held[u].w, held[u].w, ss);
        }
    }
    if (!keep)
        for (int i = threadIdx.x; i < vec; i += 256) {
            float4 v = *reinterpret_cast<const float4 *>(x + 4 * i);
            if (ADD) {
                const float4 b = *reinterpret_cast<const float4 *>(a + 4 * i);
                v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
                *reinterpret_cast<float4 *>(x + 4 * i) = v; // re-read below by the same thread
            }
            ss = fmaf(v.x, v.x, ss), ss = fmaf(v.y, v.y, ss), ss = fmaf(v.z, v.z, ss), ss = fmaf(v.w, v.w, ss);
        }
    for (int i = 4 * vec + threadIdx.x; i < dim; i += 256) {
        if (ADD)
            x[i] += a[i];
        ss = fmaf(x[i], x[i], ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0)
        s_part[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const float rms = 1.0f / sqrtf(ss / dim + eps);
    {
#pragma clang fp contract(off) // (in * rms) * weight, two roundings, as the reference's expression
        if (keep) {
#pragma unroll
            for (int u = 0; u < KEEP; u++) {
                const int i = threadIdx.x + 256 * u;
                if (i < vec) {
                    const float4 v = held[u];
                    const float4 g = *reinterpret_cast<const float4 *>(weight + 4 * i);
                    float4       r;
                    r.x = v.x * rms * g.x, r.y = v.y * rms * g.y, r.z = v.z * rms * g.z, r.w = v.w * rms * g.w;
                    *reinterpret_cast<float4 *>(y + 4 * i) = r;
                }
            }
        }
        else
            for (int i = threadIdx.x; i < vec; i += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(x + 4 * i);
                const float4 g = *reinterpret_cast<const float4 *>(weight + 4 * i);
                float4       r;
                r.x = v.x * rms * g.x, r.y = v.y * rms * g.y, r.z = v.z * rms * g.z, r.w = v.w * rms * g.w;
                *reinterpret_cast<float4 *>(y + 4 * i) = r;
            }
        for (int i = 4 * vec + threadIdx.x; i < dim; i += 256)
            y[i] = x[i] * rms * weight[i];
    }
}

// ---- C = A . W^T, many rows: BM x BN tiles (128 x 128 or 64 x 64) on fp32 MFMA ---------------------
// 4 waves as 2 x 2, each a (BM/2) x (BN/2) sub-tile of 32 x 32 MFMA tiles (64 accumulator VGPRs at 128 x 128).
// A and W tiles of BK contraction steps go global -> registers (two tiles ahead) -> LDS (row stride BK + 4
// floats: the lanes' ds_read_b128 fall in distinct banks), double-buffered, one barrier per step.  The
// instruction's two k lanes are re-associated over four MFMAs so a lane's operands are one float4:
// lane (j, half) of MFMA e supplies element k = 8c + 4*half + e of its row.
// Tile ids are dealt so that the row tiles of one column tile run on one XCD at about the same time
// (ids go round-robin over the 8 XCDs): the weight tile is then read from HBM once and met in that L2.
// SPLIT (few rows x few columns: the tiles alone leave most CUs idle -- 64 rows x 2048 columns are 32 of them): the
// contraction is cut into p.ksplit slices, a workgroup per (tile, slice).  Every workgroup stores its partial tile
// write-through (sc1), the one whose arrival completes the tile (agent-scope counter, the protocol of the decode
// kernel's split-K combine, nvk_kernels.hip.h) adds the slices IN SLICE ORDER -- the same bits whoever comes last --
// and stores the tile.  Nobody waits for anybody.
// MODE kNtStream (stream-K): per row tile, the (column tile, k-step) pairs laid end to end and cut into equal runs of
// p.kslice steps, p.ksplit runs to a row tile, a workgroup per run -- every CU gets the same number of MFMA steps
// whatever the number of tiles (64 rows x 11264 columns are 176 tiles: 1.4 per CU cut in two, 0.7 whole).  A run covers
// the tail of one tile, whole tiles, and the head of another; a tile that several runs share is completed as in
// kNtSplit, its pieces added in run order.  The cuts of a column tile are the same in every row tile, so an output's
// summation order still does not depend on its row.  A workgroup's piece of a tile lies in slot 2 g (the tile its run
// starts in) or 2 g + 1 (a later one) of p.part.
enum { kNtPlain = 0, kNtSplit = 1, kNtStream = 2 };
// WM: waves along the rows (2: the 2 x 2 layout above; 1: four waves side by side over a 32-row tile, for 17-32 rows).
template <int BM, int BN, int BK, int MODE = kNtPlain, int WM = 2>
__global__ void __launch_bounds__(256, 2) nvk_gemm_nt_kernel(const LinearParams p)
{
    constexpr bool SPLIT = MODE == kNtSplit, STREAM = MODE == kNtStream;
    constexpr int WN = 4 / WM, SM = BM / WM, SN = BN / WN; // a wave's sub-tile: SM rows x SN columns
    constexpr int TA = SM / 32, TB = SN / 32;              // 32 x 32 MFMA tiles per wave along the rows / the columns
    static_assert((WM == 1 || WM == 2) && SM % 32 == 0 && SN % 32 == 0 && TA >= 1 && TB >= 1, "wave layout");
    constexpr int kGemmBK = BK, kGemmLds = BK + 4, KV = BK / 32; // KV: float4 per thread per tile row pass
    extern __shared__ __attribute__((aligned(16))) float s_mem[]; // [2][BM + BN][kGemmLds]
    __shared__ int s_last;

    // kNtStream: this workgroup's run [pos, pos_end) of its row tile's k-steps, nk_tile of them to a tile
    const int nk_tile = (p.k + BK - 1) / BK;
    int       pos = 0, pos_end = 0, run_mt = 0, run_g = 0;
    if (STREAM) {
        run_mt  = blockIdx.x / p.ksplit;
        run_g   = blockIdx.x - run_mt * p.ksplit;
        pos     = run_g * p.kslice;
        pos_end = min(pos + p.kslice, p.n_tiles * nk_tile);
        if (pos >= pos_end)
            return;
    }
  for (;;) { // (kNtPlain / kNtSplit: once)
    int bid = blockIdx.x, slice = 0, nt, mt, kbeg, kend, run_steps = 0;
    if (STREAM) {
        nt          = pos / nk_tile;
        mt          = run_mt;
        const int a = pos - nt * nk_tile;
        run_steps   = min(nk_tile - a, pos_end - pos);
        kbeg = a * BK, kend = min(p.k, (a + run_steps) * BK);
    }
    else {
        if (SPLIT) { // slices outermost: the workgroups of one slice are dealt as an unsplit launch's
            const int per = gridDim.x / p.ksplit;
            slice         = bid / per;
            bid -= slice * per;
        }
        const int span = 8 * p.m_tiles;
        const int rem  = bid % span;
        nt             = bid / span * 8 + rem % 8;
        mt             = rem / 8;
        if (nt >= p.n_tiles)
            return; // (padding of the grid: no slice of such a tile exists, nobody counts on it)
        kbeg = SPLIT ? slice * p.kslice : 0, kend = SPLIT ? min(p.k, kbeg + p.kslice) : p.k;
    }
    const LinearSeg sg = segment_of(p, nt);
    const int       n0 = (nt - sg.tile0) * BN, m0 = mt * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int lr = tid >> 3, lc = (tid & 7) * 4; // this thread's (row, first k) inside a 32-row pass of a tile

    const float *arow[BM / 32], *wrow[BN / 32];
#pragma unroll
    for (int i = 0; i < BM / 32; i++)
        arow[i] = p.a + (int64_t)min(m0 + lr + 32 * i, p.m - 1) * p.lda + lc; // rows past the edge: computed, not stored
#pragma unroll
    for (int i = 0; i < BN / 32; i++)
        wrow[i] = sg.w + (int64_t)min(n0 + lr + 32 * i, sg.n - 1) * p.k + lc;

    // two tiles ahead in registers (HBM latency is longer than one tile's MFMAs when a SIMD has one wave)
    struct Regs
    {
        float4 a[BM / 32][KV], b[BN / 32][KV];
    };
    Regs r0, r1;
    auto fetch = [&](Regs &r, int k0) {
#pragma unroll
        for (int u = 0; u < KV; u++) {
            const bool inside = k0 + 32 * u + lc < kend; // k % 4 == 0: a float4 is wholly inside or outside
#pragma unroll
            for (int i = 0; i < BM / 32; i++)
                r.a[i][u] = inside ? *reinterpret_cast<const float4 *>(arow[i] + k0 + 32 * u)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < BN / 32; i++)
                r.b[i][u] = inside ? *reinterpret_cast<const float4 *>(wrow[i] + k0 + 32 * u)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](const Regs &r, int buf) {
        float *sa = s_mem + buf * (BM + BN) * kGemmLds, *sb = sa + BM * kGemmLds;
#pragma unroll
        for (int u = 0; u < KV; u++) {
#pragma unroll
            for (int i = 0; i < BM / 32; i++)
                *reinterpret_cast<float4 *>(sa + (lr + 32 * i) * kGemmLds + 32 * u + lc) = r.a[i][u];
#pragma unroll
            for (int i = 0; i < BN / 32; i++)
                *reinterpret_cast<float4 *>(sb + (lr + 32 * i) * kGemmLds + 32 * u + lc) = r.b[i][u];
        }
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; a++)
#pragma unroll
        for (int b = 0; b < TB; b++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                acc[a][b][r] = 0.f;

#define NVK_MFMA4(A, B, ACC)                                                                                           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, ACC, 0, 0, 0);                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, ACC, 0, 0, 0);                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, ACC, 0, 0, 0);                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, ACC, 0, 0, 0);
    auto multiply = [&](int buf) {
        const float *sa = s_mem + buf * (BM + BN) * kGemmLds + (wm * SM + j) * kGemmLds + 4 * half;
        const float *sb = s_mem + buf * (BM + BN) * kGemmLds + (BM + wn * SN + j) * kGemmLds + 4 * half;
        // operands of chunk c + 1 are read from LDS while chunk c is multiplied
        float4 av[2][TA], bv[2][TB];
        auto   operands = [&](int slot, int c) {
#pragma unroll
            for (int a = 0; a < TA; a++)
                av[slot][a] = *reinterpret_cast<const float4 *>(sa + 32 * a * kGemmLds + 8 * c);
#pragma unroll
            for (int b = 0; b < TB; b++)
                bv[slot][b] = *reinterpret_cast<const float4 *>(sb + 32 * b * kGemmLds + 8 * c);
        };
        operands(0, 0);
#pragma unroll
        for (int c = 0; c < kGemmBK / 8; c++) {
            if (c + 1 < kGemmBK / 8)
                operands((c + 1) & 1, c + 1);
#pragma unroll
            for (int a = 0; a < TA; a++)
#pragma unroll
                for (int b = 0; b < TB; b++) {
                    NVK_MFMA4(av[c & 1][a], bv[c & 1][b], acc[a][b])
                }
        }
    };

    const int nk  = (kend - kbeg + kGemmBK - 1) / kGemmBK;
    // the walk over k starts at a tile that depends on the column tile (see nvk_gemm_skinny_kernel: HBM channels)
    const int rot = NVK_K_ROTATE ? (int)(((unsigned)nt * 37u) % (unsigned)nk) : 0;
    auto      k_at = [&](int kt) { return kbeg + ((kt + rot) % nk) * kGemmBK; };
    // tile kt sits in LDS buffer kt & 1; at the top of step kt the registers r[(kt + 1) & 1] hold (or await)
    // tile kt + 1 and r[kt & 1] is free for tile kt + 2
    fetch(r0, k_at(0));
    if (nk > 1)
        fetch(r1, k_at(1));
    stash(r0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 2 < nk)
            fetch(r0, k_at(kt + 2));
        multiply(0);
        if (kt + 1 < nk)
            stash(r1, 1); // the other buffer: every wave left it before the previous barrier
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 3 < nk)
                fetch(r1, k_at(kt + 3));
            multiply(1);
            if (kt + 2 < nk)
                stash(r0, 0);
            __syncthreads();
        }
    }

    // a tile in pieces (kNtSplit: always; kNtStream: unless this run holds all of the tile's steps)
    const bool pieces = SPLIT || (STREAM && run_steps != nk_tile);
    bool       store  = true;
    if (pieces) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        constexpr int        kSc1  = 16; // aux bit 4 of the buffer instructions on gfx950: write-through
        constexpr int        kTile = BM * BN;
        const int            tile  = mt * p.n_tiles + nt;
        // kNtStream: the runs g_first .. g_last of this row tile share the tile
        const int g_first = STREAM ? (nt * nk_tile) / p.kslice : 0;
        const int n_piece = STREAM ? ((nt + 1) * nk_tile - 1) / p.kslice - g_first + 1 : p.ksplit;
        auto      slot_of = [&](int i) { // the i-th piece of the tile, in the order of the sum
            if (!STREAM)
                return (int64_t)tile * p.ksplit + i;
            const int g = g_first + i;
            return (int64_t)2 * (mt * p.ksplit + g) + (nt != (g * p.kslice) / nk_tile ? 1 : 0);
        };
        float *mine = p.part + slot_of(STREAM ? run_g - g_first : slice) * kTile;
        // a thread's accumulators as float4s, [float4 index][thread]: coalesced both ways
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, kTile * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < TA; a++)
#pragma unroll
            for (int b = 0; b < TB; b++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 v = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                           ((((a * TB + b) * 4 + q) * 256) + tid) * 16, 0, kSc1);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int      *cnt  = p.arrive + tile;
            const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == n_piece - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // zero again for the next launch
            }
            s_last = last;
        }
        __syncthreads();
        store = s_last != 0;
        if (!store && !STREAM)
            return;
        if (store) {
#pragma unroll
            for (int a = 0; a < TA; a++)
#pragma unroll
                for (int b = 0; b < TB; b++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        acc[a][b][r] = 0.f;
            for (int sl = 0; sl < n_piece; sl++) {
                const float4 *src = reinterpret_cast<const float4 *>(p.part + slot_of(sl) * kTile) + tid;
#pragma unroll
                for (int a = 0; a < TA; a++)
#pragma unroll
                    for (int b = 0; b < TB; b++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float4 v = src[((a * TB + b) * 4 + q) * 256];
                            acc[a][b][4 * q] += v.x, acc[a][b][4 * q + 1] += v.y, acc[a][b][4 * q + 2] += v.z,
                                acc[a][b][4 * q + 3] += v.w;
                        }
            }
        }
    }
    // D[row][col]: col = lane & 31, register r holds row (r & 3) + 8 * (r >> 2) + 4 * half
    if (store) {
#pragma unroll
        for (int a = 0; a < TA; a++)
#pragma unroll
            for (int b = 0; b < TB; b++) {
                const int col = n0 + wn * SN + 32 * b + j;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + wm * SM + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < p.m && col < sg.n)
                        sg.out[(int64_t)row * sg.ldo + col] = acc[a][b][r];
                }
            }
    }
    if (!STREAM)
        return;
    pos += run_steps;
    if (pos >= pos_end)
        return;
  }
}
// ---- C = A . W^T, few rows: stream the weights once -----------------------------------------------
// One workgroup = a 32 x 32 output tile (32 weight rows), 8 waves, the contraction dealt round-robin
// over the waves in steps of 64.  Every wave is on its own: it brings its 32 x 64 weight tile and the
// matching activation tile global -> registers -> its private LDS region -> MFMA operands; the weight
// tiles of the next TWO steps are in flight while the current one is multiplied (HBM latency is longer
// than one step's 32 MFMAs), the activations (L2-resident) one step ahead.  No barrier until the waves'
// partial tiles meet in LDS at the end.  Rows past m are computed on clamped addresses and dropped.
// With more than 32 rows the row tiles of one column tile are dealt to one XCD (as in the tiled kernel).
constexpr int kSkinnyWaves = 8;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// T = 32: 32 x 32 tiles on v_mfma_f32_32x32x2_f32.  T = 16 (at most 16 rows): 16 x 16 tiles on
// v_mfma_f32_16x16x4_f32 -- half the MFMA cycles spent on padding rows, twice the workgroups, half the LDS.
// TM > 1: one wave multiplies TM row tiles by the same weight tile, so up to T * TM rows still stream the weights
// once; BK is then 32 so that the waves' private LDS regions fit (two workgroups per CU).  <16, 2, 32> is the form for
// 17-32 rows: two 16-row tiles per weight tile instead of a second workgroup per row tile that re-reads the weights
// through L2 or a 32-row tile padded to 32 columns; measured no faster (NVK_SKINNY16_TM2, profiles/r03_qkv_kernel_stats_tm2.csv): off.
// Four consecutive columns (col % 4 == 0) of row `row` of the q | k | v product, segment si = 0 / 1 / 2: what
// nvk_rope_append_kernel does with them -- q and k rotated in their pairs by the row's position (positional.hpp:13-39),
// q to `q_out`, k and v to the slot of token `pos` in the row's sequence (model.hpp:190-207).
__device__ __forceinline__ void rope_append_store(const RopeAppendTarget &t, int si, int row, int col, float4 sum, float *q_out)
{
    const int pos = t.positions[row];
    if (pos < 0) // (the row is skipped, as nvk_rope_append_kernel does)
        return;
    float4 o = sum;
    if (si < 2) {
        const int   pr = (col % t.head_dim) >> 1, pr1 = ((col + 2) % t.head_dim) >> 1;
        const float a0 = (float)pos * t.inv_freq[pr], a1 = (float)pos * t.inv_freq[pr1];
        double      s0, c0, s1, c1;
        sincos((double)a0, &s0, &c0);
        sincos((double)a1, &s1, &c1);
        const float2 p0 = rope_rotate(make_float2(sum.x, sum.y), make_float2((float)c0, (float)s0));
        const float2 p1 = rope_rotate(make_float2(sum.z, sum.w), make_float2((float)c1, (float)s1));
        o               = make_float4(p0.x, p0.y, p1.x, p1.y);
    }
    if (si == 0) {
        *reinterpret_cast<float4 *>(q_out) = o;
        return;
    }
    const int     blk = token_block(pos, t.bs_magic);
    const int     off = token_offset(pos, blk, t.block_size);
    const int     pb  = clamp_block(t.block_tables[(int64_t)row * t.max_blocks + blk], t.num_blocks);
    const int64_t at  = ((int64_t)pb * t.block_size + off) * t.row_len + col;
    *reinterpret_cast<float4 *>((si == 1 ? t.k_layer : t.v_layer) + at) = o;
}

// EPI / SCALE (T = 16, TM = 1: up to 16 rows): the epilogues of nvk_gemv_kernel for decode steps of a few requests --
// kSkinnyResid and SCALE are the two ends of a NormLink, kSkinnyRope is RoPE + KV append (see nvk_gemv_kernel and
// nvk_internal.h).  SwiGLU has no form here: an 8 + 8 row tile doubles the MFMA work per weight byte and measured slower
// than the two products followed by the element-wise kernel.
enum { kSkinnyPlain = 0, kSkinnyResid = 1, kSkinnyRope = 2 };

// SPLIT (T = 16, TM = 1, plain): few column tiles (2048 columns are 128 of them, half the CUs) -- the contraction is cut
// into p.ksplit slices, a workgroup per (tile, slice); the slices' 16 x 16 partial tiles meet as in
// nvk_gemm_nt_kernel<.., SPLIT> (write-through stores, an arrival counter per tile, added in slice order by whoever
// comes last; only the workgroup's first wave is involved).
template <int T, int TM, int BK, int EPI = kSkinnyPlain, bool SCALE = false, bool SPLIT = false>
__global__ void __launch_bounds__(64 * kSkinnyWaves) nvk_gemm_skinny_kernel(const LinearParams p)
{
    static_assert((EPI == kSkinnyPlain && !SCALE) || (T == 16 && TM == 1 && BK == 64)
                      || (EPI == kSkinnyRope && !SCALE && T == 32 && TM == 1 && BK == 64),
                  "epilogues: the 16 x 16 tile kernel; RoPE + append also on 32 x 32 tiles");
    static_assert(!SPLIT || (T == 16 && TM == 1 && EPI == kSkinnyPlain && !SCALE), "split contraction: the plain 16 x 16 tile kernel");
    __shared__ float s_ssq[16]; // SCALE: every row's sum of squares
    constexpr int LDS = BK + 4;             // row stride in LDS (floats)
    constexpr int LPR = BK / 4, RPI = 64 / LPR; // lanes per tile row, rows per load instruction
    constexpr int RW = T / RPI, RX = TM * T / RPI; // float4 per lane per step: weights, activations
    extern __shared__ __attribute__((aligned(16))) float s_mem[]; // [waves][(TM + 1) * T][LDS]
    int bid = blockIdx.x, slice = 0;
    if (SPLIT) { // slices outermost
        const int per = gridDim.x / p.ksplit;
        slice         = bid / per;
        bid -= slice * per;
    }
    const int span = 8 * p.m_tiles;
    const int rem  = bid % span;
    const int nt   = bid / span * 8 + rem % 8;
    const int m0   = rem / 8 * (TM * T);
    if (nt >= p.n_tiles)
        return;
    const int kbeg = SPLIT ? slice * p.kslice : 0, kend = SPLIT ? min(p.k, kbeg + p.kslice) : p.k;
    const LinearSeg sg = segment_of(p, nt);
    const int       n0 = (nt - sg.tile0) * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & (T - 1), part = lane / T; // MFMA operand row / column, and which k lanes of the instruction
    const int lr = lane / LPR, lc = (lane % LPR) * 4;

    constexpr int kRegion = (TM + 1) * T * LDS;
    float        *sa = s_mem + wave * kRegion, *sb = sa + TM * T * LDS;
    int           aoff[RX], woff[RW]; // float offsets of this lane's rows (the host checked they fit an int)
#pragma unroll
    for (int i = 0; i < RX; i++)
        aoff[i] = min(m0 + lr + RPI * i, p.m - 1) * p.lda + lc;
#pragma unroll
    for (int i = 0; i < RW; i++)
        woff[i] = min(n0 + lr + RPI * i, sg.n - 1) * p.k + lc;
    float4 rx[RX], rw0[RW], rw1[RW];
    auto   fetch_w = [&](float4(&r)[RW], int k0) {
        const bool inside = k0 + lc < kend;
#pragma unroll
        for (int i = 0; i < RW; i++)
            r[i] = inside ? load_stream(sg.w + woff[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto fetch_x = [&](int k0) {
        const bool inside = k0 + lc < kend;
#pragma unroll
        for (int i = 0; i < RX; i++)
            rx[i] = inside ? *reinterpret_cast<const float4 *>(p.a + aoff[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](const float4(&r)[RW]) {
#pragma unroll
        for (int i = 0; i < RX; i++)
            *reinterpret_cast<float4 *>(sa + (lr + RPI * i) * LDS + lc) = rx[i];
#pragma unroll
        for (int i = 0; i < RW; i++)
            *reinterpret_cast<float4 *>(sb + (lr + RPI * i) * LDS + lc) = r[i];
    };

    constexpr int NACC = T == 32 ? 16 : 4; // accumulator registers of one tile
    f32x16 acc32[TM];
    f32x4  acc16[TM];
#pragma unroll
    for (int t = 0; t < TM; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc32[t][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++)
            acc16[t][r] = 0.f;
    }
    // lane (j, part) of MFMA e supplies element 4 * part + e of the chunk (8 wide at T = 32, 16 wide at T = 16)
    const float *pa = sa + j * LDS + 4 * part, *pb = sb + j * LDS + 4 * part;
    auto         multiply = [&]() {
        constexpr int CH = T == 32 ? 8 : 16;
#pragma unroll
        for (int c = 0; c < BK / CH; c++) {
            const float4 b0 = *reinterpret_cast<const float4 *>(pb + CH * c);
            if constexpr (T == 32) {
#pragma unroll
                for (int t = 0; t < TM; t++) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(pa + t * T * LDS + CH * c);
                    NVK_MFMA4(a0, b0, acc32[t])
                }
            }
            else {
#pragma unroll
                for (int t = 0; t < TM; t++) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(pa + t * T * LDS + CH * c);
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc16[t], 0, 0, 0);
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc16[t], 0, 0, 0);
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc16[t], 0, 0, 0);
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc16[t], 0, 0, 0);
                }
            }
        }
    };

    const int nk    = (kend - kbeg + BK - 1) / BK;
    const int steps = nk > wave ? (nk - wave + kSkinnyWaves - 1) / kSkinnyWaves : 0; // this wave's share
    // Every workgroup starts the contraction somewhere else (a function of its column tile only, so an output's
    // summation order does not depend on its row): rows of W are in_dim * 4 bytes apart, and workgroups walking k
    // in lockstep would all be asking the same few HBM channels for the same 256-byte column of every row.
    const int rot   = NVK_K_ROTATE ? (int)(((unsigned)nt * 37u) % (unsigned)nk) : 0;
    auto      k_of  = [&](int s) { return kbeg + ((wave + s * kSkinnyWaves + rot) % nk) * BK; };
    if (steps > 0) {
        fetch_w(rw0, k_of(0));
        fetch_x(k_of(0));
    }
    if (steps > 1)
        fetch_w(rw1, k_of(1));
    // small operands of the epilogue, asked for now so that they are here when the product is
    // (SCALE: the rows' sums of squares go to LDS right away -- registers held across the main loop would cost the second
    //  workgroup of the CU.  32 threads a row, each a few of the producer's per-tile partial sums.)
    float4 old4 = make_float4(0.f, 0.f, 0.f, 0.f), gain4 = old4;
    if (SCALE) {
        const int row = tid >> 5, sub = tid & 31; // 512 threads: rows 0 .. 15
        float     v   = 0.f;
        if (row < p.m)
            for (int i = sub; i < p.link.ss_n; i += 32)
                v += p.link.ss_in[row * p.link.ss_n + i];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
            v += __shfl_xor(v, o);
        if (sub == 0)
            s_ssq[row] = v;
    }
    if (EPI == kSkinnyResid && tid < 64) {
        const int l4 = (tid & 15) * 4, row = m0 + 4 * (l4 / T) + (tid >> 4), col = n0 + (l4 & (T - 1));
        if (row < p.m && col + 3 < sg.n) {
            old4  = *reinterpret_cast<const float4 *>(sg.out + (int64_t)row * sg.ldo + col);
            gain4 = *reinterpret_cast<const float4 *>(p.link.gain_out + col);
        }
    }
    for (int s = 0; s < steps; s += 2) {
        stash(rw0);
        if (s + 2 < steps)
            fetch_w(rw0, k_of(s + 2));
        if (s + 1 < steps)
            fetch_x(k_of(s + 1));
        multiply();
        if (s + 1 < steps) {
            stash(rw1);
            if (s + 3 < steps)
                fetch_w(rw1, k_of(s + 3));
            if (s + 2 < steps)
                fetch_x(k_of(s + 2));
            multiply();
        }
    }
#undef NVK_MFMA4

    // the waves' partial tiles meet in LDS (each wave's own region, TM x NACC x 64 floats); then NACC x 16 threads
    // add the eight partials of four outputs each, in wave order.  D[row][col]: col = lane % T; register r of lane
    // part p holds row (r & 3) + 8 * (r >> 2) + 4 * p at T = 32, row 4 * p + r at T = 16.
    static_assert(TM * NACC * 64 <= kRegion, "the reduction does not fit the wave's LDS region");
#pragma unroll
    for (int t = 0; t < TM; t++)
#pragma unroll
        for (int r = 0; r < NACC; r++)
            sa[(t * NACC + r) * 64 + lane] = T == 32 ? acc32[t][r] : acc16[t][r & 3]; // own region: operand reads are behind us
    __syncthreads();
    if (T == 16 && (EPI != kSkinnyPlain || SCALE)) { // T = 16, TM = 1: thread (r, l4) owns row 4 * (l4 / 16) + r, columns (l4 % 16) .. + 3
        if (tid < 64) {
            const int r = tid >> 4, l4 = (tid & 15) * 4, cg = l4 & 15;
            const int row = m0 + 4 * (l4 / T) + r;
            auto      total = [&](int at) {
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < kSkinnyWaves; w++) {
                    const float4 v = *reinterpret_cast<const float4 *>(s_mem + w * kRegion + r * 64 + at);
                    sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
                }
                return sum;
            };
            float4 sum = total(l4);
            float  rms = 1.f;
            if (SCALE) { // normalization.hpp:15-22 over the row the producer summed
                rms = 1.0f / sqrtf(s_ssq[min(row, 15)] / p.k + p.link.eps);
                sum.x *= rms, sum.y *= rms, sum.z *= rms, sum.w *= rms;
            }
            const int  col   = n0 + cg;
            const bool whole = row < p.m && col + 3 < sg.n; // (the host checked n % 4 == 0: a float4 is in or out)
            if (EPI == kSkinnyPlain) {
                if (whole)
                    *reinterpret_cast<float4 *>(sg.out + (int64_t)row * sg.ldo + col) = sum;
            }
            else if (EPI == kSkinnyResid) {
                if (whole) { // model.hpp:227-228 / 238-239, and the next norm's operands
                    const float4 xn = make_float4(old4.x + sum.x, old4.y + sum.y, old4.z + sum.z, old4.w + sum.w);
                    *reinterpret_cast<float4 *>(sg.out + (int64_t)row * sg.ldo + col) = xn;
                    *reinterpret_cast<float4 *>(p.link.xg_out + (int64_t)row * sg.ldo + col) =
                        make_float4(xn.x * gain4.x, xn.y * gain4.y, xn.z * gain4.z, xn.w * gain4.w);
                }
                // the 16-column tile's share of the row's next sum of squares: the four threads of a row are neighbours
                float v = whole ? fmaf(old4.w + sum.w, old4.w + sum.w, fmaf(old4.z + sum.z, old4.z + sum.z,
                                       fmaf(old4.y + sum.y, old4.y + sum.y, (old4.x + sum.x) * (old4.x + sum.x)))) : 0.f;
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                if (cg == 0 && row < p.m && col < sg.n)
                    p.link.ss_out[row * p.link.ss_out_n + col / 16] = v;
            }
            else if (whole) { // kSkinnyRope
                const int si = (p.nseg > 1 && nt >= p.seg[1].tile0) + (p.nseg > 2 && nt >= p.seg[2].tile0);
                rope_append_store(p.rope, si, row, col, sum, sg.out + (int64_t)row * sg.ldo + col);
            }
        }
        return;
    }
    if (tid < NACC * 16) {
        const int r = tid >> 4, l4 = (tid & 15) * 4; // register r, lanes l4 .. l4 + 3
        const int pt = l4 / T;                        // the four lanes share a part
        const int col = n0 + (l4 & (T - 1));
#pragma unroll
        for (int t = 0; t < TM; t++) {
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < kSkinnyWaves; w++) {
                const float4 v = *reinterpret_cast<const float4 *>(s_mem + w * kRegion + (t * NACC + r) * 64 + l4);
                sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
            }
            const int row = m0 + t * T + (T == 32 ? (r & 3) + 8 * (r >> 2) + 4 * pt : 4 * pt + r);
            if constexpr (SPLIT) { // (tid < 64: the first wave alone, all its lanes)
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                constexpr int        kSc1 = 16; // aux bit 4 of the buffer instructions on gfx950: write-through
                const int            tile = (rem / 8) * p.n_tiles + nt;
                float               *all  = p.part + (int64_t)tile * p.ksplit * 256;
                const __amdgpu_buffer_rsrc_t rs =
                    __builtin_amdgcn_make_buffer_rsrc(all + slice * 256, 0, 256 * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rs, tid * 16, 0, kSc1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int last = 0;
                if (tid == 0) {
                    int      *cnt  = p.arrive + tile;
                    const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last           = prev == p.ksplit - 1;
                    if (last) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // zero again for the next launch
                    }
                }
                last = __builtin_amdgcn_readfirstlane(last);
                if (!last)
                    return;
                sum = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int sl = 0; sl < p.ksplit; sl++) {
                    const float4 v = reinterpret_cast<const float4 *>(all + sl * 256)[tid];
                    sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
                }
            }
            if (EPI == kSkinnyRope) { // (T = 32; the host checked n % 4 == 0: a float4 is in or out)
                if (row < p.m && col + 3 < sg.n) {
                    const int si = (p.nseg > 1 && nt >= p.seg[1].tile0) + (p.nseg > 2 && nt >= p.seg[2].tile0);
                    rope_append_store(p.rope, si, row, col, sum, sg.out + (int64_t)row * sg.ldo + col);
                }
            }
            else if (row < p.m) {
                float *o = sg.out + (int64_t)row * sg.ldo + col;
                if (col + 3 < sg.n && (sg.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(sg.out) & 15) == 0))
                    *reinterpret_cast<float4 *>(o) = sum;
                else {
                    if (col < sg.n) o[0] = sum.x;
                    if (col + 1 < sg.n) o[1] = sum.y;
                    if (col + 2 < sg.n) o[2] = sum.z;
                    if (col + 3 < sg.n) o[3] = sum.w;
                }
            }
        }
    }
}

// ---- the same for up to 16 rows per row tile, with the output columns dealt EVENLY over the CUs ---------------
// A launch of the kernel above takes as long as its busiest CU: 6144 columns in 16-column tiles are 384 workgroups,
// 1.5 per CU, and run as long as 8192 columns do (23.3 against 25.2 us; 4096 columns: 13.3).  Here a workgroup takes
// `cw` columns (a multiple of 4, at most 16 TN; p.cw, chosen by the host so that the launch is whole rounds of one
// workgroup per CU) as TN MFMA tiles of 16 that share each activation tile; weight rows past the workgroup's columns
// and activation rows past m are not loaded; four weight tiles in flight per wave (two at TN > 1).
// Otherwise nvk_gemm_skinny_kernel<16, 1, 64>: 8 waves deal the contraction in steps of 64, each through its own LDS
// region, partial tiles meet in LDS.  EPI: plain or RoPE + append.
// What this kernel is bound by (profiles/r03_skinny_balance.log): 3.3 us + 1.2 us per 512 of k per workgroup, whatever
// the rows (3 or 16), the columns per workgroup (8 or 16) and the tiles in flight (2 or 4) -- a wave-step is 8 loads,
// 16 LDS instructions and 16 MFMAs, each about 0.4 us per round of the CU's eight waves, one after the other within a
// wave.  The same contraction with the MFMA operands loaded straight from memory in the operand's layout (16 rows x
// 64 B per instruction, no LDS) was built, is bit-for-bit as good against the checker, and is SLOWER at 16 rows
// (wo 8.1 -> 10.2 us, down 16.5 -> 20.9): twice the cache lines per load instruction.  Not kept.
template <int TN, int EPI>
__global__ void __launch_bounds__(64 * kSkinnyWaves) nvk_gemm_skinny16_kernel(const LinearParams p)
{
    static_assert(EPI == kSkinnyPlain || EPI == kSkinnyRope, "epilogues here: none, or RoPE + append");
    constexpr int T = 16, BK = 64, LDS = BK + 4, LPR = BK / 4, RPI = 64 / LPR; // 16 lanes a tile row, 4 rows an instruction
    constexpr int RX = T / RPI, RW = TN * T / RPI;                             // float4 per lane per step
    extern __shared__ __attribute__((aligned(16))) float s_mem[];              // [waves][(1 + TN) * 16][LDS]
    const int span = 8 * p.m_tiles;
    const int rem  = blockIdx.x % span;
    const int nt   = blockIdx.x / span * 8 + rem % 8;
    const int m0   = rem / 8 * T;
    if (nt >= p.n_tiles)
        return;
    const LinearSeg sg = segment_of(p, nt);
    const int       n0 = (nt - sg.tile0) * p.cw, n1 = min(n0 + p.cw, sg.n); // this workgroup's columns [n0, n1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, part = lane >> 4;
    const int lr = lane / LPR, lc = (lane % LPR) * 4;

    constexpr int kRegion = (1 + TN) * T * LDS;
    float        *sa = s_mem + wave * kRegion, *sb = sa + T * LDS;
    int           aoff[RX], woff[RW];
    bool          wlive[RW], xlive[RX];
#pragma unroll
    for (int i = 0; i < RX; i++) {
        xlive[i] = m0 + lr + RPI * i < p.m;
        aoff[i]  = min(m0 + lr + RPI * i, p.m - 1) * p.lda + lc;
    }
#pragma unroll
    for (int i = 0; i < RW; i++) {
        wlive[i] = n0 + lr + RPI * i < n1;
        woff[i]  = min(n0 + lr + RPI * i, sg.n - 1) * p.k + lc;
    }
    constexpr int D = TN == 1 ? 4 : 2; // weight tiles in flight per wave: 16 KiB (24 at TN 3)
    float4 rx[2][RX], rw[D][RW];
    auto   fetch_w = [&](float4(&r)[RW], int k0) {
        const bool inside = k0 + lc < p.k;
#pragma unroll
        for (int i = 0; i < RW; i++)
            r[i] = inside && wlive[i] ? load_stream(sg.w + woff[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto fetch_x = [&](float4(&r)[RX], int k0) {
        const bool inside = k0 + lc < p.k;
#pragma unroll
        for (int i = 0; i < RX; i++)
            r[i] = inside && xlive[i] ? *reinterpret_cast<const float4 *>(p.a + aoff[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](const float4(&x)[RX], const float4(&r)[RW]) {
#pragma unroll
        for (int i = 0; i < RX; i++)
            *reinterpret_cast<float4 *>(sa + (lr + RPI * i) * LDS + lc) = x[i];
#pragma unroll
        for (int i = 0; i < RW; i++)
            *reinterpret_cast<float4 *>(sb + (lr + RPI * i) * LDS + lc) = r[i];
    };
    f32x4 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; t++)
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *pa = sa + j * LDS + 4 * part, *pb = sb + j * LDS + 4 * part;
    auto         multiply = [&]() {
#pragma unroll
        for (int c = 0; c < BK / 16; c++) {
            const float4 a0 = *reinterpret_cast<const float4 *>(pa + 16 * c);
#pragma unroll
            for (int t = 0; t < TN; t++) {
                const float4 b0 = *reinterpret_cast<const float4 *>(pb + t * T * LDS + 16 * c);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[t], 0, 0, 0);
            }
        }
    };
    const int nk    = (p.k + BK - 1) / BK;
    const int steps = nk > wave ? (nk - wave + kSkinnyWaves - 1) / kSkinnyWaves : 0;
    // (the walk over k starts at a tile that depends on the column tile only: see nvk_gemm_skinny_kernel)
    const int rot   = NVK_K_ROTATE ? (int)(((unsigned)nt * 37u) % (unsigned)nk) : 0;
    auto      k_of  = [&](int s) { return ((wave + s * kSkinnyWaves + rot) % nk) * BK; };
#pragma unroll
    for (int d = 0; d < D; d++)
        if (d < steps)
            fetch_w(rw[d], k_of(d));
#pragma unroll
    for (int d = 0; d < 2; d++)
        if (d < steps)
            fetch_x(rx[d], k_of(d));
    for (int s = 0; s < steps; s += D) {
#pragma unroll
        for (int d = 0; d < D; d++) // (D is even: step s + d uses activation buffer d & 1)
            if (s + d < steps) {
                stash(rx[d & 1], rw[d]);
                if (s + d + D < steps)
                    fetch_w(rw[d], k_of(s + d + D));
                if (s + d + 2 < steps)
                    fetch_x(rx[d & 1], k_of(s + d + 2));
                multiply();
            }
    }
    // the waves' partial tiles meet in LDS (own regions), then TN x 64 threads add the eight partials of four outputs
    // each, in wave order: D[row][col], col = lane % 16, register r of lane part p holds row 4 p + r
    static_assert(TN * 4 * 64 <= kRegion, "the reduction does not fit the wave's LDS region");
#pragma unroll
    for (int t = 0; t < TN; t++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            sa[(t * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    if (tid < TN * 64) {
        const int t = tid >> 6, r = (tid >> 4) & 3, l4 = (tid & 15) * 4;
        const int row = m0 + 4 * (l4 / T) + r, col = n0 + t * T + (l4 & (T - 1));
        float4    sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < kSkinnyWaves; w++) {
            const float4 v = *reinterpret_cast<const float4 *>(s_mem + w * kRegion + (t * 4 + r) * 64 + l4);
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        if (row < p.m && col < n1) {
            float *o = sg.out + (int64_t)row * sg.ldo + col;
            if (EPI == kSkinnyRope) { // (the host checked n % 4 == 0 and cw % 4 == 0: a float4 is in or out)
                const int si = (p.nseg > 1 && nt >= p.seg[1].tile0) + (p.nseg > 2 && nt >= p.seg[2].tile0);
                rope_append_store(p.rope, si, row, col, sum, o);
            }
            else if (col + 3 < n1 && (sg.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(sg.out) & 15) == 0))
                *reinterpret_cast<float4 *>(o) = sum;
            else {
                o[0] = sum.x;
                if (col + 1 < n1) o[1] = sum.y;
                if (col + 2 < n1) o[2] = sum.z;
                if (col + 3 < n1) o[3] = sum.w;
            }
        }
    }
}

// ---- C = A . W^T, one or two rows: the reference's own case (one token: Ops::matmul is a GEMV) -------
// HBM-bound: every weight byte is read once and multiplied on the vector ALUs (an MFMA tile padded to 32
// rows would cost as much time as the stream itself).  One workgroup = kGemvCols output columns (weight rows: 4),
// 8 waves dealing the contraction round-robin in steps of 64; a lane holds a float4 of kGemvCols / 4 weight rows
// per step (256 contiguous bytes per row and instruction), two steps in flight, and the matching float4 of every
// activation row (L2-resident).  Per lane partial sums -> DPP row sum over the 16 lanes of a weight row -> the eight
// waves meet in LDS.  An output's summation order depends on k only.
// GLU (kGemvCols = 4, two segments of equal n: w_gate, w_up): the workgroup's four weight rows are gate rows n0, n0 + 1
// and the same two rows of up, and it writes silu(gate) * up to the first segment's output (Ops::swiglu,
// activation.hpp:34-40) -- one launch less per layer where a launch costs as much as the work.  Each 16-lane group
// walks k in the order the plain kernel uses for its column, so the sums (and, expf being the same function, the
// products) have the bits of nvk_batch_matmul2_device followed by the separate SwiGLU kernel.
constexpr int kGemvWaves = 8, kGemvBK = 64;

// ROPE (kGemvCols = 4, the three segments wq, wk, wv): the epilogue is nvk_rope_append_kernel -- q and k columns are
// rotated in their pairs (both columns of a pair are in the workgroup's tile) by the row's position, q goes to the
// first segment's output, k and v straight to the row's slot in the paged cache.  Same expressions, same bits.
//
// The norm between two GEMVs without a kernel of its own (NormLink, nvk_internal.h):
// RESID (a plain 4-column tile; the producer: wo, w_down): out = out + product (the residual add), and beside it
// xg = out * gain (the next norm's gain, not yet its 1 / rms) and ss[tile] = the sum of squares of the tile's four new
// values.  SCALE (the consumer: the input rows are such an xg): the product is multiplied by 1 / rms, rms from the
// ss[] the producer left -- summed here by the whole workgroup in a fixed order while the weights stream.
// out = rms * sum_k w[k] (x[k] g[k]) instead of sum_k w[k] ((x[k] rms) g[k]): a few ulp from the separate norm.
template <int M, int kGemvCols, int DEPTH, bool GLU = false, bool ROPE = false, bool SCALE = false, bool RESID = false>
__global__ void __launch_bounds__(64 * kGemvWaves) nvk_gemv_kernel(const LinearParams p)
{
    static_assert(!GLU || kGemvCols == 4, "GLU pairs two gate rows with two up rows");
    static_assert(!ROPE || (kGemvCols == 4 && !GLU), "ROPE needs a rotation pair inside the tile");
    static_assert(!RESID || (kGemvCols == 4 && !GLU && !ROPE && !SCALE && M * 4 <= 64), "RESID: a plain 4-column tile");
    __shared__ float s_red[kGemvWaves][M][kGemvCols];
    __shared__ float s_ss[kGemvWaves][M];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane >> 4, lc = (lane & 15) * 4;
    // GLU: lane groups 0, 1 carry gate rows n0, n0 + 1, groups 2, 3 the same rows of up; `nt` is the tile the plain
    // kernel would give the group's column (it sets where the walk over k starts)
    const LinearSeg sg = GLU ? p.seg[lr >> 1] : segment_of(p, blockIdx.x);
    const int       n0 = GLU ? (int)blockIdx.x * 2 : ((int)blockIdx.x - sg.tile0) * kGemvCols;
    const int       nt = GLU ? sg.tile0 + n0 / kGemvCols : (int)blockIdx.x;

    constexpr int R = kGemvCols / 4; // weight rows per lane
    int woff[R], xoff[M];
#pragma unroll
    for (int i = 0; i < R; i++)
        woff[i] = min(n0 + (GLU ? lr & 1 : lr + 4 * i), sg.n - 1) * p.k + lc;
#pragma unroll
    for (int m = 0; m < M; m++)
        xoff[m] = min(m, p.m - 1) * p.lda + lc;

    struct Step
    {
        float4 w[R], x[M];
    };
    Step ring[DEPTH]; // DEPTH steps of loads in flight (indices are compile-time after unrolling)
    auto fetch = [&](Step &t, int k0) {
        const bool inside = k0 + lc < p.k;
#pragma unroll
        for (int i = 0; i < R; i++)
            t.w[i] = inside ? load_stream(sg.w + woff[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < M; m++)
            t.x[m] = inside ? *reinterpret_cast<const float4 *>(p.a + xoff[m] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float acc[M][R];
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int i = 0; i < R; i++)
            acc[m][i] = 0.f;
    auto consume = [&](const Step &t) {
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int i = 0; i < R; i++) {
                acc[m][i] = fmaf(t.w[i].x, t.x[m].x, acc[m][i]);
                acc[m][i] = fmaf(t.w[i].y, t.x[m].y, acc[m][i]);
                acc[m][i] = fmaf(t.w[i].z, t.x[m].z, acc[m][i]);
                acc[m][i] = fmaf(t.w[i].w, t.x[m].w, acc[m][i]);
            }
    };

    const int nk    = (p.k + kGemvBK - 1) / kGemvBK;
    const int steps = nk > wave ? (nk - wave + kGemvWaves - 1) / kGemvWaves : 0;
    const int rot   = NVK_K_ROTATE ? (int)(((unsigned)nt * 37u) % (unsigned)nk) : 0;
    auto      k_of  = [&](int s) { return ((wave + s * kGemvWaves + rot) % nk) * kGemvBK; };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < steps)
            fetch(ring[d], k_of(d));
    // small operands of the epilogue, asked for now so that they are here when the product is
    float ssq[M], old = 0.f, gain_n = 0.f;
#pragma unroll
    for (int m = 0; m < M; m++) {
        ssq[m] = 0.f;
        if (SCALE && m < p.m)
            for (int i = tid; i < p.link.ss_n; i += 64 * kGemvWaves)
                ssq[m] += p.link.ss_in[m * p.link.ss_n + i];
    }
    if (RESID && tid < M * kGemvCols) {
        const int m = tid / kGemvCols, n = n0 + tid % kGemvCols;
        if (m < p.m && n < sg.n)
            old = sg.out[(int64_t)m * sg.ldo + n], gain_n = p.link.gain_out[n];
    }
    for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (s + d < steps) {
                consume(ring[d]);
                if (s + d + DEPTH < steps)
                    fetch(ring[d], k_of(s + d + DEPTH));
            }
    }

#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int i = 0; i < R; i++) {
            float v = acc[m][i]; // sum over the 16 lanes that share weight row lr + 4i
            v += __shfl_xor(v, 1), v += __shfl_xor(v, 2), v += __shfl_xor(v, 4), v += __shfl_xor(v, 8);
            if ((lane & 15) == 0)
                s_red[wave][m][lr + 4 * i] = v;
        }
    if (SCALE) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            float v = ssq[m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                v += __shfl_xor(v, o);
            if (lane == 0)
                s_ss[wave][m] = v;
        }
    }
    __syncthreads();
    // SCALE: 1 / rms of row m (normalization.hpp:15-22 over the row the producer summed)
    auto inv_rms = [&](int m) {
        float sq = 0.f;
#pragma unroll
        for (int w = 0; w < kGemvWaves; w++)
            sq += s_ss[w][m];
        return 1.0f / sqrtf(sq / p.k + p.link.eps);
    };
    if (GLU) {
        if (tid < M * 2) {
            const int m = tid / 2, c = tid % 2;
            float     gate = 0.f, up = 0.f;
#pragma unroll
            for (int w = 0; w < kGemvWaves; w++)
                gate += s_red[w][m][c], up += s_red[w][m][c + 2];
            if (SCALE) {
                const float r = inv_rms(m);
                gate *= r, up *= r;
            }
            if (m < p.m && n0 + c < p.seg[0].n) {
#pragma clang fp contract(off) // val = g / (1 + expf(-g)); val * up -- two roundings, as the separate kernel
                const float val = gate / (1.0f + expf(-gate));
                p.seg[0].out[(int64_t)m * p.seg[0].ldo + n0 + c] = val * up;
            }
        }
    }
    else if (ROPE) {
        if (tid < M * kGemvCols) {
            const int m = tid / kGemvCols, c = tid % kGemvCols, n = n0 + c;
            float     sum = 0.f, mate = 0.f; // this column and the other one of its pair
#pragma unroll
            for (int w = 0; w < kGemvWaves; w++)
                sum += s_red[w][m][c], mate += s_red[w][m][c ^ 1];
            if (SCALE) {
                const float r = inv_rms(m);
                sum *= r, mate *= r;
            }
            const RopeAppendTarget &t  = p.rope;
            const int               si = (p.nseg > 1 && nt >= p.seg[1].tile0) + (p.nseg > 2 && nt >= p.seg[2].tile0);
            const int               pos = m < p.m ? t.positions[m] : -1;
            if (pos >= 0 && n < sg.n) { // (a negative position: the row is skipped, as nvk_rope_append_kernel does)
                float val = sum;
                if (si < 2) { // q, k: positional.hpp:13-39 as nvk_rope_append_kernel evaluates it
                    const float angle = (float)pos * t.inv_freq[(n % t.head_dim) >> 1];
                    double      sn, cn;
                    sincos((double)angle, &sn, &cn);
                    const float2 r = rope_rotate(c & 1 ? make_float2(mate, sum) : make_float2(sum, mate),
                                                 make_float2((float)cn, (float)sn));
                    val            = c & 1 ? r.y : r.x;
                }
                if (si == 0)
                    sg.out[(int64_t)m * sg.ldo + n] = val;
                else { // model.hpp:190-207: the slot of token `pos` in the row's sequence
                    const int     blk = token_block(pos, t.bs_magic);
                    const int     off = token_offset(pos, blk, t.block_size);
                    const int     pb  = clamp_block(t.block_tables[(int64_t)m * t.max_blocks + blk], t.num_blocks);
                    const int64_t at  = ((int64_t)pb * t.block_size + off) * t.row_len + n;
                    (si == 1 ? t.k_layer : t.v_layer)[at] = val;
                }
            }
        }
    }
    else if (RESID) {
        if (tid < 64) { // the writers are threads 0 .. 4 M - 1; the whole first wave takes part in the shuffles
            const int  m = min(tid / kGemvCols, M - 1), c = tid % kGemvCols;
            const bool mine = tid < M * kGemvCols && m < p.m && n0 + c < sg.n;
            float      sum = 0.f;
#pragma unroll
            for (int w = 0; w < kGemvWaves; w++)
                sum += s_red[w][m][c];
            const float xn = mine ? old + sum : 0.f; // model.hpp:227-228 / 238-239
            if (mine) {
                sg.out[(int64_t)m * sg.ldo + n0 + c]         = xn;
                p.link.xg_out[(int64_t)m * sg.ldo + n0 + c] = xn * gain_n;
            }
            // the tile's share of the next norm's sum of squares, in column order
            const int   base = tid - c;
            const float x0 = __shfl(xn, base), x1 = __shfl(xn, base + 1), x2 = __shfl(xn, base + 2), x3 = __shfl(xn, base + 3);
            if (c == 0 && tid < M * kGemvCols && m < p.m)
                p.link.ss_out[m * p.link.ss_out_n + nt] = fmaf(x3, x3, fmaf(x2, x2, fmaf(x1, x1, x0 * x0)));
        }
    }
    else if (tid < M * kGemvCols) {
        const int m = tid / kGemvCols, c = tid % kGemvCols;
        float     sum = 0.f;
#pragma unroll
        for (int w = 0; w < kGemvWaves; w++)
            sum += s_red[w][m][c];
        if (SCALE)
            sum *= inv_rms(m);
        if (m < p.m && n0 + c < sg.n)
            sg.out[(int64_t)m * sg.ldo + n0 + c] = sum;
    }
}

// hb = silu(hb) * up, silu(g) = g / (1 + expf(-g))  (Ops::swiglu, include/ops/activation.hpp:34-40)
__global__ void nvk_swiglu_kernel(float *__restrict__ hb, const float *__restrict__ up, int64_t n)
{
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g   = hb[i];
        const float val = g / (1.0f + expf(-g));
        hb[i]           = val * up[i];
    }
}

} // namespace nvk

namespace {

int check_device(int device, const char *who)
{
    const int ndev = nvk::device_count_cached();
    if (ndev <= 0)
        return nvk::fail(NVK_ERR_NO_DEVICE, "%s: no HIP device visible (this library has no CPU path)", who);
    if (device < 0 || device >= ndev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: device %d outside [0,%d)", who, device, ndev);
    return NVK_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// One launch for up to three (weight, out, n) segments sharing the activations `in` [batch, in_dim].
std::mutex g_first_use; // the per-device caches below are filled on first use, possibly from several threads

int cu_count_of(int device)
{
    static int                  cu_count[64]; // per device, asked once
    std::lock_guard<std::mutex> lock(g_first_use);
    if (cu_count[device & 63] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0)
            n = 256;
        cu_count[device & 63] = n;
    }
    return cu_count[device & 63];
}

// more than 64 KiB of dynamic LDS needs the attribute, once per device and kernel
int opt_in_lds(const void *kernel, size_t lds_bytes, int device)
{
    static const void          *done[64][32];
    std::lock_guard<std::mutex> lock(g_first_use);
    const void                **row = done[device & 63];
    for (int i = 0; i < 32; i++) {
        if (row[i] == kernel)
            return NVK_OK;
        if (!row[i]) {
            HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            row[i] = kernel;
            return NVK_OK;
        }
    }
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); // table full: every time
    return NVK_OK;
}

// Where the split-K launches of one stream keep their partial tiles and arrival counters: one buffer per (device,
// stream), allocated the first time the stream needs it and NEVER moved or freed afterwards -- a captured graph holds
// its address.  Launches on one stream run one after the other, so they can share it; two streams never do.  Its size
// is what the largest cut launch can need (at most two workgroups per CU, two 64 x 64 pieces each: 16 MiB on 256 CUs).  Allocating is not
// possible while the stream is being captured: the caller then takes the uncut path (an engine's first step of a
// shape runs uncaptured, which is when this is made).
struct SplitWorkspace
{
    float *part    = nullptr;
    int   *arrive  = nullptr;
    size_t floats  = 0, counters = 0;
};
constexpr size_t kSplitCounters = 16384; // tiles a cut launch may have (an arrival counter each)
std::mutex g_split_mutex;
struct SplitKey
{
    int         device;
    hipStream_t stream;
    bool        operator<(const SplitKey &o) const { return device != o.device ? device < o.device : stream < o.stream; }
};
std::map<SplitKey, SplitWorkspace> g_split_ws;

// returns false when the stream has no buffer yet and none can be made now (stream capture in progress, out of memory),
// or the request is larger than any cut launch is (cannot happen: see split_slices)
bool split_workspace(int device, int cus, hipStream_t st, size_t floats, size_t counters, SplitWorkspace *out)
{
    std::lock_guard<std::mutex> lock(g_split_mutex);
    SplitWorkspace &w = g_split_ws[SplitKey{device, st}];
    if (!w.part) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return false;
        }
        // (stream-K: two pieces per workgroup, two workgroups per CU; twice that for the 128 x 128 form when an A/B build turns it on)
        const size_t nf = (size_t)(NVK_NT_STREAM128_MIN_ROWS < 100000 || NVK_LINEAR_ENV_ON ? 8 : 4) * cus * 64 * 64, nc = kSplitCounters;
        float       *pf = nullptr;
        int         *pc = nullptr;
        if (hipMalloc(&pf, nf * sizeof(float)) != hipSuccess || hipMalloc(&pc, nc * sizeof(int)) != hipSuccess
            || hipMemset(pc, 0, nc * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            if (pf)
                (void)hipFree(pf);
            if (pc)
                (void)hipFree(pc);
            return false;
        }
        w.part = pf, w.arrive = pc, w.floats = nf, w.counters = nc;
    }
    if (w.floats < floats || w.counters < counters)
        return false;
    *out = w;
    return true;
}

// Slices of the contraction for `tiles` 64 x 64 output tiles on `cus` CUs (two workgroups fit a CU): as many as fill
// the chip about twice, each at least NVK_NT_SPLIT_MIN_K long and a whole number of 64-steps; 1 = not worth cutting.
int split_slices(long long tiles, int k, int cus, int *kslice)
{
    *kslice = k;
    if (tiles <= 0 || tiles >= cus)
        return 1;
    int fill = NVK_NT_SPLIT_FILL;
#ifdef NVK_LINEAR_ENV // A/B build
    if (const char *e = getenv("NVK_NT_SPLIT_FILL"))
        fill = atoi(e);
#endif
    int want = (int)std::min<long long>(16, (long long)fill * cus / tiles); // (no more workgroups than are resident at once)
    want     = std::min(want, std::max(1, k / NVK_NT_SPLIT_MIN_K));
    if (want <= 1)
        return 1;
    const int steps = (k + 63) / 64, per = (steps + want - 1) / want;
    *kslice         = per * 64;
    return (steps + per - 1) / per; // (every slice has at least one step)
}

// 16 x 16 tiles in the weight-streaming MFMA kernel: always up to 16 rows; with more rows (several row tiles per
// column tile, the weights' further readings L2 hits) while 32-column tiles would leave CUs idle -- measured
// (profiles/r01_qkv_projection.log): 2048 columns x 32 rows 14.6 -> 7.5 us, 5632 -> 2048 at 32 / 64 rows
// 36.5 -> 16.5 / 37.6 -> 31.2 us
bool tile16(int batch, const int *n, int nseg, int cus)
{
    long long t32 = 0;
    for (int s = 0; s < nseg; s++)
        t32 += (n[s] + 31) / 32;
    return batch <= NVK_SKINNY16_MAX_ROWS || (NVK_SKINNY16_WIDE && batch <= 32 && t32 < cus)
           || (NVK_SKINNY16_WIDE && batch <= 64 && t32 <= cus / 4);
}

// Output columns per workgroup of nvk_gemm_skinny16_kernel (a multiple of 4 in [8, 48]): the fewest rounds of one
// workgroup per CU that the columns fit in, and within that round count the narrowest workgroups -- a launch takes as
// long as its busiest CU (see the kernel), so every CU gets the same share wherever the output is wide enough.
int balanced_cols(int batch, const int *n, int nseg, int cus)
{
    const int m_tiles = (batch + 15) / 16;
    auto      wgs_at  = [&](int cw) {
        long long t = 0;
        for (int s = 0; s < nseg; s++)
            t += (n[s] + cw - 1) / cw;
        return t * m_tiles;
    };
    for (int rounds = 1; rounds <= 64; rounds++)
        for (int cw = 8; cw <= 48; cw += 4)
            if (wgs_at(cw) <= (long long)rounds * cus)
                return cw;
    return 48;
}

// the weight-streaming kernel's tile (16 or 32) and row tiles per wave for `batch` rows
void skinny_choice(int batch, const int *n, int nseg, int cus, int *t, int *tm, bool epilogue = false)
{
    *t = tile16(batch, n, nseg, cus) ? 16 : 32;
    long long tiles = 0;
    for (int s = 0; s < nseg; s++)
        tiles += (n[s] + *t - 1) / *t;
    // (several row tiles per wave only when the column tiles alone occupy the chip: with few columns, separate
    //  workgroups per row tile are twice as many workgroups and faster -- 2560 columns x 64 rows: 15.9 us against 24.6)
    *tm = (*t == 32 && NVK_SKINNY_TM && batch > 32 && tiles >= 128) ? (batch + 31) / 32 : 1;
    // 17-32 rows on 16 x 16 tiles: both row tiles in one wave wherever the column tiles alone occupy the chip
    // (not under a fused epilogue: RoPE + append wants a workgroup per row tile)
    if (*t == 16 && NVK_SKINNY16_TM2 && !epilogue && batch > 16 && batch <= 32 && tiles >= cus)
        *tm = 2;
}

int launch_linear(const float *in, int batch, int in_dim, const float *const *w, float *const *out, const int *n,
                  int nseg, int device, hipStream_t st, const char *who, bool glu = false,
                  const nvk::RopeAppendTarget *rope = nullptr, const nvk::NormLink *link = nullptr)
{
    if (in_dim % 4 != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: in_dim=%d must be a multiple of 4", who, in_dim);
    if (!aligned16(in))
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: the activations must be 16-byte aligned", who);
    if ((long long)batch * in_dim > 0x7fffffffLL)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: %d x %d activations exceed 2^31 floats", who, batch, in_dim);
    for (int s = 0; s < nseg; s++)
        if ((long long)n[s] * in_dim > 0x7fffffffLL)
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: a %d x %d weight exceeds 2^31 floats", who, n[s], in_dim);
    long long cols = 0;
    for (int s = 0; s < nseg; s++)
        cols += n[s];
    // (a norm link exists only in the GEMV: two linked rows stay there at any width)
    const bool gemv   = batch <= NVK_GEMV_MAX_ROWS && (batch < 2 || cols <= NVK_GEMV2_MAX_COLS || link);
    const int  cus = cu_count_of(device);
    // a plain product of more rows than the weight-streaming kernel takes whose 64 x 64 tiles would leave CUs idle: the
    // tiled kernel with the contraction cut -- 128 rows of 5632 -> 2048: 118.7 -> 33.9 us; at 33-96 rows it is within 10 %
    // of the weight-streaming kernel either way and stays off
    // (measured against the weight-streaming kernel and the unsplit tiles: profiles/r03_nt_splitk.log)
    int       split = 1, kslice = in_dim;
    long long t64   = 0;
    for (int s = 0; s < nseg; s++)
        t64 += (n[s] + 63) / 64;
    t64 *= (batch + 63) / 64;
    int split_min_rows = NVK_NT_SPLIT_MIN_ROWS;
#ifdef NVK_LINEAR_ENV // A/B build
    if (const char *e = getenv("NVK_NT_SPLIT_MIN_ROWS"))
        split_min_rows = atoi(e);
#endif
    nvk::LinearParams p;
    p.ksplit = 1, p.kslice = in_dim, p.part = nullptr, p.arrive = nullptr;
    if (!glu && !rope && !link && batch >= split_min_rows && batch > NVK_GEMV_MAX_ROWS && NVK_GEMM_FORCE_TILE != 128) {
        split = split_slices(t64, in_dim, cus, &kslice);
        SplitWorkspace ws;
        if (split > 1 && split_workspace(device, cus, st, (size_t)t64 * split * 64 * 64, (size_t)t64, &ws))
            p.ksplit = split, p.kslice = kslice, p.part = ws.part, p.arrive = ws.arrive;
        else
            split = 1;
    }
    // ... or, for more rows than the weight-streaming kernel does well, the (tile, k-step) pairs dealt evenly over the CUs
    int  stream_steps = 0; // k-steps per workgroup of nvk_gemm_nt_kernel<64, 64, 64, kNtStream>; 0: not that kernel
    bool stream32     = false; // ... of nvk_gemm_nt_kernel<32, 128, 32, kNtStream, 1>
    bool stream128    = false; // ... of nvk_gemm_nt_kernel<128, 128, 32, kNtStream>
    {
        int stream_min_rows = NVK_NT_STREAM_MIN_ROWS, stream_fill = NVK_NT_STREAM_FILL, stream_min_steps = 16;
        bool stream_always = false;
#ifdef NVK_LINEAR_ENV // A/B build
        if (const char *e = getenv("NVK_NT_STREAM_MIN_STEPS")) // (0: wherever the kernel can run -- the tests' way in)
            stream_min_steps = atoi(e), stream_always = stream_min_steps == 0;
        if (const char *e = getenv("NVK_NT_STREAM_MIN_ROWS"))
            stream_min_rows = atoi(e);
        if (const char *e = getenv("NVK_NT_STREAM_FILL"))
            stream_fill = atoi(e);
#endif
        long long n64 = 0;
        for (int s = 0; s < nseg; s++)
            n64 += (n[s] + 63) / 64;
        const int       m64       = (batch + 63) / 64;
        const long long row_steps = n64 * ((in_dim + 63) / 64); // k-steps of one row tile
        // where it pays (profiles/r03_nt_streamk.log): at least 16 steps to a CU, and the kernel it replaces would leave
        // a tenth of the chip idle or more -- plain tiles always do below 3 per CU, the cut tiles where their slices do
        // not fill the slots
        const bool cut_fills = split > 1 && t64 * split * 10 >= 2LL * cus * 9;
        // (one row tile with 8-16 steps to a CU over at least 32 column tiles: one workgroup per CU -- 5632 -> 2048 at 64 rows
        //  29.7 -> 25.9 us, 2048 -> 4096 27.0 -> 20.2; with fewer column tiles too many runs share a tile: 8192 -> 1024 loses)
        if (m64 == 1 && n64 >= 32 && row_steps >= 8LL * cus && row_steps < 16LL * cus && stream_min_steps == 16)
            stream_min_steps = 8, stream_fill = 1;
        if (!glu && !rope && !link && batch >= stream_min_rows && batch > NVK_GEMV_MAX_ROWS && NVK_GEMM_FORCE_TILE != 128
            && (t64 < (long long)NVK_NT_STREAM_MAX_TILES_PER_CU * cus || stream_always) && t64 <= (long long)kSplitCounters
            && row_steps * m64 >= (long long)stream_min_steps * cus && row_steps < (1LL << 24) && (!cut_fills || stream_always)) {
            long long per_row = std::max<long long>(1, (long long)stream_fill * cus / m64); // runs to a row tile
            if (m64 > 1)
                per_row = std::max<long long>(8, per_row / 8 * 8); // (the row tiles' runs over one k-range on one XCD)
            per_row = std::min(per_row, row_steps);
            const long long per = (row_steps + per_row - 1) / per_row; // steps to a run
            per_row             = (row_steps + per - 1) / per;
            SplitWorkspace ws;
            if (per >= 2 && split_workspace(device, cus, st, (size_t)2 * per_row * m64 * 64 * 64, (size_t)t64, &ws)) {
                stream_steps = (int)per;
                split        = 1;
                p.ksplit = (int)per_row, p.kslice = stream_steps, p.part = ws.part, p.arrive = ws.arrive;
            }
        }
        // 17-32 rows: the same over 32 x 128 tiles in 32-steps (four waves side by side), where the weight-streaming kernel
        // would run two 16-row tiles per weight tile (4096 -> 6144: 22 us at 16 rows, 33.6 at 17-32): 2048 -> 5632 x 2 at 32
        // rows 34.7 -> 26.9 us, 4096 -> 6144 33.6 -> 29.2; smaller products lose (5632 -> 2048: 16.8 -> 23.2) and stay
        int stream32_min_steps = NVK_NT_STREAM32_MIN_STEPS;
#ifdef NVK_LINEAR_ENV // A/B build
        if (const char *e = getenv("NVK_NT_STREAM32_MIN_STEPS"))
            stream32_min_steps = atoi(e);
#endif
        long long n128 = 0;
        for (int s = 0; s < nseg; s++)
            n128 += (n[s] + 127) / 128;
        const long long row_steps32 = n128 * ((in_dim + 31) / 32);
        if (stream_steps == 0 && !glu && !rope && !link && batch > 16 && batch <= 32 && batch > NVK_GEMV_MAX_ROWS
            && stream32_min_steps > 0 && row_steps32 >= (long long)stream32_min_steps * cus && row_steps32 < (1LL << 24)
            && n128 <= (long long)kSplitCounters) {
            long long       per_row = std::min<long long>((long long)stream_fill * cus, row_steps32);
            const long long per     = (row_steps32 + per_row - 1) / per_row;
            per_row                 = (row_steps32 + per - 1) / per;
            SplitWorkspace ws;
            if (per >= 2 && split_workspace(device, cus, st, (size_t)2 * per_row * 32 * 128, (size_t)n128, &ws)) {
                stream_steps = (int)per;
                stream32     = true;
                split        = 1;
                p.ksplit = (int)per_row, p.kslice = stream_steps, p.part = ws.part, p.arrive = ws.arrive;
            }
        }
        // many rows, yet fewer 128 x 128 tiles than fill the chip a few times over (512 rows x 6144 columns: 192, 0.75 to a
        // CU; as 64 x 64 tiles 768, three rounds of which the last is half empty): stream-K over the 128 x 128 tiles.
        // Built, parity green, NOT on: 2048 -> 5632 x 2 at 384 rows 207 -> 189 us and 4096 -> 6144 at 768 rows 394 -> 373, but
        // 4096 -> 6144 at 512 rows 240 -> 229 on one box and 244 -> 247 on the next, nothing at 1024 rows
        // (profiles/r03_nt_streamk.log) -- not worth 16 MiB more per stream
        int stream128_min_rows = NVK_NT_STREAM128_MIN_ROWS;
#ifdef NVK_LINEAR_ENV // A/B build
        if (const char *e = getenv("NVK_NT_STREAM128_MIN_ROWS"))
            stream128_min_rows = atoi(e);
#endif
        long long       c128 = 0; // (n128 above: column tiles of 128)
        for (int s = 0; s < nseg; s++)
            c128 += (n[s] + 127) / 128;
        const int       m128        = (batch + 127) / 128;
        const long long row_steps128 = c128 * ((in_dim + 31) / 32);
        if (stream_steps == 0 && !glu && !rope && !link && batch >= stream128_min_rows && NVK_GEMM_FORCE_TILE != 64
            && c128 * m128 < 6LL * cus && c128 * m128 <= (long long)kSplitCounters && row_steps128 * m128 >= 32LL * cus
            && row_steps128 < (1LL << 24)) {
            long long per_row = std::max<long long>(1, (long long)NVK_NT_STREAM128_FILL * cus / m128);
            if (m128 > 1)
                per_row = std::max<long long>(8, per_row / 8 * 8);
            per_row = std::min(per_row, row_steps128);
            const long long per = (row_steps128 + per_row - 1) / per_row;
            per_row             = (row_steps128 + per - 1) / per;
            SplitWorkspace ws;
            if (per >= 2 && split_workspace(device, cus, st, (size_t)2 * per_row * m128 * 128 * 128, (size_t)(c128 * m128), &ws)) {
                stream_steps = (int)per;
                stream128    = true;
                split        = 1;
                p.ksplit = (int)per_row, p.kslice = stream_steps, p.part = ws.part, p.arrive = ws.arrive;
            }
        }
    }
    const bool skinny = batch <= NVK_SKINNY_MAX_ROWS && split == 1 && stream_steps == 0;
    p.a = in, p.m = batch, p.k = in_dim, p.lda = in_dim;
    const bool t16 = tile16(batch, n, nseg, cus);
    // fused epilogues beyond the GEMV: the norm link in the 16 x 16 tile kernel up to 16 rows; RoPE + append wherever the
    // weight-streaming kernel gives every row tile a workgroup of its own; SwiGLU only in the GEMV
    int sk_t = 32, sk_tm = 1;
    skinny_choice(batch, n, nseg, cus, &sk_t, &sk_tm, rope != nullptr || link != nullptr);
    if (!gemv && (glu || (link && !(batch <= NVK_SKINNY16_MAX_ROWS && NVK_SKINNY16_MAX_ROWS <= 16))
                  || (rope && !(skinny && sk_tm == 1))))
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: internal: no fused epilogue for %d rows", who, batch);
    if ((rope || link) && !gemv) {
        for (int s = 0; s < nseg; s++)
            if (n[s] % 4 != 0)
                return nvk::fail(NVK_ERR_INVALID_ARG, "%s: internal: fused epilogues need out_dim %% 4 == 0", who);
        if (rope && rope->head_dim % 4 != 0)
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: internal: the RoPE epilogue needs head_dim %% 4 == 0", who);
    }
    if (rope && !(((gemv && NVK_GEMV_COLS == 4) || !gemv) && nseg == 3 && !glu))
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: internal: the RoPE epilogue needs the GEMV and q, k, v segments", who);
    if (rope)
        p.rope = *rope;
    const bool scale = link && link->ss_in, resid = link && link->ss_out;
    const bool fused16 = !gemv && batch <= NVK_SKINNY16_MAX_ROWS && NVK_SKINNY16_MAX_ROWS <= 16; // the 16-row MFMA kernel
    if (link) {
        if (!((gemv && NVK_GEMV_COLS == 4) || fused16) || (scale && resid) || (resid && (glu || rope || nseg != 1))
            || (scale && link->ss_n <= 0) || (resid && (!link->gain_out || !link->xg_out)) || !(link->eps >= 0.f))
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: internal: a norm link needs the GEMV (scale in, or residual out)", who);
        p.link = *link;
    }
    if (glu && !(gemv && NVK_GEMV_COLS == 4 && nseg == 2 && n[0] == n[1] && n[0] > 0))
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: internal: the GLU form needs the GEMV and two equal segments", who);
    // tiled kernel: 128 x 128 tiles when they fill the chip about twice over, else 64 x 64 (four times the workgroups)
    int bm = 128, bn = 128;
    if (!gemv && !skinny) {
        long long t128 = 0;
        for (int s = 0; s < nseg; s++)
            t128 += (n[s] + 127) / 128;
        t128 *= (batch + 127) / 128;
        // measured (profiles/r01_qkv_projection.log): two 128 x 128 workgroups fit a CU, and a launch that is not
        // at least three full rounds of them loses more to the ragged last round than 64 x 64 tiles lose to their
        // four times higher LDS and L2 traffic per flop
        if (split > 1 || stream_steps > 0 || NVK_GEMM_FORCE_TILE == 64 || (NVK_GEMM_FORCE_TILE == 0 && t128 < 6LL * cus))
            bm = bn = 64;
        if (stream32)
            bm = 32, bn = 128;
        if (stream128)
            bm = bn = 128;
    }
    else
        bn = gemv ? (batch <= 2 ? NVK_GEMV_COLS : 32) : (t16 ? 16 : 32);
    // 16-row tiles without a norm link: the columns dealt evenly over the CUs (nvk_gemm_skinny16_kernel), `cw` to a workgroup
    // ... only where 16-column tiles leave the chip visibly unbalanced (the busiest CU 1.25 x the mean or more: 6144
    // columns are 1.5 workgroups per CU and run as long as 8192).  Elsewhere the 16-column kernel is as fast or faster
    // (TinyLlama's 2048-wide shapes, whole model step: 1.72 against 1.82 ms at 4 requests, profiles/r03_skinny_balance.log)
    bool balanced = NVK_SKINNY16_BALANCED && !gemv && skinny && t16 && sk_tm == 1 && !link;
    {
        long long t16s = 0;
        for (int s = 0; s < nseg; s++)
            t16s += (n[s] + 15) / 16;
        t16s *= (batch + 15) / 16;
        const long long rounds = (t16s + cus - 1) / cus;
        balanced = balanced && t16s > cus && rounds * cus * 100 >= t16s * (100 + NVK_SKINNY16_BALANCE_PCT);
    }
    for (int s = 0; s < nseg; s++)
        balanced = balanced && n[s] % 4 == 0; // (its workgroups own whole float4s of a row)
    if (balanced)
        bn = balanced_cols(batch, n, nseg, cus);
    long long tiles = 0;
    p.nseg = 0;
    for (int s = 0; s < nseg; s++) {
        if (n[s] == 0)
            continue;
        if (!aligned16(w[s]))
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: weight %d must be 16-byte aligned", who, s);
        nvk::LinearSeg &sg = p.seg[p.nseg++];
        sg.w = w[s], sg.out = out[s], sg.n = n[s], sg.ldo = n[s], sg.tile0 = (int)tiles;
        tiles += (n[s] + bn - 1) / bn;
    }
    if (p.nseg == 0 || batch == 0)
        return NVK_OK;
    p.n_tiles = (int)tiles;
    if (gemv) {
        p.m_tiles = 1;
        const dim3 grid((unsigned)(glu ? (n[0] + 1) / 2 : tiles)), block(64 * nvk::kGemvWaves);
#define NVK_GEMV_LAUNCH(M_, COLS_, ...) \
    hipLaunchKernelGGL((nvk::nvk_gemv_kernel<M_, COLS_, NVK_GEMV_DEPTH, __VA_ARGS__>), grid, block, 0, st, p)
#define NVK_GEMV_ROWS(COLS_, ...)                    \
    do {                                             \
        if (batch == 1)                              \
            NVK_GEMV_LAUNCH(1, COLS_, __VA_ARGS__);  \
        else                                         \
            NVK_GEMV_LAUNCH(2, COLS_, __VA_ARGS__);  \
    } while (0)
        //                 GLU    ROPE   SCALE  RESID
        if (glu && scale)
            NVK_GEMV_ROWS(4, true, false, true, false);
        else if (glu)
            NVK_GEMV_ROWS(4, true, false, false, false);
        else if (rope && scale)
            NVK_GEMV_ROWS(4, false, true, true, false);
        else if (rope)
            NVK_GEMV_ROWS(4, false, true, false, false);
        else if (resid)
            NVK_GEMV_ROWS(4, false, false, false, true);
        else if (scale)
            NVK_GEMV_ROWS(4, false, false, true, false);
        else
            NVK_GEMV_ROWS(NVK_GEMV_COLS, false, false, false, false);
#undef NVK_GEMV_ROWS
#undef NVK_GEMV_LAUNCH
    }
    else if (skinny) {
        // 16 x 16 tiles up to 16 rows; 32 x 32 up to 32; above, one wave carries 2 or 3 row tiles of 32 (NVK_SKINNY_TM)
        const int t  = balanced ? 16 : bn; // (balanced: bn is the workgroup's column count, the row tile stays 16)
        const int tm = sk_tm;
        const int bk = (tm > 1 || (t == 16 && NVK_SKINNY16_BK == 32 && !rope && !link)) ? 32 : 64;
        p.m_tiles = tm > 1 ? 1 : (batch + t - 1) / t;
        const long long wgs = p.m_tiles == 1 ? tiles : (tiles + 7) / 8 * 8 * p.m_tiles;
        if (wgs > 0x7fffffffLL)
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: %lld workgroups exceed the grid", who, wgs);
        const size_t lds = (size_t)nvk::kSkinnyWaves * (tm + 1) * t * (bk + 4) * sizeof(float);
        const dim3 block(64 * nvk::kSkinnyWaves);
        if (balanced) {
            const int tn = (bn + 15) / 16;
            p.cw         = bn;
            const size_t lds16 = (size_t)nvk::kSkinnyWaves * (1 + tn) * 16 * (64 + 4) * sizeof(float);
            const dim3   grid((unsigned)wgs);
#define NVK_SK16(TN_)                                                                                                  \
    do {                                                                                                               \
        const void *fn16 = rope ? (const void *)nvk::nvk_gemm_skinny16_kernel<TN_, nvk::kSkinnyRope>                   \
                                : (const void *)nvk::nvk_gemm_skinny16_kernel<TN_, nvk::kSkinnyPlain>;                  \
        int rc16 = opt_in_lds(fn16, lds16, device);                                                                    \
        if (rc16 != NVK_OK)                                                                                            \
            return rc16;                                                                                               \
        if (rope)                                                                                                      \
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny16_kernel<TN_, nvk::kSkinnyRope>), grid, block, lds16, st, p);     \
        else                                                                                                           \
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny16_kernel<TN_, nvk::kSkinnyPlain>), grid, block, lds16, st, p);    \
    } while (0)
            if (tn == 1)
                NVK_SK16(1);
            else if (tn == 2)
                NVK_SK16(2);
            else
                NVK_SK16(3);
#undef NVK_SK16
            HIP_TRY(hipGetLastError());
            return NVK_OK;
        }
        if (rope || link) { // the 16 x 16 tile kernel with an epilogue (checked above)
            using namespace nvk;
            const void *fused =
                rope && t == 32 ? (const void *)nvk_gemm_skinny_kernel<32, 1, 64, kSkinnyRope, false>
                : rope  ? (scale ? (const void *)nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyRope, true>
                                 : (const void *)nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyRope, false>)
                : resid ? (const void *)nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyResid, false>
                        : (const void *)nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyPlain, true>;
            int rc_lds = opt_in_lds(fused, lds, device);
            if (rc_lds != NVK_OK)
                return rc_lds;
            const dim3 grid((unsigned)wgs); // (several row tiles only with the RoPE epilogue)
            if (rope && t == 32)
                hipLaunchKernelGGL((nvk_gemm_skinny_kernel<32, 1, 64, kSkinnyRope, false>), grid, block, lds, st, p);
            else if (rope && scale)
                hipLaunchKernelGGL((nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyRope, true>), grid, block, lds, st, p);
            else if (rope)
                hipLaunchKernelGGL((nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyRope, false>), grid, block, lds, st, p);
            else if (resid)
                hipLaunchKernelGGL((nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyResid, false>), grid, block, lds, st, p);
            else
                hipLaunchKernelGGL((nvk_gemm_skinny_kernel<16, 1, 64, kSkinnyPlain, true>), grid, block, lds, st, p);
            HIP_TRY(hipGetLastError());
            return NVK_OK;
        }
        // up to 16 rows over so few column tiles that half the CUs or more would idle (2048 columns: 128 tiles): the
        // contraction cut so that every CU has a workgroup, slices of at least 2048 -- 5632 -> 2048 at 16 rows 16.5 -> 12.8 us;
        // with slices of 1024 the hand-over costs what the second half of the CUs brings (2048 -> 2048: 7.8 -> 8.4 us)
        int sk_split = 1, sk_slice = in_dim;
        if (t == 16 && tm == 1 && bk == 64 && p.m_tiles == 1 && NVK_SKINNY16_SPLIT && tiles * 2 <= cus) {
            const int want = (int)std::min<long long>(std::min<long long>(cus / tiles, in_dim / 2048), 8);
            if (want > 1) {
                const int steps = (in_dim + 63) / 64, per = (steps + want - 1) / want;
                sk_slice        = per * 64;
                sk_split        = (steps + per - 1) / per;
            }
#ifdef NVK_LINEAR_ENV // A/B build
            if (getenv("NVK_SKINNY16_SPLIT_OFF"))
                sk_split = 1;
#endif
            SplitWorkspace ws;
            if (sk_split > 1 && split_workspace(device, cus, st, (size_t)tiles * sk_split * 256, (size_t)tiles, &ws)) {
                p.ksplit = sk_split, p.kslice = sk_slice, p.part = ws.part, p.arrive = ws.arrive;
                const void *fns = (const void *)nvk::nvk_gemm_skinny_kernel<16, 1, 64, nvk::kSkinnyPlain, false, true>;
                int rc_lds = opt_in_lds(fns, lds, device);
                if (rc_lds != NVK_OK)
                    return rc_lds;
                hipLaunchKernelGGL((nvk::nvk_gemm_skinny_kernel<16, 1, 64, nvk::kSkinnyPlain, false, true>),
                                   dim3((unsigned)(wgs * sk_split)), block, lds, st, p);
                HIP_TRY(hipGetLastError());
                return NVK_OK;
            }
        }
        const void  *fn  = t == 16 ? (tm == 2 ? (const void *)nvk::nvk_gemm_skinny_kernel<16, 2, 32>
                                      : bk == 32 ? (const void *)nvk::nvk_gemm_skinny_kernel<16, 1, 32>
                                              : (const void *)nvk::nvk_gemm_skinny_kernel<16, 1, 64>)
                           : tm == 1 ? (const void *)nvk::nvk_gemm_skinny_kernel<32, 1, 64>
                           : tm == 2 ? (const void *)nvk::nvk_gemm_skinny_kernel<32, 2, 32>
                                     : (const void *)nvk::nvk_gemm_skinny_kernel<32, 3, 32>;
        int rc_lds = opt_in_lds(fn, lds, device);
        if (rc_lds != NVK_OK)
            return rc_lds;
        const dim3 grid((unsigned)wgs);
        if (t == 16 && tm == 2)
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny_kernel<16, 2, 32>), grid, block, lds, st, p);
        else if (t == 16 && bk == 32)
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny_kernel<16, 1, 32>), grid, block, lds, st, p);
        else if (t == 16)
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny_kernel<16, 1, 64>), grid, block, lds, st, p);
        else if (tm == 1)
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny_kernel<32, 1, 64>), grid, block, lds, st, p);
        else if (tm == 2)
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny_kernel<32, 2, 32>), grid, block, lds, st, p);
        else
            hipLaunchKernelGGL((nvk::nvk_gemm_skinny_kernel<32, 3, 32>), grid, block, lds, st, p);
    }
    else {
        p.m_tiles = (batch + bm - 1) / bm;
        const long long wgs = (tiles + 7) / 8 * 8 * p.m_tiles * split;
        if (wgs > 0x7fffffffLL)
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: %lld workgroups exceed the grid", who, wgs);
        const int    bk  = bm == 128 ? 32 : (batch <= 256 || split > 1 || stream_steps > 0 ? 64 : 32); // 64 x 64: fewer barriers while waves are few
        const size_t lds = (size_t)2 * (bm + bn) * (bk + 4) * sizeof(float);
        if (stream32) { // (32 x 128 tiles, 32-steps)
            const size_t    lds32 = (size_t)2 * (32 + 128) * (32 + 4) * sizeof(float);
            const long long grid  = (long long)p.ksplit * p.m_tiles;
            int rc_lds = opt_in_lds((const void *)nvk::nvk_gemm_nt_kernel<32, 128, 32, nvk::kNtStream, 1>, lds32, device);
            if (rc_lds != NVK_OK)
                return rc_lds;
            hipLaunchKernelGGL((nvk::nvk_gemm_nt_kernel<32, 128, 32, nvk::kNtStream, 1>), dim3((unsigned)grid), dim3(256), lds32, st, p);
        }
        else if (stream128) { // (128 x 128 tiles, 32-steps)
            const size_t    lds128 = (size_t)2 * (128 + 128) * (32 + 4) * sizeof(float);
            const long long grid   = (long long)p.ksplit * p.m_tiles;
            int rc_lds = opt_in_lds((const void *)nvk::nvk_gemm_nt_kernel<128, 128, 32, nvk::kNtStream>, lds128, device);
            if (rc_lds != NVK_OK)
                return rc_lds;
            hipLaunchKernelGGL((nvk::nvk_gemm_nt_kernel<128, 128, 32, nvk::kNtStream>), dim3((unsigned)grid), dim3(256), lds128, st, p);
        }
        else if (stream_steps > 0) { // (64 x 64 tiles, 64-steps)
            const long long grid = (long long)p.ksplit * p.m_tiles;
            int rc_lds = opt_in_lds((const void *)nvk::nvk_gemm_nt_kernel<64, 64, 64, nvk::kNtStream>, lds, device);
            if (rc_lds != NVK_OK)
                return rc_lds;
            hipLaunchKernelGGL((nvk::nvk_gemm_nt_kernel<64, 64, 64, nvk::kNtStream>), dim3((unsigned)grid), dim3(256), lds, st, p);
        }
        else if (split > 1) { // (64 x 64 tiles, 64-steps: the slices are whole numbers of them)
            int rc_lds = opt_in_lds((const void *)nvk::nvk_gemm_nt_kernel<64, 64, 64, nvk::kNtSplit>, lds, device);
            if (rc_lds != NVK_OK)
                return rc_lds;
            hipLaunchKernelGGL((nvk::nvk_gemm_nt_kernel<64, 64, 64, nvk::kNtSplit>), dim3((unsigned)wgs), dim3(256), lds, st, p);
        }
        else if (bm == 128) {
            int rc_lds = opt_in_lds((const void *)nvk::nvk_gemm_nt_kernel<128, 128, 32>, lds, device);
            if (rc_lds != NVK_OK)
                return rc_lds;
            hipLaunchKernelGGL((nvk::nvk_gemm_nt_kernel<128, 128, 32>), dim3((unsigned)wgs), dim3(256), lds, st, p);
        }
        else if (bk == 64) {
            int rc_lds = opt_in_lds((const void *)nvk::nvk_gemm_nt_kernel<64, 64, 64>, lds, device);
            if (rc_lds != NVK_OK)
                return rc_lds;
            hipLaunchKernelGGL((nvk::nvk_gemm_nt_kernel<64, 64, 64>), dim3((unsigned)wgs), dim3(256), lds, st, p);
        }
        else
            hipLaunchKernelGGL((nvk::nvk_gemm_nt_kernel<64, 64, 32>), dim3((unsigned)wgs), dim3(256), lds, st, p);
    }
    HIP_TRY(hipGetLastError());
    return NVK_OK;
}

} // namespace

// partial sums of squares a producer leaves per row: one per 4-column GEMV tile, or per 16-column MFMA tile
int nvk::linked_ss_n(int rows, int dim) { return rows <= NVK_GEMV_MAX_ROWS ? (dim + 3) / 4 : (dim + 15) / 16; }

bool nvk::linked_gemv_ok() { return NVK_GEMV_COLS == 4 && NVK_GEMV_MAX_ROWS == 2 && NVK_SKINNY16_MAX_ROWS == 16; }

int nvk::linked_gemv(const NormLink &link, bool glu, float *out_dev, const float *in_dev, const float *w0_dev,
                     const float *w1_dev, int rows, int in_dim, int out_dim, int device, void *stream, float *scratch_dev)
{
    if (rows < 1 || rows > kFusedRowsMax || !out_dev || !in_dev || !w0_dev || (glu && !w1_dev) || in_dim <= 0
        || out_dim <= 0 || (link.ss_out && out_dim % 4 != 0) || (rows > NVK_GEMV_MAX_ROWS && out_dim % 4 != 0)
        || (rows > NVK_GEMV_MAX_ROWS && link.ss_out && out_dim % 16 != 0))
        return nvk::fail(NVK_ERR_INVALID_ARG, "linked_gemv: bad arguments");
    int rc = check_device(device, "linked_gemv");
    if (rc != NVK_OK)
        return rc;
    NVK_SCOPE(device);
    const float *const w[2]   = {w0_dev, w1_dev};
    float *const       out[2] = {out_dev, out_dev};
    const int          n[2]   = {out_dim, out_dim};
    NormLink l = link;
    l.ss_out_n = linked_ss_n(rows, out_dim);
    if (glu && rows > NVK_GEMV_MAX_ROWS) {
        // the MFMA kernel's SwiGLU tile is 8 + 8 operand rows for 8 outputs: twice the MFMA work per weight byte, measured
        // slower than the two products in 16-column tiles followed by the element-wise kernel
        if (!scratch_dev || scratch_dev == out_dev)
            return nvk::fail(NVK_ERR_INVALID_ARG, "linked_gemv: SwiGLU over more than %d rows needs a scratch buffer",
                             NVK_GEMV_MAX_ROWS);
        float *const out2[2] = {out_dev, scratch_dev};
        rc = launch_linear(in_dev, rows, in_dim, w, out2, n, 2, device, (hipStream_t)stream, "linked_gemv", false, nullptr, &l);
        if (rc != NVK_OK)
            return rc;
        const int64_t count = (int64_t)rows * out_dim;
        hipLaunchKernelGGL(nvk::nvk_swiglu_kernel, dim3((unsigned)std::min<int64_t>(1024, (count + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, out_dev, scratch_dev, count);
        HIP_TRY(hipGetLastError());
        return NVK_OK;
    }
    return launch_linear(in_dev, rows, in_dim, w, out, n, glu ? 2 : 1, device, (hipStream_t)stream, "linked_gemv", glu,
                         nullptr, &l);
}

int nvk::qkv_rope_append(const RopeAppendTarget &t, float *q_dev, const float *x_dev, const float *wq_dev,
                         const float *wk_dev, const float *wv_dev, int rows, int dim, int q_dim, int device, void *stream,
                         const NormLink *link)
{
    if (!qkv_rope_append_ok(rows, dim, q_dim, t.row_len, device) || dim <= 0 || q_dim <= 0
        || !q_dev || !x_dev || !wq_dev || !wk_dev || !wv_dev || t.head_dim <= 0 || t.head_dim % 2 != 0
        || q_dim % t.head_dim != 0 || t.row_len % t.head_dim != 0 || !t.k_layer || !t.v_layer || !t.block_tables
        || !t.positions || !t.inv_freq)
        return nvk::fail(NVK_ERR_INVALID_ARG, "qkv_rope_append: bad arguments");
    int rc = check_device(device, "qkv_rope_append");
    if (rc != NVK_OK)
        return rc;
    NVK_SCOPE(device);
    const float *const w[3]   = {wq_dev, wk_dev, wv_dev};
    float *const       out[3] = {q_dev, q_dev, q_dev}; // k and v go to the cache
    const int          n[3]   = {q_dim, t.row_len, t.row_len};
    return launch_linear(x_dev, rows, dim, w, out, n, 3, device, (hipStream_t)stream, "qkv_rope_append", false, &t, link);
}

// can rows x (q_dim + 2 kv_dim) go through qkv_rope_append?
bool nvk::qkv_rope_append_ok(int rows, int dim, int q_dim, int kv_dim, int device)
{
    if (NVK_GEMV_COLS != 4 || NVK_GEMV_MAX_ROWS < 2 || NVK_SKINNY16_MAX_ROWS != 16 || rows < 1 || rows > NVK_SKINNY_MAX_ROWS
        || dim % 4 != 0 || q_dim % 4 != 0 || kv_dim % 4 != 0)
        return false;
    // one or two rows: the GEMV (a pair inside a 4-column tile); more: the weight-streaming MFMA kernel wherever it gives
    // every row tile a workgroup of its own
    if (rows <= kFusedRowsMax)
        return true;
    const int n[3] = {q_dim, kv_dim, kv_dim};
    int       t = 32, tm = 1;
    skinny_choice(rows, n, 3, cu_count_of(device), &t, &tm, true);
    return tm == 1;
}

extern "C" {

int nvk_batch_rms_norm_device(float *out_dev, const float *in_dev, const float *weight_dev, int batch_size, int dim,
                              float eps, int device, void *stream)
{
    if (batch_size < 0 || dim <= 0 || !(eps >= 0.f) || (batch_size > 0 && (!out_dev || !in_dev || !weight_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_rms_norm_device: bad arguments");
    int rc = check_device(device, "nvk_batch_rms_norm_device");
    if (rc != NVK_OK || batch_size == 0)
        return rc;
    if (dim % 4 == 0 && (!aligned16(out_dev) || !aligned16(in_dev) || !aligned16(weight_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_rms_norm_device: pointers must be 16-byte aligned");
    NVK_SCOPE(device);
    hipLaunchKernelGGL(nvk::nvk_rms_norm_kernel<false>, dim3((unsigned)batch_size), dim3(256), 0, (hipStream_t)stream,
                       out_dev, const_cast<float *>(in_dev), (const float *)nullptr, weight_dev, dim, eps);
    HIP_TRY(hipGetLastError());
    return NVK_OK;
}

int nvk_batch_add_rms_norm_device(float *out_dev, float *x_dev, const float *addend_dev, const float *weight_dev,
                                  int batch_size, int dim, float eps, int device, void *stream)
{
    if (batch_size < 0 || dim <= 0 || !(eps >= 0.f)
        || (batch_size > 0 && (!out_dev || !x_dev || !addend_dev || !weight_dev)) || out_dev == x_dev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_add_rms_norm_device: bad arguments");
    int rc = check_device(device, "nvk_batch_add_rms_norm_device");
    if (rc != NVK_OK || batch_size == 0)
        return rc;
    if (dim % 4 == 0 && (!aligned16(out_dev) || !aligned16(x_dev) || !aligned16(addend_dev) || !aligned16(weight_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_add_rms_norm_device: pointers must be 16-byte aligned");
    NVK_SCOPE(device);
    hipLaunchKernelGGL(nvk::nvk_rms_norm_kernel<true>, dim3((unsigned)batch_size), dim3(256), 0, (hipStream_t)stream,
                       out_dev, x_dev, addend_dev, weight_dev, dim, eps);
    HIP_TRY(hipGetLastError());
    return NVK_OK;
}

int nvk_batch_matmul_device(float *out_dev, const float *in_dev, const float *weight_dev, int batch_size, int in_dim,
                            int out_dim, int device, void *stream)
{
    if (batch_size < 0 || in_dim <= 0 || out_dim <= 0 || (batch_size > 0 && (!out_dev || !in_dev || !weight_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_matmul_device: bad arguments");
    int rc = check_device(device, "nvk_batch_matmul_device");
    if (rc != NVK_OK || batch_size == 0)
        return rc;
    NVK_SCOPE(device);
    return launch_linear(in_dev, batch_size, in_dim, &weight_dev, &out_dev, &out_dim, 1, device, (hipStream_t)stream,
                         "nvk_batch_matmul_device");
}

int nvk_batch_matmul2_device(float *out_a_dev, float *out_b_dev, const float *in_dev, const float *weight_a_dev,
                             const float *weight_b_dev, int batch_size, int in_dim, int out_dim_a, int out_dim_b,
                             int device, void *stream)
{
    if (batch_size < 0 || in_dim <= 0 || out_dim_a <= 0 || out_dim_b <= 0
        || (batch_size > 0 && (!out_a_dev || !out_b_dev || !in_dev || !weight_a_dev || !weight_b_dev)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_matmul2_device: bad arguments");
    int rc = check_device(device, "nvk_batch_matmul2_device");
    if (rc != NVK_OK || batch_size == 0)
        return rc;
    NVK_SCOPE(device);
    const float *const w[2]   = {weight_a_dev, weight_b_dev};
    float *const       out[2] = {out_a_dev, out_b_dev};
    const int          n[2]   = {out_dim_a, out_dim_b};
    return launch_linear(in_dev, batch_size, in_dim, w, out, n, 2, device, (hipStream_t)stream, "nvk_batch_matmul2_device");
}

int nvk_batch_swiglu_ffn_device(float *hb_dev, float *hb2_dev, const float *in_dev, const float *w_gate_dev,
                                const float *w_up_dev, int batch_size, int in_dim, int hidden_dim, int device, void *stream)
{
    if (batch_size < 0 || in_dim <= 0 || hidden_dim <= 0
        || (batch_size > 0 && (!hb_dev || !hb2_dev || !in_dev || !w_gate_dev || !w_up_dev)) || hb_dev == hb2_dev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_batch_swiglu_ffn_device: bad arguments");
    int rc = check_device(device, "nvk_batch_swiglu_ffn_device");
    if (rc != NVK_OK || batch_size == 0)
        return rc;
    NVK_SCOPE(device);
    const float *const w[2]   = {w_gate_dev, w_up_dev};
    float *const       out[2] = {hb_dev, hb2_dev};
    const int          n[2]   = {hidden_dim, hidden_dim};
    // one or two rows on the GEMV: SwiGLU in its epilogue (hb2 is not touched); otherwise the two products, then SwiGLU
    const bool gemv = NVK_GEMV_COLS == 4 && batch_size <= NVK_GEMV_MAX_ROWS
                      && (batch_size < 2 || 2LL * hidden_dim <= NVK_GEMV2_MAX_COLS);
    rc = launch_linear(in_dev, batch_size, in_dim, w, out, n, 2, device, (hipStream_t)stream, "nvk_batch_swiglu_ffn_device",
                       gemv);
    if (rc != NVK_OK || gemv)
        return rc;
    const int64_t count = (int64_t)batch_size * hidden_dim;
    hipLaunchKernelGGL(nvk::nvk_swiglu_kernel, dim3((unsigned)std::min<int64_t>(1024, (count + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, hb_dev, hb2_dev, count);
    HIP_TRY(hipGetLastError());
    return NVK_OK;
}

int nvk_qkv_project_device(float *q_dev, float *k_dev, float *v_dev, const float *x_dev, float *xb_dev,
                           const float *rms_weight_dev, const float *wq_dev, const float *wk_dev, const float *wv_dev,
                           int batch_size, int dim, int q_dim, int kv_dim, float rms_eps, int device, void *stream)
{
    if (batch_size < 0 || dim <= 0 || q_dim <= 0 || kv_dim <= 0 || !(rms_eps >= 0.f)
        || (batch_size > 0 && (!q_dev || !k_dev || !v_dev || !x_dev || !wq_dev || !wk_dev || !wv_dev))
        || (batch_size > 0 && rms_weight_dev && !xb_dev))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_qkv_project_device: bad arguments");
    int rc = check_device(device, "nvk_qkv_project_device");
    if (rc != NVK_OK || batch_size == 0)
        return rc;
    const float *act = x_dev;
    if (rms_weight_dev) { // model.hpp:161: xb = rms_norm(x, rms_att_weight)
        rc = nvk_batch_rms_norm_device(xb_dev, x_dev, rms_weight_dev, batch_size, dim, rms_eps, device, stream);
        if (rc != NVK_OK)
            return rc;
        act = xb_dev;
    }
    NVK_SCOPE(device);
    const float *const w[3]   = {wq_dev, wk_dev, wv_dev}; // model.hpp:164-166
    float *const       out[3] = {q_dev, k_dev, v_dev};
    const int          n[3]   = {q_dim, kv_dim, kv_dim};
    return launch_linear(act, batch_size, dim, w, out, n, 3, device, (hipStream_t)stream, "nvk_qkv_project_device");
}

} // extern "C"
