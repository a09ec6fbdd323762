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
// Two kernels behind one dispatcher:
//   * nvk_gemm_nt_kernel     -- more than 32 rows: a 128x128 tile per workgroup on
//                               v_mfma_f32_32x32x2_f32 (exact fp32 products and sums, an fmaf chain
//                               per output), bound by the fp32 MFMA rate;
//   * nvk_gemm_skinny_kernel -- at most 32 rows (decode batches of a few requests): the weights are
//                               streamed once, 32 output columns per workgroup, the contraction
//                               split over the workgroup's 8 waves; bound by HBM (the weight bytes).
// Both take up to three (weight, out) segments in one launch, so wq, wk and wv share the launch and
// the normalised activations.  gfx950 only; no CPU path.

#include <hip/hip_runtime.h>

#include <algorithm>

#include "nvk_internal.h"

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return nvk::fail(e__ == hipErrorOutOfMemory ? NVK_ERR_OOM : NVK_ERR_HIP, "%s failed: %s (%s:%d)", #expr,  \
                             hipGetErrorString(e__), __FILE__, __LINE__);                                              \
    } while (0)

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
__global__ void __launch_bounds__(256) nvk_rms_norm_kernel(float *__restrict__ out, const float *__restrict__ in,
                                                           const float *__restrict__ weight, int dim, float eps)
{
    __shared__ float s_part[4];
    const float     *x = in + (int64_t)blockIdx.x * dim;
    float           *y = out + (int64_t)blockIdx.x * dim;
    float            ss = 0.f;
    const int        vec = (dim % 4 == 0) ? dim / 4 : 0;
    for (int i = threadIdx.x; i < vec; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + 4 * i);
        ss = fmaf(v.x, v.x, ss), ss = fmaf(v.y, v.y, ss), ss = fmaf(v.z, v.z, ss), ss = fmaf(v.w, v.w, ss);
    }
    for (int i = 4 * vec + threadIdx.x; i < dim; i += 256)
        ss = fmaf(x[i], x[i], ss);
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

// ---- C = A . W^T, more than 32 rows: 128 x 128 tiles on fp32 MFMA ---------------------------------
// 4 waves as 2 x 2, each a 64 x 64 sub-tile = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs).
// A and W tiles of 32 contraction steps go global -> registers -> LDS (row stride 36 floats: the
// lanes' ds_read_b128 fall in distinct banks), double-buffered, one barrier per step.  The
// instruction's two k lanes are re-associated over four MFMAs so a lane's operands are one float4:
// lane (j, half) of MFMA e supplies element k = 8c + 4*half + e of its row.
// Tile ids are dealt so that the row tiles of one column tile run on one XCD at about the same time
// (ids go round-robin over the 8 XCDs): the weight tile is then read from HBM once and met in that L2.
constexpr int kGemmBK  = 32;
constexpr int kGemmLds = kGemmBK + 4;

__global__ void __launch_bounds__(256, 2) nvk_gemm_nt_kernel(const LinearParams p)
{
    constexpr int BM = 128, BN = 128;
    extern __shared__ __attribute__((aligned(16))) float s_mem[]; // [2][BM + BN][kGemmLds]

    const int span = 8 * p.m_tiles;
    const int rem  = blockIdx.x % span;
    const int nt   = blockIdx.x / span * 8 + rem % 8;
    const int mt   = rem / 8;
    if (nt >= p.n_tiles)
        return;
    const LinearSeg sg = segment_of(p, nt);
    const int       n0 = (nt - sg.tile0) * BN, m0 = mt * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int lr = tid >> 3, lc = (tid & 7) * 4; // this thread's (row, first k) inside a 32-row pass of a tile

    const float *arow[BM / 32], *wrow[BN / 32];
#pragma unroll
    for (int i = 0; i < BM / 32; i++)
        arow[i] = p.a + (int64_t)min(m0 + lr + 32 * i, p.m - 1) * p.lda + lc; // rows past the edge: computed, not stored
#pragma unroll
    for (int i = 0; i < BN / 32; i++)
        wrow[i] = sg.w + (int64_t)min(n0 + lr + 32 * i, sg.n - 1) * p.k + lc;

    float4 ra[BM / 32], rb[BN / 32];
    auto   fetch = [&](int k0) {
        const bool inside = k0 + lc < p.k; // k % 4 == 0: a float4 is wholly inside or outside
#pragma unroll
        for (int i = 0; i < BM / 32; i++)
            ra[i] = inside ? *reinterpret_cast<const float4 *>(arow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < BN / 32; i++)
            rb[i] = inside ? *reinterpret_cast<const float4 *>(wrow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int buf) {
        float *sa = s_mem + buf * (BM + BN) * kGemmLds, *sb = sa + BM * kGemmLds;
#pragma unroll
        for (int i = 0; i < BM / 32; i++)
            *reinterpret_cast<float4 *>(sa + (lr + 32 * i) * kGemmLds + lc) = ra[i];
#pragma unroll
        for (int i = 0; i < BN / 32; i++)
            *reinterpret_cast<float4 *>(sb + (lr + 32 * i) * kGemmLds + lc) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                acc[a][b][r] = 0.f;

    const int nk = (p.k + kGemmBK - 1) / kGemmBK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk)
            fetch((kt + 1) * kGemmBK);
        const float *sa = s_mem + (kt & 1) * (BM + BN) * kGemmLds + (wm * 64 + j) * kGemmLds + 4 * half;
        const float *sb = s_mem + (kt & 1) * (BM + BN) * kGemmLds + (BM + wn * 64 + j) * kGemmLds + 4 * half;
#pragma unroll
        for (int c = 0; c < kGemmBK / 8; c++) {
            const float4 a0 = *reinterpret_cast<const float4 *>(sa + 8 * c);
            const float4 a1 = *reinterpret_cast<const float4 *>(sa + 32 * kGemmLds + 8 * c);
            const float4 b0 = *reinterpret_cast<const float4 *>(sb + 8 * c);
            const float4 b1 = *reinterpret_cast<const float4 *>(sb + 32 * kGemmLds + 8 * c);
#define NVK_MFMA4(A, B, ACC)                                                                                           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, ACC, 0, 0, 0);                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, ACC, 0, 0, 0);                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, ACC, 0, 0, 0);                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, ACC, 0, 0, 0);
            NVK_MFMA4(a0, b0, acc[0][0])
            NVK_MFMA4(a0, b1, acc[0][1])
            NVK_MFMA4(a1, b0, acc[1][0])
            NVK_MFMA4(a1, b1, acc[1][1])
        }
        if (kt + 1 < nk)
            stash((kt + 1) & 1); // the other buffer: every wave left it before the previous barrier
        __syncthreads();
    }

    // D[row][col]: col = lane & 31, register r holds row (r & 3) + 8 * (r >> 2) + 4 * half
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int col = n0 + wn * 64 + 32 * b + j;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < p.m && col < sg.n)
                    sg.out[(int64_t)row * sg.ldo + col] = acc[a][b][r];
            }
        }
}

// ---- C = A . W^T, at most 32 rows: stream the weights once ----------------------------------------
// One workgroup = 32 output columns (32 weight rows), 8 waves, the contraction dealt round-robin over
// the waves in steps of 64.  Every wave is on its own: it brings its 32 x 64 weight tile and the
// matching activation tile global -> registers (the whole next step in flight while the current one
// is multiplied) -> its private LDS region -> MFMA operands; no barrier until the waves' partial
// 32 x 32 tiles meet in LDS at the end.  Rows past m are computed on clamped addresses and dropped.
constexpr int kSkinnyBK  = 64;
constexpr int kSkinnyLds = kSkinnyBK + 4;
constexpr int kSkinnyWaves = 8;

__global__ void __launch_bounds__(64 * kSkinnyWaves) nvk_gemm_skinny_kernel(const LinearParams p)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[]; // [waves][2][32][kSkinnyLds]
    const int nt = blockIdx.x;
    const LinearSeg sg = segment_of(p, nt);
    const int       n0 = (nt - sg.tile0) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int lr = lane >> 4, lc = (lane & 15) * 4; // 4 rows x 64 k per load instruction

    float *sa = s_mem + wave * 2 * 32 * kSkinnyLds, *sb = sa + 32 * kSkinnyLds;
    const float *arow[8], *wrow[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        arow[i] = p.a + (int64_t)min(lr + 4 * i, p.m - 1) * p.lda + lc;
        wrow[i] = sg.w + (int64_t)min(n0 + lr + 4 * i, sg.n - 1) * p.k + lc;
    }
    float4 ra[8], rb[8];
    auto   fetch = [&](int k0) {
        const bool inside = k0 + lc < p.k;
#pragma unroll
        for (int i = 0; i < 8; i++)
            rb[i] = inside ? load_stream(wrow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; i++)
            ra[i] = inside ? *reinterpret_cast<const float4 *>(arow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = 0.f;

    const int nk = (p.k + kSkinnyBK - 1) / kSkinnyBK;
    if (wave < nk)
        fetch(wave * kSkinnyBK);
    for (int kt = wave; kt < nk; kt += kSkinnyWaves) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            *reinterpret_cast<float4 *>(sa + (lr + 4 * i) * kSkinnyLds + lc) = ra[i];
            *reinterpret_cast<float4 *>(sb + (lr + 4 * i) * kSkinnyLds + lc) = rb[i];
        }
        if (kt + kSkinnyWaves < nk)
            fetch((kt + kSkinnyWaves) * kSkinnyBK);
        const float *pa = sa + j * kSkinnyLds + 4 * half, *pb = sb + j * kSkinnyLds + 4 * half;
#pragma unroll
        for (int c = 0; c < kSkinnyBK / 8; c++) {
            const float4 a0 = *reinterpret_cast<const float4 *>(pa + 8 * c);
            const float4 b0 = *reinterpret_cast<const float4 *>(pb + 8 * c);
            NVK_MFMA4(a0, b0, acc)
        }
    }
#undef NVK_MFMA4

    // the waves' partial tiles meet in LDS (each wave's own region, 16 x 64 floats); then 256 of the
    // 512 threads add the eight partials of four outputs each, in wave order
#pragma unroll
    for (int r = 0; r < 16; r++)
        sa[r * 64 + lane] = acc[r]; // the wave's own region: its operand reads are behind it
    __syncthreads();
    if (tid < 256) {
        const int r = tid >> 4, l4 = (tid & 15) * 4; // register r, lanes l4 .. l4 + 3
        float4    sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < kSkinnyWaves; w++) {
            const float4 v = *reinterpret_cast<const float4 *>(s_mem + w * 2 * 32 * kSkinnyLds + r * 64 + l4);
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        const int h   = l4 >> 5;                               // the four lanes share a half
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = n0 + (l4 & 31);
        if (row < p.m) {
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

} // namespace nvk

namespace {

struct DeviceScope
{
    int  previous = -1;
    bool ok       = false;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&previous) != hipSuccess)
            previous = -1;
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (previous >= 0)
            (void)hipSetDevice(previous);
    }
};

int check_device(int device, const char *who)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nvk::fail(NVK_ERR_NO_DEVICE, "%s: no HIP device visible (this library has no CPU path)", who);
    if (device < 0 || device >= ndev)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: device %d outside [0,%d)", who, device, ndev);
    return NVK_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// One launch for up to three (weight, out, n) segments sharing the activations `in` [batch, in_dim].
int launch_linear(const float *in, int batch, int in_dim, const float *const *w, float *const *out, const int *n,
                  int nseg, int device, hipStream_t st, const char *who)
{
    if (in_dim % 4 != 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: in_dim=%d must be a multiple of 4", who, in_dim);
    if (!aligned16(in))
        return nvk::fail(NVK_ERR_INVALID_ARG, "%s: the activations must be 16-byte aligned", who);
    nvk::LinearParams p;
    p.a = in, p.m = batch, p.k = in_dim, p.lda = in_dim;
    const bool skinny = batch <= 32;
    const int  bn     = skinny ? 32 : 128;
    long long  tiles  = 0;
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
    if (skinny) {
        p.m_tiles = 1;
        const size_t lds = (size_t)nvk::kSkinnyWaves * 2 * 32 * nvk::kSkinnyLds * sizeof(float);
        static bool  opted[64]; // > 64 KiB of dynamic LDS needs the attribute, once per device
        if (!opted[device & 63]) {
            HIP_TRY(hipFuncSetAttribute((const void *)nvk::nvk_gemm_skinny_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            opted[device & 63] = true;
        }
        hipLaunchKernelGGL(nvk::nvk_gemm_skinny_kernel, dim3((unsigned)tiles), dim3(64 * nvk::kSkinnyWaves), lds, st, p);
    }
    else {
        p.m_tiles = (batch + 127) / 128;
        const long long wgs = (tiles + 7) / 8 * 8 * p.m_tiles;
        if (wgs > 0x7fffffffLL)
            return nvk::fail(NVK_ERR_INVALID_ARG, "%s: %lld workgroups exceed the grid", who, wgs);
        const size_t lds = (size_t)2 * 256 * nvk::kGemmLds * sizeof(float);
        static bool  opted[64];
        if (!opted[device & 63]) {
            HIP_TRY(hipFuncSetAttribute((const void *)nvk::nvk_gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
            opted[device & 63] = true;
        }
        hipLaunchKernelGGL(nvk::nvk_gemm_nt_kernel, dim3((unsigned)wgs), dim3(256), lds, st, p);
    }
    HIP_TRY(hipGetLastError());
    return NVK_OK;
}

} // namespace

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
    DeviceScope scope(device);
    hipLaunchKernelGGL(nvk::nvk_rms_norm_kernel, dim3((unsigned)batch_size), dim3(256), 0, (hipStream_t)stream, out_dev,
                       in_dev, weight_dev, dim, eps);
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
    DeviceScope scope(device);
    return launch_linear(in_dev, batch_size, in_dim, &weight_dev, &out_dev, &out_dim, 1, device, (hipStream_t)stream,
                         "nvk_batch_matmul_device");
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
    DeviceScope        scope(device);
    const float *const w[3]   = {wq_dev, wk_dev, wv_dev}; // model.hpp:164-166
    float *const       out[3] = {q_dev, k_dev, v_dev};
    const int          n[3]   = {q_dim, kv_dim, kv_dim};
    return launch_linear(act, batch_size, dim, w, out, n, 3, device, (hipStream_t)stream, "nvk_qkv_project_device");
}

} // extern "C"
