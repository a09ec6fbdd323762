// Litmus for a one-request model step as ONE resident kernel: a chain of GEMVs y_s = W_s . y_{s-1}
// (TinyLlama's four projections per layer, 22 layers, every weight its own memory) run
//   (a) as one launch per product (the shipped GEMV's shape: 4 weight rows per workgroup, 8 waves dealing k), back to back;
//   (b) as ONE launch: two resident workgroups per CU, stages separated by a grid-wide barrier (data stored through to memory
//       and loaded from there, no fences: profiles/r03_grid_barrier.log), the first weight loads of stage s + 1 issued BEFORE the
//       barrier, so that the weight stream does not drain around it.
// Results of (a) and (b) are compared word for word (same summation order), (a) against a double-precision host product.
//   hipcc --offload-arch=gfx950 -O3 -o build/probe/resident_chain_probe tools/resident_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kWaves = 8, kDepth = 4, kMaxK = 8192, kMaxStages = 96;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Stage
{
    const float *w; // [n][k]
    const float *x; // [k]
    float       *y; // [n]
    int          k, n;
};
struct Chain
{
    Stage     st[kMaxStages];
    int       n_stages;
    unsigned *count, *abort_flag; // the barrier's counter (zero at launch), the flag a bounded spin raises
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_w(const float *p)
{
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// one 4-row tile, the shipped GEMV's order: wave w takes k-steps w, w + 8, ... (rotated by the tile), 16 lanes to a row
template <bool RESIDENT>
__device__ __forceinline__ void reduce_store(float acc, float (*s_red)[4], const Stage &s, int n0, int tid, int lane, int wave)
{
    float v = acc;
    v += __shfl_xor(v, 1), v += __shfl_xor(v, 2), v += __shfl_xor(v, 4), v += __shfl_xor(v, 8);
    if ((lane & 15) == 0)
        s_red[wave][lane >> 4] = v;
    __syncthreads();
    if (tid < 4 && n0 + tid < s.n) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; w++)
            sum += s_red[w][tid];
        if (RESIDENT)
            __hip_atomic_store(s.y + n0 + tid, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // through to memory (sc1)
        else
            s.y[n0 + tid] = sum;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64 * kWaves) gemv_launch(const Stage s)
{
    __shared__ float s_red[kWaves][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane >> 4, lc = (lane & 15) * 4;
    const int n0 = blockIdx.x * 4, nk = s.k / 64, rot = (blockIdx.x * 37u) % nk;
    const float *wrow = s.w + (size_t)min(n0 + lr, s.n - 1) * s.k + lc;
    const int steps = nk > wave ? (nk - wave + kWaves - 1) / kWaves : 0;
    auto k_of = [&](int i) { return ((wave + i * kWaves + rot) % nk) * 64; };
    float4 rw[kDepth], rx[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth; d++)
        if (d < steps)
            rw[d] = load_w(wrow + k_of(d)), rx[d] = *reinterpret_cast<const float4 *>(s.x + k_of(d) + lc);
    float acc = 0.f;
    for (int i = 0; i < steps; i += kDepth)
#pragma unroll
        for (int d = 0; d < kDepth; d++)
            if (i + d < steps) {
                acc = fmaf(rw[d].x, rx[d].x, acc), acc = fmaf(rw[d].y, rx[d].y, acc);
                acc = fmaf(rw[d].z, rx[d].z, acc), acc = fmaf(rw[d].w, rx[d].w, acc);
                if (i + d + kDepth < steps)
                    rw[d] = load_w(wrow + k_of(i + d + kDepth)), rx[d] = *reinterpret_cast<const float4 *>(s.x + k_of(i + d + kDepth) + lc);
            }
    reduce_store<false>(acc, s_red, s, n0, tid, lane, wave);
}

// every workgroup has stored (sc1) and waited; meet the others.  One counter per XCD, the XCD's last arrival adds to the
// grid's, the grid's last arrival publishes the round.  false: the spin ran out (the flag is raised, everyone leaves)
__device__ __forceinline__ bool grid_meet(unsigned *count, unsigned *abort_flag, unsigned round)
{
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        bool      ok = true, wait = true;
        unsigned *flag = count + 32, *mine = count + 256 + 32 * (blockIdx.x & 7);
        if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == round * (gridDim.x / 8) - 1)
            if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 8 * round - 1) {
                __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wait = false;
            }
        int spins = 0;
        while (wait && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
            if (++spins > (1 << 20) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <bool PREFETCH>
__global__ void __launch_bounds__(64 * kWaves, 2) chain_resident(const Chain c)
{
    __shared__ float s_red[kWaves][4];
    __shared__ __attribute__((aligned(16))) float s_x[kMaxK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane >> 4, lc = (lane & 15) * 4;
    const int G = gridDim.x;

    // this wave's walk over (tile, k-step) pairs of a stage, flattened: item i = (tile wg + (i / steps) G, step i % steps)
    struct Walk
    {
        const float *w;
        int          k, n, nk, steps, tiles, items;
    };
    auto walk_of = [&](const Stage &s) {
        Walk q;
        q.w = s.w, q.k = s.k, q.n = s.n, q.nk = s.k / 64;
        q.steps = q.nk > wave ? (q.nk - wave + kWaves - 1) / kWaves : 0;
        const int all = (s.n + 3) / 4;
        q.tiles = (int)blockIdx.x < all ? (all - (int)blockIdx.x + G - 1) / G : 0;
        q.items = q.tiles * q.steps;
        return q;
    };
    auto addr = [&](const Walk &q, int i, int *k0) {
        const int t = i / q.steps, st = i - t * q.steps, tile = blockIdx.x + t * G;
        const int rot = (tile * 37u) % q.nk;
        *k0 = ((wave + st * kWaves + rot) % q.nk) * 64;
        return q.w + (size_t)min(tile * 4 + lr, q.n - 1) * q.k + lc + *k0;
    };
    float4 rw[kDepth];
    int    rk[kDepth];
    Walk   q = walk_of(c.st[0]);
#pragma unroll
    for (int d = 0; d < kDepth; d++)
        if (d < q.items)
            rw[d] = load_w(addr(q, d, &rk[d]));
    for (int si = 0; si < c.n_stages; si++) {
        const Stage &s = c.st[si];
        if (PREFETCH && si > 0 && !grid_meet(c.count, c.abort_flag, (unsigned)si))
            return;
        // the stage's input: from memory (another workgroup of this launch wrote it, stage 0 apart), into LDS
        for (int i = tid * 4; i < s.k; i += 64 * kWaves * 4) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.x), 0, s.k * 4, 0x00020000);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 4, 0, 16 /* sc1 */);
            *reinterpret_cast<u32x4 *>(s_x + i) = v;
        }
        __syncthreads();
        float acc = 0.f;
        int   done = 0; // steps of the current tile consumed
        int   tile = 0;
        for (int i = 0; i < q.items; i += kDepth)
#pragma unroll
            for (int d = 0; d < kDepth; d++)
                if (i + d < q.items) {
                    const float4 xv = *reinterpret_cast<const float4 *>(s_x + rk[d] + lc);
                    acc = fmaf(rw[d].x, xv.x, acc), acc = fmaf(rw[d].y, xv.y, acc);
                    acc = fmaf(rw[d].z, xv.z, acc), acc = fmaf(rw[d].w, xv.w, acc);
                    if (i + d + kDepth < q.items)
                        rw[d] = load_w(addr(q, i + d + kDepth, &rk[d]));
                    if (++done == q.steps) { // (steps is the same for every tile of the wave; all waves reach the barriers inside)
                        reduce_store<true>(acc, s_red, s, (blockIdx.x + tile * G) * 4, tid, lane, wave);
                        acc = 0.f, done = 0, tile++;
                    }
                }
        // waves with no step of their own (k / 64 < 8) still have to meet the others at each tile's barriers
        if (q.steps == 0)
            for (int t = 0; t < q.tiles; t++)
                reduce_store<true>(0.f, s_red, s, (blockIdx.x + t * G) * 4, tid, lane, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's stores are in memory
        if (si + 1 < c.n_stages) {
            q = walk_of(c.st[si + 1]);
            if (PREFETCH) {
                __syncthreads(); // (every wave's stores: before anyone's arrival, see grid_meet's thread 0)
#pragma unroll
                for (int d = 0; d < kDepth; d++)
                    if (d < q.items)
                        rw[d] = load_w(addr(q, d, &rk[d]));
            }
        }
        __syncthreads();
        if (!PREFETCH && si + 1 < c.n_stages) { // (the control: the next stage's first loads behind the barrier)
            if (!grid_meet(c.count, c.abort_flag, (unsigned)si + 1))
                return;
#pragma unroll
            for (int d = 0; d < kDepth; d++)
                if (d < q.items)
                    rw[d] = load_w(addr(q, d, &rk[d]));
        }
    }
}

int main(int argc, char **argv)
{
    const int layers = argc > 1 ? atoi(argv[1]) : 22, reps = argc > 2 ? atoi(argv[2]) : 5;
    const int dims[4][2] = {{2048, 2048}, {2048, 11264}, {5632, 2048}, {2048, 2560}}; // wo, gate|up, down, q|k|v: k, n
    const int n_stages = 4 * layers;
    if (n_stages > kMaxStages) { printf("at most %d layers\n", kMaxStages / 4); return 1; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    size_t wfloats = 0;
    for (int s = 0; s < n_stages; s++) wfloats += (size_t)dims[s % 4][0] * dims[s % 4][1];
    printf("%d stages, %.2f GB of weights, %d CUs\n", n_stages, wfloats * 4 / 1e9, prop.multiProcessorCount);
    std::vector<float> hw(wfloats);
    unsigned           seed = 12345;
    for (size_t i = 0; i < wfloats; i++) {
        seed     = seed * 1664525u + 1013904223u;
        hw[i]    = ((seed >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.06f;
    }
    float *dw, *dact[2];
    CHECK(hipMalloc(&dw, wfloats * 4));
    CHECK(hipMemcpy(dw, hw.data(), wfloats * 4, hipMemcpyHostToDevice));
    // activations: every stage writes a buffer of its own (two sets: launches / resident)
    const size_t act_stride = 12288;
    for (int v = 0; v < 2; v++) {
        CHECK(hipMalloc(&dact[v], (n_stages + 1) * act_stride * 4));
        CHECK(hipMemset(dact[v], 0, (n_stages + 1) * act_stride * 4));
    }
    std::vector<float> hx(2048);
    for (int i = 0; i < 2048; i++) hx[i] = sinf(0.37f * i);
    for (int v = 0; v < 2; v++) CHECK(hipMemcpy(dact[v], hx.data(), 2048 * 4, hipMemcpyHostToDevice));
    Chain  ch[2];
    size_t off = 0;
    for (int v = 0; v < 2; v++) {
        off = 0;
        for (int s = 0; s < n_stages; s++) {
            Stage &st = ch[v].st[s];
            st.k = dims[s % 4][0], st.n = dims[s % 4][1];
            st.w = dw + off, off += (size_t)st.k * st.n;
            st.x = dact[v] + s * act_stride, st.y = dact[v] + (s + 1) * act_stride;
        }
        ch[v].n_stages = n_stages;
    }
    unsigned *ctl;
    CHECK(hipMalloc(&ctl, 8192));
    ch[1].count = ctl, ch[1].abort_flag = ctl + 1024;
    int fit = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, chain_resident<true>, 64 * kWaves, 0));
    printf("resident kernel: %d workgroups fit a CU\n", fit);
    if (fit < 2) return 1;
    const int grid = 2 * prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    // (a) as a graph of launches
    hipGraph_t     graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < n_stages; s++)
        hipLaunchKernelGGL(gemv_launch, dim3((ch[0].st[s].n + 3) / 4), dim3(64 * kWaves), 0, st, ch[0].st[s]);
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0, st));
        CHECK(hipGraphLaunch(exec, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) %d launches replayed as a graph: %.3f ms = %.1f us a layer, %.2f TB/s\n", n_stages, ms, ms * 1e3 / layers, wfloats * 4 / ms / 1e9);
    }
    for (int variant = 0; variant < 2; variant++)
        for (int r = 0; r < reps; r++) {
            CHECK(hipMemsetAsync(ctl, 0, 8192, st));
            CHECK(hipEventRecord(e0, st));
            if (variant == 0)
                hipLaunchKernelGGL(chain_resident<true>, dim3(grid), dim3(64 * kWaves), 0, st, ch[1]);
            else
                hipLaunchKernelGGL(chain_resident<false>, dim3(grid), dim3(64 * kWaves), 0, st, ch[1]);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned ab = 0;
            CHECK(hipMemcpy(&ab, ctl + 1024, 4, hipMemcpyDeviceToHost));
            printf("(b) one resident launch, %s: %.3f ms = %.1f us a layer, %.2f TB/s%s\n", variant == 0 ? "next stage's first loads BEFORE the barrier" : "first loads behind the barrier",
                   ms, ms * 1e3 / layers, wfloats * 4 / ms / 1e9, ab ? "  ABORTED (a spin ran out)" : "");
            if (ab) return 1;
        }
    // the two ways agree word for word; the first stage against the host
    std::vector<float> ya((n_stages + 1) * act_stride), yb((n_stages + 1) * act_stride);
    CHECK(hipMemcpy(ya.data(), dact[0], ya.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(yb.data(), dact[1], yb.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nonzero = 0;
    for (size_t i = 0; i < ya.size(); i++) diff += ya[i] != yb[i], nonzero += ya[i] != 0.f;
    double worst = 0;
    for (int n = 0; n < 2048; n++) {
        double sum = 0;
        for (int k = 0; k < 2048; k++) sum += (double)hw[(size_t)n * 2048 + k] * hx[k];
        worst = fmax(worst, fabs(sum - ya[act_stride + n]));
    }
    printf("words that differ between (a) and (b): %zu of %zu (%zu not zero); stage 0 against the host: max |err| %.2e\n", diff, ya.size(), nonzero, worst);
    return diff != 0 || worst > 1e-4;
}
