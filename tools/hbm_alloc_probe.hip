// hbm_alloc_probe -- does the SIZE of a hipMalloc allocation change how fast a fixed 8 GiB of it streams?
// (5b reads the same 8.6 GB at 0.875 of peak from a one-layer pool and at 0.79 from a pool of 2+ layers.)
//   hipcc --offload-arch=gfx950 -O3 -o build/hbm_alloc_probe tools/hbm_alloc_probe.hip && build/hbm_alloc_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                                       \
    do {                                                                                                               \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess) {                                                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                                     \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// two streams read side by side (K and V): lane i of the grid reads a[i], b[i], a[i + stride], ...
__global__ void __launch_bounds__(512) read2_kernel(const v4f *__restrict__ a, const v4f *__restrict__ b, size_t n4, float *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t       i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4f          acc    = {0.f, 0.f, 0.f, 0.f};
    for (; i + 3 * stride < n4; i += 4 * stride) {
        v4f t[8];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            t[2 * u]     = __builtin_nontemporal_load(a + i + u * stride);
            t[2 * u + 1] = __builtin_nontemporal_load(b + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            acc += t[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f)
        out[0] = acc.x;
}

static int time_read(const v4f *a, const v4f *b, size_t n4, float *out, int cus, const char *what)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(read2_kernel, dim3(cus), dim3(512), 0, 0, a, b, n4, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best)
            best = ms;
    }
    printf("  %-58s %8.3f ms  %7.1f GB/s\n", what, best, 2.0 * n4 * 16 / (best * 1e-3) / 1e9);
    fflush(stdout);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    float *out;
    CHECK(hipMalloc((void **)&out, 4));
    const size_t gib = 1ull << 30, n4 = 4 * gib / 16; // 4 GiB from each of two buffers = 8 GiB per launch
    printf("%s: two streams of 4 GiB read side by side, 1 workgroup of 512 per CU\n", prop.name);
    for (size_t alloc : {(size_t)4, (size_t)8, (size_t)16, (size_t)32, (size_t)64}) {
        v4f *a, *b;
        CHECK(hipMalloc((void **)&a, alloc * gib));
        CHECK(hipMalloc((void **)&b, alloc * gib));
        CHECK(hipMemset(a, 0, alloc * gib));
        CHECK(hipMemset(b, 0, alloc * gib));
        char what[128];
        snprintf(what, sizeof what, "2 allocations of %zu GiB, first 4 GiB of each", alloc);
        if (time_read(a, b, n4, out, prop.multiProcessorCount, what))
            return 1;
        if (alloc > 4) {
            snprintf(what, sizeof what, "2 allocations of %zu GiB, last 4 GiB of each", alloc);
            if (time_read(a + (alloc - 4) * gib / 16, b + (alloc - 4) * gib / 16, n4, out, prop.multiProcessorCount, what))
                return 1;
        }
        CHECK(hipFree(a));
        CHECK(hipFree(b));
    }
    // one allocation holding both streams at different distances
    v4f *c;
    CHECK(hipMalloc((void **)&c, 64 * gib));
    CHECK(hipMemset(c, 0, 64 * gib));
    for (size_t dist : {(size_t)4, (size_t)8, (size_t)16, (size_t)32}) {
        char what[128];
        snprintf(what, sizeof what, "1 allocation of 64 GiB, streams %zu GiB apart", dist);
        if (time_read(c, c + dist * gib / 16, n4, out, prop.multiProcessorCount, what))
            return 1;
    }
    CHECK(hipFree(c));
    return 0;
}
