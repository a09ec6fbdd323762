// hbm_read_peak -- what a plain streaming read achieves on this GPU: the ceiling the decode kernel's
// "achieved GB/s" should be read against (the 8 TB/s of MI355X_MICROARCH.md is the pin rate, not reachable
// by any kernel).  Each lane sums float4 non-temporal loads, 8 independent loads in flight, grid-stride.
//   hipcc --offload-arch=gfx950 -O3 -o build/hbm_read_peak tools/hbm_read_peak.hip && build/hbm_read_peak [GiB]
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

template <int UNROLL> __global__ void __launch_bounds__(512) read_kernel(const v4f *__restrict__ src, size_t n4, float *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t       i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4f          acc    = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        v4f t[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            t[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            acc += t[u];
    }
    for (; i < n4; i += stride)
        acc += __builtin_nontemporal_load(src + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) // never true: keeps the loads alive
        out[0] = acc.x;
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t n4  = (size_t)(gib * (1ull << 30)) / 16;
    v4f         *buf;
    float       *out;
    CHECK(hipMalloc((void **)&buf, n4 * 16));
    CHECK(hipMalloc((void **)&out, 4));
    CHECK(hipMemset(buf, 0, n4 * 16));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    printf("%s, %d CUs, %.1f GiB buffer (beyond the 256 MB Infinity Cache), float4 non-temporal loads\n", prop.name,
           prop.multiProcessorCount, gib);
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        const int grid = prop.multiProcessorCount * wgs_per_cu;
        float     best = 1e30f;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(512), 0, 0, buf, n4, out);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best)
                best = ms;
        }
        printf("  %d workgroups of 512 per CU: %8.3f ms  %7.1f GB/s\n", wgs_per_cu, best, n4 * 16 / (best * 1e-3) / 1e9);
    }
    // a launch the size of config 2 (1.08 GB): the ramp-up and the tail are part of a 160 us kernel
    const size_t small4 = (size_t)1075871744 / 16;
    float        best   = 1e30f;
    for (int rep = 0; rep < 10; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(read_kernel<8>, dim3(prop.multiProcessorCount * 2), dim3(512), 0, 0, buf + (rep % 3) * small4,
                           small4, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best)
            best = ms;
    }
    printf("  1.076 GB (config 2's bytes), rotating over 3 regions: %8.1f us  %7.1f GB/s\n", best * 1e3,
           small4 * 16 / (best * 1e-3) / 1e9);
    return 0;
}
