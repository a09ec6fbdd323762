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

// The decode kernel's shape of access instead of a grid-stride walk: workgroup w streams ONE contiguous region, 4-KiB rows,
// wave v the v-th 512-B piece of 8 consecutive rows per step (16 lanes x 16 B x 2 per row piece, 4 rows per instruction
// pair), two steps in flight.  LOCK: a bare s_barrier per step keeps the workgroup's 8 waves on the same rows.
template <bool LOCK> __global__ void __launch_bounds__(512) rows_kernel(const v4f *__restrict__ src, size_t rows_per_wg, float *out)
{
    const int    lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row0 = (size_t)blockIdx.x * rows_per_wg;
    // float4 index of (row, wave piece, lane): a row is 256 float4, a piece 32 float4; lane -> (row of 4, 16 lanes)
    const v4f *base = src + row0 * 256 + wave * 32 + (lane >> 4) * 256 + (lane & 15);
    v4f        acc  = {0.f, 0.f, 0.f, 0.f};
    for (size_t r = 0; r + 8 <= rows_per_wg; r += 8) {
        if (LOCK)
            __builtin_amdgcn_s_barrier();
        const v4f *p = base + r * 256;
        v4f        t[4];
        t[0] = __builtin_nontemporal_load(p);
        t[1] = __builtin_nontemporal_load(p + 16);
        t[2] = __builtin_nontemporal_load(p + 4 * 256);
        t[3] = __builtin_nontemporal_load(p + 4 * 256 + 16);
        acc += t[0] + t[1] + t[2] + t[3];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f)
        out[0] = acc.x;
}

// MHA's shape: a token row is PIECES x 4 KiB and a workgroup (WAVES waves, 512 B each) reads one piece of WAVES / 8 x 4 KiB of
// every row of its region; the row's other pieces belong to workgroups that run at another time (kv-head group outermost).
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) strided_kernel(const v4f *__restrict__ src, size_t rows_per_wg, int row_f4, int wgs_per_region,
                                                             float *out)
{
    const int    lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t region = blockIdx.x % wgs_per_region, piece = blockIdx.x / wgs_per_region; // piece outermost, as kv-head groups are
    const v4f   *base = src + region * rows_per_wg * row_f4 + piece * (WAVES * 32) + wave * 32 + (size_t)(lane >> 4) * row_f4 + (lane & 15);
    v4f          acc  = {0.f, 0.f, 0.f, 0.f};
    for (size_t r = 0; r + 8 <= rows_per_wg; r += 8) {
        const v4f *p = base + r * row_f4;
        v4f        t[4];
        t[0] = __builtin_nontemporal_load(p);
        t[1] = __builtin_nontemporal_load(p + 16);
        t[2] = __builtin_nontemporal_load(p + 4 * (size_t)row_f4);
        t[3] = __builtin_nontemporal_load(p + 4 * (size_t)row_f4 + 16);
        acc += t[0] + t[1] + t[2] + t[3];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f)
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
    // one contiguous region per workgroup, whole 4-KiB rows by its 8 waves (the decode kernel's pattern), free-running and
    // in lockstep; 2 workgroups per CU's worth of regions, one workgroup resident per CU at a time is not enforced here
    for (int wgs_per_cu : {1, 2}) {
        const int    grid = prop.multiProcessorCount * wgs_per_cu;
        const size_t rows = (n4 / 256 / grid) / 8 * 8;
        for (int lock = 0; lock < 2; lock++) {
            float best = 1e30f;
            for (int rep = 0; rep < 6; rep++) {
                CHECK(hipEventRecord(a));
                if (lock)
                    hipLaunchKernelGGL(rows_kernel<true>, dim3(grid), dim3(512), 0, 0, buf, rows, out);
                else
                    hipLaunchKernelGGL(rows_kernel<false>, dim3(grid), dim3(512), 0, 0, buf, rows, out);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep > 0 && ms < best)
                    best = ms;
            }
            printf("  rows: %d workgroups per CU, one contiguous region each, %s: %8.3f ms  %7.1f GB/s\n", wgs_per_cu,
                   lock ? "waves in lockstep" : "waves free-running", best, (double)rows * grid * 4096 / (best * 1e-3) / 1e9);
        }
    }
    // MHA 32/32/128: 16-KiB token rows; 8-wave workgroups read 4-KiB pieces (4 per row), 16-wave workgroups 8-KiB pieces (2 per row)
    {
        const int    row_f4 = 1024;                       // 16 KiB
        const size_t rows_all = n4 / row_f4;
        const int    regions = 512;                       // "items": contiguous runs of rows
        const size_t rows = rows_all / regions / 8 * 8;
        for (int waves : {8, 16}) {
            const int pieces = 32 / waves;
            float     best = 1e30f;
            for (int rep = 0; rep < 6; rep++) {
                CHECK(hipEventRecord(a));
                if (waves == 8)
                    hipLaunchKernelGGL(strided_kernel<8>, dim3(regions * pieces), dim3(512), 0, 0, buf, rows, row_f4, regions, out);
                else
                    hipLaunchKernelGGL(strided_kernel<16>, dim3(regions * pieces), dim3(1024), 0, 0, buf, rows, row_f4, regions, out);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep > 0 && ms < best)
                    best = ms;
            }
            printf("  16-KiB rows in %d pieces by %d-wave workgroups (piece outermost): %8.3f ms  %7.1f GB/s\n", pieces, waves, best,
                   (double)rows * regions * row_f4 * 16 / (best * 1e-3) / 1e9);
        }
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
    // the same bytes in the decode kernel's shape: 256 workgroups, one contiguous 4.2-MB region each
    {
        const int    grid = prop.multiProcessorCount;
        const size_t rows = (small4 / 256 / grid) / 8 * 8;
        best              = 1e30f;
        for (int rep = 0; rep < 10; rep++) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(rows_kernel<false>, dim3(grid), dim3(512), 0, 0, buf + (rep % 3) * small4, rows, out);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best)
                best = ms;
        }
        printf("  1.076 GB as rows, one region per workgroup, rotating over 3 regions: %8.1f us  %7.1f GB/s\n", best * 1e3,
               (double)rows * grid * 4096 / (best * 1e-3) / 1e9);
    }
    return 0;
}
