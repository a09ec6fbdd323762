// What does one grid-wide barrier cost on this GPU?  (the question DESIGN.md section 9 leaves open for a one-request model step
// that runs as ONE resident kernel: its stages are separated by such barriers instead of by launches)
//   hipcc --offload-arch=gfx950 -O3 -o build/probe/grid_barrier_probe tools/grid_barrier_probe.hip
//   build/probe/grid_barrier_probe [workgroups-per-CU=2] [threads=512] [rounds=2000]
// Every workgroup writes a word, meets the others, reads a neighbour's word of the same round (checked), meets them again.
// Spins are bounded: a barrier that does not complete raises `*abort_flag` and every wave leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 3: as 2 without the two fences -- the data goes through to memory and is read from there (agent-scope relaxed
// atomic stores and loads: sc1), each thread waits for its own stores before it arrives.
// MODE 0: everyone polls the counter the arrivals add to.  MODE 1: the last arrival publishes the round in a word of its own
// (another 128-B line), the others poll that.  MODE 2: as 1, one counter per XCD (workgroup id % 8) and the last of an XCD
// arrives for it on the grid's counter.  `nap` = s_sleep argument between polls.
template <int MODE, int NAP>
__device__ __forceinline__ bool grid_meet(unsigned *count, unsigned *abort_flag, unsigned target)
{
    if (MODE == 3) __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this thread's stores through to memory have been acknowledged
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (MODE < 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned *flag = count + 32; // the published round (MODE 1, 2)
        const unsigned round = target / gridDim.x;
        bool           wait = true;
        if (MODE == 0)
            __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) {
            if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target - 1) {
                __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wait = false;
            }
        }
        else if (MODE >= 2) {
            unsigned *mine = count + 256 + 32 * (blockIdx.x & 7); // (gridDim.x % 8 == 0)
            if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target / 8 - 1)
                if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 8 * round - 1) {
                    __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    wait = false;
                }
        }
        int spins = 0;
        while (wait && (MODE == 0 ? __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target
                                  : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round)) {
            if (++spins > (1 << 20) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(NAP);
        }
        if (MODE < 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = ok;
    __syncthreads();
    return s_ok;
}

template <int MODE, int NAP>
__global__ void __launch_bounds__(1024) probe(unsigned *count, unsigned *abort_flag, unsigned *words, unsigned *bad, int rounds, int work)
{
    const unsigned g = gridDim.x;
    unsigned       target = 0;
    for (int r = 0; r < rounds; r++) {
        for (int i = threadIdx.x; i < work; i += blockDim.x)
            if (MODE == 3)
                __hip_atomic_store(&words[(size_t)blockIdx.x * work + i], r * 1000003u + blockIdx.x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                words[(size_t)blockIdx.x * work + i] = r * 1000003u + blockIdx.x + i;
        target += g;
        if (!grid_meet<MODE, NAP>(count, abort_flag, target)) return;
        const unsigned nb = (blockIdx.x + 37 * (r + 1)) % g;
        for (int i = threadIdx.x; i < work; i += blockDim.x)
            if ((MODE == 3 ? __hip_atomic_load(&words[(size_t)nb * work + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : words[(size_t)nb * work + i]) != r * 1000003u + nb + i)
                atomicAdd(bad, 1u);
        target += g;
        if (!grid_meet<MODE, NAP>(count, abort_flag, target)) return;
    }
}

int main(int argc, char **argv)
{
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2, threads = argc > 2 ? atoi(argv[2]) : 512, rounds = argc > 3 ? atoi(argv[3]) : 2000;
    const int work = argc > 4 ? atoi(argv[4]) : 4, mode = argc > 5 ? atoi(argv[5]) : 0, nap = argc > 6 ? atoi(argv[6]) : 1;
    void (*kern)(unsigned *, unsigned *, unsigned *, unsigned *, int, int) =
        mode == 0 ? (nap == 1 ? probe<0, 1> : probe<0, 16>) : mode == 1 ? (nap == 1 ? probe<1, 1> : probe<1, 16>) : mode == 2 ? (nap == 1 ? probe<2, 1> : probe<2, 16>) : (nap == 1 ? probe<3, 1> : probe<3, 16>);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int fit = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kern, threads, 0));
    if (fit < per_cu) { printf("only %d workgroups of %d threads fit a CU\n", fit, threads); return 1; }
    const int grid = per_cu * prop.multiProcessorCount;
    unsigned *ctl, *words;
    CHECK(hipMalloc(&ctl, 4096));
    CHECK(hipMalloc(&words, (size_t)grid * work * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipMemset(ctl, 0, 4096));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, ctl, ctl + 64, words, ctl + 128, rounds, work);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[256];
        CHECK(hipMemcpy(h, ctl, sizeof h, hipMemcpyDeviceToHost));
        printf("mode %d nap %d: %d workgroups x %d threads, %d words each: %d barriers in %.3f ms = %.2f us each; aborted %u, stale reads %u\n", mode, nap, grid, threads,
               work, 2 * rounds, ms, ms * 1e3 / (2 * rounds), h[64], h[128]);
    }
    return 0;
}
