// shard_sync_probe.hip -- what the cross-stream hand-over of nvk_shard_decode_gather costs, protocol by protocol.
//
//   hipcc --offload-arch=gfx950 -O2 -o build/shard_sync_probe tools/shard_sync_probe.hip && build/shard_sync_probe [us]
//
// A "layer-call" here is a stand-in: a kernel that spins for `us` microseconds on the compute stream, then a 128 KiB
// device-to-device copy (what a one-rank all-gather is) on the communication stream, double-buffered, the consumer of
// copy i waiting one step behind -- the shape of bench.py --shard heads.  Variants of the hand-over:
//   plain        the kernels alone (no second stream): the floor
//   events       hipEventRecord / hipStreamWaitEvent both ways (round 2's protocol)
//   events-dedup the same, minus the wait a stream has already done
//   values       hipStreamWriteValue32 / hipStreamWaitValue32 on two counters in signal memory
//   values-dedup the same, minus the wait a stream has already done
//   inline       the copy on the compute stream itself (no hand-over, no overlap)
// Prints us per step for each.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess) {                                                                                         \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                                                            \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

__global__ void spin_kernel(float *out, long long ticks) // 100 MHz timestamp ticks
{
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    }
    if (threadIdx.x == 0)
        out[blockIdx.x] = 1.f;
}

int main(int argc, char **argv)
{
    const double us     = argc > 1 ? atof(argv[1]) : 26.0;
    const int    steps  = 400;
    const size_t shard  = 128 << 10;
    float       *out[2], *gath[2];
    for (int b = 0; b < 2; b++) {
        CK(hipMalloc(&out[b], shard));
        CK(hipMalloc(&gath[b], shard));
    }
    hipStream_t cs, comm;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&comm, hipStreamNonBlocking));
    hipEvent_t computed[2], gathered[2], t0, t1;
    for (int b = 0; b < 2; b++) {
        CK(hipEventCreateWithFlags(&computed[b], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&gathered[b], hipEventDisableTiming));
    }
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    uint32_t *flag_c = nullptr, *flag_g = nullptr; // layer-calls computed / gathered (signal memory comes in 8-byte pieces)
    if (can) {
        CK(hipExtMallocWithFlags((void **)&flag_c, 8, hipMallocSignalMemory));
        CK(hipExtMallocWithFlags((void **)&flag_g, 8, hipMallocSignalMemory));
    }
    printf("stand-in kernel %.0f us, %d steps; hipStreamWaitValue32 %s\n", us, steps, can ? "available" : "NOT available");

    const long long ticks  = (long long)(us * 100.0);
    auto            kernel = [&](int b) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, cs, out[b], ticks); };

    for (int mode = 0; mode < 6; mode++) {
        const char *names[] = {"plain", "events", "events-dedup", "values", "values-dedup", "inline"};
        if ((mode == 3 || mode == 4) && !can)
            continue;
        if (flag_c) {
            CK(hipMemset(flag_c, 0, 8));
            CK(hipMemset(flag_g, 0, 8));
        }
        CK(hipDeviceSynchronize());
        bool pending[2] = {false, false}, cs_waited[2] = {false, false};
        uint32_t call_of[2] = {0, 0};
        uint32_t n = 0;
        auto step = [&](int i) {
            const int b = i & 1;
            n++;
            switch (mode) {
            case 0: kernel(b); break;
            case 1:
            case 2:
                if (pending[b] && !(mode == 2 && cs_waited[b]))
                    CK(hipStreamWaitEvent(cs, gathered[b], 0));
                kernel(b);
                CK(hipEventRecord(computed[b], cs));
                CK(hipStreamWaitEvent(comm, computed[b], 0));
                CK(hipMemcpyAsync(gath[b], out[b], shard, hipMemcpyDeviceToDevice, comm));
                CK(hipEventRecord(gathered[b], comm));
                pending[b] = true, cs_waited[b] = false;
                if (i > 0) { // the consumer of the previous layer-call's rows
                    CK(hipStreamWaitEvent(cs, gathered[1 - b], 0));
                    cs_waited[1 - b] = true;
                }
                break;
            case 3:
            case 4:
                if (pending[b] && !(mode == 4 && cs_waited[b]))
                    CK(hipStreamWaitValue32(cs, flag_g, call_of[b], hipStreamWaitValueGte, 0xffffffffu));
                kernel(b);
                CK(hipStreamWriteValue32(cs, flag_c, n, 0));
                CK(hipStreamWaitValue32(comm, flag_c, n, hipStreamWaitValueGte, 0xffffffffu));
                CK(hipMemcpyAsync(gath[b], out[b], shard, hipMemcpyDeviceToDevice, comm));
                CK(hipStreamWriteValue32(comm, flag_g, n, 0));
                pending[b] = true, cs_waited[b] = false, call_of[b] = n;
                if (i > 0) {
                    CK(hipStreamWaitValue32(cs, flag_g, call_of[1 - b], hipStreamWaitValueGte, 0xffffffffu));
                    cs_waited[1 - b] = true;
                }
                break;
            case 5:
                kernel(b);
                CK(hipMemcpyAsync(gath[b], out[b], shard, hipMemcpyDeviceToDevice, cs));
                break;
            }
        };
        for (int i = 0; i < 20; i++)
            step(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, cs));
        for (int i = 0; i < steps; i++)
            step(i);
        CK(hipEventRecord(t1, cs));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, t0, t1));
        printf("%-13s %8.2f us per step\n", names[mode], ms * 1e3 / steps);
    }
    return 0;
}
