// What is left of a stream whose graph recording another thread invalidated, and what brings it back?  (ROCm 7.2, gfx950)
// Thread A records on a non-blocking stream in thread-local capture mode; thread B meanwhile does what applications do:
// a null-stream memset, an allocation, a device-wide wait.  Then A ends the recording and tries to use the stream.
//   hipcc --offload-arch=gfx950 -O2 -o build/probe/capture_recovery_probe tools/capture_recovery_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <thread>

__global__ void bump(int *p) { atomicAdd(p, 1); }
#define SAY(x) do { hipError_t e_ = (x); printf("  %-58s -> %s\n", #x, e_ == hipSuccess ? "ok" : hipGetErrorString(e_)); } while (0)

int main()
{
    int *d = nullptr, *other = nullptr;
    hipMalloc(&d, 4096);
    hipMalloc(&other, 4096);
    hipMemset(d, 0, 4096);
    for (int what = 0; what < 3; what++) {
        hipStream_t st;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, d);
        hipStreamSynchronize(st);
        std::atomic<int> phase{0};
        hipError_t       in_b = hipSuccess;
        std::thread      b([&] {
            while (phase.load() != 1) {}
            if (what == 0) in_b = hipMemset(other, 0, 4096);          // null stream
            else if (what == 1) { void *p = nullptr; in_b = hipMalloc(&p, 1 << 20); if (p) hipFree(p); }
            else in_b = hipDeviceSynchronize();
            phase.store(2);
        });
        printf("thread B: %s while thread A records\n", what == 0 ? "hipMemset on the null stream" : what == 1 ? "hipMalloc + hipFree" : "hipDeviceSynchronize");
        hipGraph_t g = nullptr;
        SAY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, d);
        phase.store(1);
        while (phase.load() != 2) {}
        printf("  (thread B's call returned: %s)\n", in_b == hipSuccess ? "ok" : hipGetErrorString(in_b));
        hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, d);
        SAY(hipGetLastError());
        SAY(hipStreamEndCapture(st, &g));
        b.join();
        hipStreamCaptureStatus cs;
        SAY(hipStreamIsCapturing(st, &cs));
        printf("  capture status now %d (0 none, 1 active, 2 invalidated)\n", (int)cs);
        SAY(hipMemcpyAsync(other, d, 64, hipMemcpyDeviceToDevice, st));
        (void)hipGetLastError();
        SAY(hipStreamEndCapture(st, &g));                     // a second time
        SAY(hipMemcpyAsync(other, d, 64, hipMemcpyDeviceToDevice, st));
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        SAY(hipThreadExchangeStreamCaptureMode(&mode));
        SAY(hipMemcpyAsync(other, d, 64, hipMemcpyDeviceToDevice, st));
        SAY(hipStreamSynchronize(st));
        hipStream_t st2;                                       // a fresh stream on the same thread
        SAY(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        SAY(hipMemcpyAsync(other, d, 64, hipMemcpyDeviceToDevice, st2));
        SAY(hipStreamSynchronize(st2));
        SAY(hipStreamDestroy(st));
        SAY(hipMemcpyAsync(other, d, 64, hipMemcpyDeviceToDevice, st2));
        SAY(hipStreamSynchronize(st2));
        hipStreamDestroy(st2);
        (void)hipGetLastError();
    }
    return 0;
}
