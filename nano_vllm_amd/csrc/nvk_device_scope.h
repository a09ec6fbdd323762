// nvk_device_scope.h -- makes a device current for a scope (HIP translation units only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include "nvk_internal.h"

namespace nvk {

// Makes `device` current and restores the caller's device after; when it already is current (the usual case: one
// process per GPU) nothing is set or restored -- the entry points are called hundreds of times per model step.
struct DeviceScope
{
    int  previous = -1;
    bool ok       = true;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&previous) != hipSuccess)
            previous = -1;
        if (previous == device)
            previous = -1; // nothing to do, nothing to undo
        else
            ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (previous >= 0)
            (void)hipSetDevice(previous);
    }
    DeviceScope(const DeviceScope &)            = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

} // namespace nvk

// Every entry point that works on a device opens with this: a failed switch is an error here, never a launch on
// whatever device happened to be current (pointers of device A used on device B fail far from their cause).
#define NVK_SCOPE(device)                                                                                              \
    nvk::DeviceScope scope(device);                                                                                    \
    if (!scope.ok)                                                                                                     \
    return nvk::fail(NVK_ERR_HIP, "%s: hipSetDevice(%d) failed", __func__, (int)(device))

namespace nvk {

// hipGetDeviceCount, asked once per process (0 when no device is visible or the runtime fails)
inline int device_count_cached()
{
    static std::atomic<int> cached{0};
    const int               seen = cached.load(std::memory_order_relaxed);
    if (seen > 0)
        return seen;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return 0; // keep asking while there is none
    cached.store(n, std::memory_order_relaxed);
    return n;
}

} // namespace nvk
