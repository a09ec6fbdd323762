// nvk_device_scope.h -- makes a device current for a scope (HIP translation units only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <mutex>

#include "nvk_internal.h"

namespace nvk {

// The recording gate.  A model records its decode step as a hipGraph (stream capture, nvk_model_step_chunks).  While a
// stream of the process is capturing, ROCm 7.2 refuses work that other threads put on the null stream ("operation would
// make the legacy stream depend on a capturing blocking stream" -- also for a non-blocking capturing stream, in
// thread-local capture mode), refuses first-use work such as allocations, and any such call invalidates the recording and
// leaves the recording stream unusable for good.  So the library keeps its own calls out of each other's way: every
// entry point that touches the device holds the gate shared for its duration (the outermost one of a thread: entry
// points call each other), a recording holds it alone, for the microseconds it takes.  Writers go first, so a recording
// cannot starve behind threads that step back to back.  (What the APPLICATION does on other threads during a recording
// is outside the library's reach: INTEGRATION.md.)
class RecordGate
{
    std::mutex              m_;
    std::condition_variable cv_;
    int                     readers_ = 0, writers_waiting_ = 0;
    bool                    writer_ = false;

  public:
    void lock_shared()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !writer_ && writers_waiting_ == 0; });
        ++readers_;
    }
    void unlock_shared()
    {
        std::lock_guard<std::mutex> l(m_);
        if (--readers_ == 0)
            cv_.notify_all();
    }
    void lock()
    {
        std::unique_lock<std::mutex> l(m_);
        ++writers_waiting_;
        cv_.wait(l, [&] { return !writer_ && readers_ == 0; });
        --writers_waiting_;
        writer_ = true;
    }
    void unlock()
    {
        std::lock_guard<std::mutex> l(m_);
        writer_ = false;
        cv_.notify_all();
    }
};
inline RecordGate &record_gate()
{
    static RecordGate gate;
    return gate;
}
inline thread_local int g_gate_depth = 0; // entry points of this thread that are inside NVK_SCOPE

// An entry point's shared hold of the gate (the outermost one of the thread takes it).
struct GateHold
{
    GateHold()
    {
        if (g_gate_depth++ == 0)
            record_gate().lock_shared();
    }
    ~GateHold()
    {
        if (--g_gate_depth == 0)
            record_gate().unlock_shared();
    }
    GateHold(const GateHold &)            = delete;
    GateHold &operator=(const GateHold &) = delete;
};

// Held by the thread that records: gives up the thread's shared hold, takes the gate alone, and puts things back.
struct RecordAlone
{
    RecordAlone()
    {
        if (g_gate_depth > 0)
            record_gate().unlock_shared();
        record_gate().lock();
    }
    ~RecordAlone()
    {
        record_gate().unlock();
        if (g_gate_depth > 0)
            record_gate().lock_shared();
    }
    RecordAlone(const RecordAlone &)            = delete;
    RecordAlone &operator=(const RecordAlone &) = delete;
};

// Makes `device` current and restores the caller's device after; when it already is current (the usual case: one
// process per GPU) nothing is set or restored -- the entry points are called hundreds of times per model step.
struct DeviceScope
{
    GateHold hold; // (first: taken before the device is touched, given back last)
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
