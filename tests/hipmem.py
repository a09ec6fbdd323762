"""Minimal device-memory helper for tests: hipMalloc/hipMemcpy/hipFree through ctypes, on the HIP
runtime libnvk_paged_attn.so already loaded (so the tests need no torch and mix no two runtimes)."""
import ctypes as C

import numpy as np

import nano_vllm_amd  # noqa: F401  (loads libamdhip64 via the product library)

_hip = C.CDLL("libamdhip64.so")
_hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_hip.hipFree.argtypes = [C.c_void_p]
_hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_hip.hipDeviceSynchronize.argtypes = []
H2D, D2H = 1, 2


class DeviceArray:
    def __init__(self, host: np.ndarray):
        host = np.ascontiguousarray(host)
        self.shape, self.dtype, self.nbytes = host.shape, host.dtype, host.nbytes
        self.ptr = C.c_void_p()
        assert _hip.hipMalloc(C.byref(self.ptr), max(self.nbytes, 16)) == 0
        if self.nbytes:
            assert _hip.hipMemcpy(self.ptr, C.c_void_p(host.ctypes.data), self.nbytes, H2D) == 0

    def data_ptr(self) -> int:  # what nano_vllm_amd's device-pointer entry points look for
        return int(self.ptr.value)

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        assert _hip.hipDeviceSynchronize() == 0
        if self.nbytes:
            assert _hip.hipMemcpy(C.c_void_p(out.ctypes.data), self.ptr, self.nbytes, D2H) == 0
        return out

    def free(self):
        if self.ptr:
            _hip.hipFree(self.ptr)
            self.ptr = C.c_void_p()

    __del__ = free


# ---- streams and graphs (tests of capture-safety of the device-pointer entry points) ----
_hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
_hip.hipStreamDestroy.argtypes = [C.c_void_p]
_hip.hipStreamSynchronize.argtypes = [C.c_void_p]
_hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
_hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
_hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
_hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
_hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
_hip.hipGraphDestroy.argtypes = [C.c_void_p]


class Stream:
    def __init__(self):
        self.h = C.c_void_p()
        assert _hip.hipStreamCreate(C.byref(self.h)) == 0

    def data_ptr(self) -> int:
        return int(self.h.value)

    def synchronize(self):
        assert _hip.hipStreamSynchronize(self.h) == 0

    def capture(self, fn):
        """Runs fn() under stream capture and returns an instantiated graph (hipGraphExec_t)."""
        assert _hip.hipStreamBeginCapture(self.h, 0) == 0  # hipStreamCaptureModeGlobal
        try:
            fn()
        finally:
            graph = C.c_void_p()
            rc = _hip.hipStreamEndCapture(self.h, C.byref(graph))
        assert rc == 0, f"hipStreamEndCapture -> {rc}"
        exe = C.c_void_p()
        assert _hip.hipGraphInstantiate(C.byref(exe), graph, None, None, 0) == 0
        _hip.hipGraphDestroy(graph)
        return exe

    def launch(self, exe):
        assert _hip.hipGraphLaunch(exe, self.h) == 0

    def close(self):
        if self.h:
            _hip.hipStreamDestroy(self.h)
            self.h = C.c_void_p()
