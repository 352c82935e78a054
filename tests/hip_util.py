"""Device memory for the GPU tests without torch: hipMalloc / hipMemcpy of the HIP runtime the product library itself runs on
(ctypes; torch, if some other test imported it, brings a second copy of the runtime, which is not the one picked)."""
import ctypes

import numpy as np

import rocket_path_amd as rp


def _hip():
    rp.load_library()
    paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}, key=lambda p: "torch" in p)
    return ctypes.CDLL(paths[0])


class DeviceBuffer:
    """nbytes of device memory, optionally filled with a byte; .ptr is the address, .read(dtype) copies it to the host."""

    def __init__(self, nbytes, fill=None, offset=0):
        self.hip, self.nbytes, self.offset = _hip(), int(nbytes), int(offset)
        self._raw = ctypes.c_void_p()
        assert self.hip.hipMalloc(ctypes.byref(self._raw), ctypes.c_size_t(self.nbytes + self.offset)) == 0
        if fill is not None:
            assert self.hip.hipMemset(self._raw, int(fill), ctypes.c_size_t(self.nbytes + self.offset)) == 0
        self.ptr = self._raw.value + self.offset

    def read(self, dtype):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(self.ptr), ctypes.c_size_t(out.nbytes), 2) == 0
        return out

    def write(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        assert self.hip.hipMemcpy(ctypes.c_void_p(self.ptr), ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes), 1) == 0

    def close(self):
        if self._raw:
            self.hip.hipFree(self._raw)
            self._raw = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
