"""Runtime compilation for the rocm backend: the Python side of py_interop/cuda/Nvrtc.cpp (zpc_jit's `cuda_compile_program`,
`cuda_load_module`, `cuda_get_kernel`) over hiprtc, plus `launch` = `launch__device(pol, kernel, dim, args)`
(py_interop/cuda/ExecutionPolicy.cpp:11-39: block 128, grid ceil(dim/128), the policy's stream, sync if shouldSync())."""
import ctypes as C
import os
import tempfile

from ._lib import lib


def compile_program(src, output_path=None, arch=950, include_dir=".", debug=False, verbose=False, fast_math=False):
    """source text -> code object file; returns the path.  Raises RuntimeError with hiprtc's result code on failure."""
    if output_path is None:
        fd, output_path = tempfile.mkstemp(suffix=".hsaco")
        os.close(fd)
    rc = lib().rocm_compile_program(src.encode() if isinstance(src, str) else src, int(arch), include_dir.encode(), bool(debug),
                                    bool(verbose), False, bool(fast_math), output_path.encode())
    if rc != 0:
        raise RuntimeError("rocm_compile_program failed (hiprtc result %d)" % rc)
    return output_path


class Module:
    def __init__(self, pol, path):
        self.pol = pol
        self._m = lib().rocm_load_module(pol.handle, path.encode())
        if not self._m:
            raise RuntimeError("rocm_load_module failed for " + path)

    def kernel(self, name):
        k = lib().rocm_get_kernel(self.pol.handle, self._m, name.encode())
        if not k:
            raise KeyError(name)
        return k

    def unload(self):
        if self._m:
            lib().rocm_unload_module(self.pol.handle, self._m)
            self._m = None


def launch(pol, kernel, dim, *args):
    """launch__device(pol, kernel, dim, args): args are ctypes scalars / c_void_p device pointers, passed by address."""
    holders = [a if isinstance(a, C._SimpleCData) else C.c_void_p(a) for a in args]
    arr = (C.c_void_p * len(holders))(*[C.cast(C.pointer(h), C.c_void_p) for h in holders])
    lib().launch__device(pol.handle, kernel, dim, arr)
