"""hiprtc-compiled user kernels through the reference's JIT entry points (py_interop/cuda/Nvrtc.cpp, ExecutionPolicy.cpp:11-39):
compile -> load module -> get kernel -> launch__device on the policy's stream."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SRC = r"""
extern "C" __global__ void saxpy(float a, const float *x, float *y, unsigned long n) {
  unsigned long i = blockIdx.x * (unsigned long)blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + y[i];
}
extern "C" __global__ void iota(int *p, unsigned long n) {
  unsigned long i = blockIdx.x * (unsigned long)blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int)i;
}
"""


def test_compile_load_launch(pol, tmp_path):
    from zpc_amd import jit, lib
    path = jit.compile_program(SRC, str(tmp_path / "m.hsaco"), arch=0)  # arch from the current device
    mod = jit.Module(pol, path)
    n = 100_003
    x = torch.arange(n, dtype=torch.float32, device="cuda")
    y = torch.ones(n, dtype=torch.float32, device="cuda")
    jit.launch(pol, mod.kernel("saxpy"), n, C.c_float(2.5), x.data_ptr(), y.data_ptr(), C.c_ulong(n))
    assert np.array_equal(y.cpu().numpy(), 2.5 * np.arange(n, dtype=np.float32) + 1)
    # on a spare stream, not synchronised by the call; then the raw launcher rocm_launch_kernel
    pol2 = type(pol)()
    pol2.sync(False).stream(3)
    p = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    jit.launch(pol2, mod.kernel("iota"), n, p.data_ptr(), C.c_ulong(n))
    pol2.syncCtx()
    assert np.array_equal(p.cpu().numpy(), np.arange(n, dtype=np.int32))
    q = torch.zeros(1000, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    a0, a1 = C.c_void_p(q.data_ptr()), C.c_ulong(1000)
    args = (C.c_void_p * 2)(C.cast(C.pointer(a0), C.c_void_p), C.cast(C.pointer(a1), C.c_void_p))
    assert lib().rocm_launch_kernel(None, mod.kernel("iota"), 1000, args, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), np.arange(1000, dtype=np.int32))
    with pytest.raises(KeyError):
        mod.kernel("nope")
    lib().zs_rocm_clear_error(-1)
    mod.unload()
