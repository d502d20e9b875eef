"""GPU parity for Vector / TileVector behind the reference's py_interop-style C ABI and the raw AoSoA kernels.
Bit-exact copies; layout pinned by container/TileVector.hpp:108,397 through oracle/tilevector.c."""
import ctypes as C

import numpy as np
import pytest

from util import rng

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_vector_basic(pol):
    """test/cuda/basic.cu:65-107: fill on host vs device + clone compare; resize growth (Vector.hpp:228-256,407-416)."""
    from zpc_amd.containers import Vector, Allocator, memsrc_device, memsrc_host, memsrc_um
    v = Vector("int", 100, Allocator(memsrc_device, 0))
    assert v.size() == 100 and v.capacity() == 100
    v.reset(0)
    v.setVal(42, 7)
    assert v.getVal(7) == 42 and v.getVal(8) == 0
    v.resize(120)  # 100 -> max(150, 120)
    assert v.size() == 120 and v.capacity() == 150 and v.getVal(7) == 42
    v.resize(10)
    assert v.size() == 10 and v.capacity() == 150
    v.relocate(memsrc_host, -1)
    assert v.getVal(7) == 42
    v.relocate(memsrc_um, 0)
    assert v.getVal(7) == 42
    f = Vector("double", 3)
    f.setVal(2.5, 1)
    assert f.getVal(1) == 2.5


@pytest.mark.parametrize("L", [8, 32, 64, 512])
def test_tilevector_layout_and_iterators(pol, oracle, L):
    import zpc_amd as zs
    from zpc_amd.containers import TileVector
    from zpc_amd.primitives import Iter
    tags = [("a", 3), ("b", 2), ("c", 1)]  # test/utils/initialization.hpp:75
    n = 1000
    tv = TileVector("int", L, tags, n)
    assert tv.numChannels() == 6 and tv.size() == n
    assert [tv.getPropertyOffset(k) for k in ("a", "b", "c", "zz")] == [0, 3, 5, -1]
    tiles = (n + L - 1) // L
    total = tiles * L * 6
    host = rng(11).integers(-100, 100, total, dtype=np.int32)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy(C.c_void_p(tv.data()), host.ctypes.data_as(C.c_void_p), C.c_size_t(total * 4), 1)
    oracle.orc_tv_offset.restype = C.c_size_t
    it = tv.iterator("b")
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    it_last = type(it)(it.base, it.idx + n, it.numTileBits, it.tileMask, it.numChns)
    zs.lib().reduce_sum__rocm_int_1(pol.handle, it, it_last, Iter.aos(out).port)
    exp = sum(int(host[oracle.orc_tv_offset(C.c_size_t(i), C.c_size_t(3), C.c_size_t(L), C.c_size_t(6))]) for i in range(n))
    assert int(out.item()) == exp
    # append_channels (TileVector.hpp:583-623): old channels preserved, new ones zero
    tv.append_channels(pol, [("d", 4), ("b", 2)])
    assert tv.numChannels() == 10 and tv.getPropertyOffset("d") == 6
    new = np.empty(tiles * L * 10, np.int32)
    hip.hipMemcpy(new.ctypes.data_as(C.c_void_p), C.c_void_p(tv.data()), C.c_size_t(new.nbytes), 2)
    new = new.reshape(tiles, 10, L)
    old = host.reshape(tiles, 6, L)
    assert np.array_equal(new[:, :6], old) and not new[:, 6:].any()
    # fill + reorder
    tv.fill(pol, 7)
    hip.hipMemcpy(new.ctypes.data_as(C.c_void_p), C.c_void_p(tv.data()), C.c_size_t(new.nbytes), 2)
    assert (new.reshape(-1) == 7).all()


@pytest.mark.parametrize("L,Cn,n", [(32, 25, 1), (32, 25, 1000), (64, 25, 4097), (8, 3, 77), (512, 7, 5000)])
def test_aos_aosoa_roundtrip(pol, oracle, L, Cn, n):
    import zpc_amd as zs
    aos = rng(12).standard_normal((n, Cn)).astype(np.float32)
    tiles = (n + L - 1) // L
    d_aos = torch.from_numpy(aos).cuda()
    d_tv = torch.zeros(tiles * L * Cn, dtype=torch.float32, device="cuda")
    zs.lib().zs_rocm_tv_from_aos_f32(pol.handle, d_aos.data_ptr(), n, Cn, L, d_tv.data_ptr())
    exp = np.zeros(tiles * L * Cn, np.float32)
    oracle.orc_tv_from_aos_f32(aos.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_size_t(Cn), C.c_size_t(L), exp.ctypes.data_as(C.c_void_p))
    assert np.array_equal(d_tv.cpu().numpy(), exp)
    back = torch.empty_like(d_aos)
    zs.lib().zs_rocm_tv_to_aos_f32(pol.handle, d_tv.data_ptr(), n, Cn, L, back.data_ptr())
    assert np.array_equal(back.cpu().numpy(), aos)
    # scale = load all channels + store all channels
    zs.lib().zs_rocm_tv_scale_f32(pol.handle, d_tv.data_ptr(), n, Cn, L, C.c_float(2.0))
    assert np.array_equal(d_tv.cpu().numpy(), exp * 2)
    # gather by a permutation
    perm = rng(13).permutation(n).astype(np.int32)
    dst = torch.zeros_like(d_tv)
    zs.lib().zs_rocm_tv_gather_f32(pol.handle, d_tv.data_ptr(), dst.data_ptr(), n, Cn, L, torch.from_numpy(perm).cuda().data_ptr())
    back2 = torch.empty_like(d_aos)
    zs.lib().zs_rocm_tv_to_aos_f32(pol.handle, dst.data_ptr(), n, Cn, L, back2.data_ptr())
    assert np.array_equal(back2.cpu().numpy(), aos[perm] * 2)
