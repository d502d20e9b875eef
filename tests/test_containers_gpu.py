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


def test_vector_abi_surface_matches_reference_signatures(pol):
    """get_val / set_val (no index) vs get_val_i / set_val_i, copy_to / copy_from (assignVals / retrieveVals),
    get_handle, pyview, get_iterator_1 / _3 -- py_interop/VectorInstantiations.cpp:8-170."""
    import ctypes as C
    from zpc_amd import lib, Port
    from zpc_amd.containers import Allocator
    L = lib()
    a = Allocator()
    v = L.container__v_float(a._h, 12)
    L.reset_container__v_float(v, 0)
    L.set_val_container__v_float(v, C.c_float(1.5))
    L.set_val_i_container__v_float(v, 5, C.c_float(-2.0))
    assert L.get_val_container__v_float(v) == 1.5 and L.get_val_i_container__v_float(v, 5) == -2.0
    src = np.arange(12, dtype=np.float32) * 0.5
    L.copy_to_container__v_float(v, src.ctypes.data)
    dst = np.zeros(12, np.float32)
    L.copy_from_container__v_float(v, dst.ctypes.data)
    assert np.array_equal(src, dst)
    h = L.get_handle_container__v_float(v)
    pv = L.pyview__v_float(v)
    assert pv[0] == h == L.container_data__v_float(v)
    L.del_pyview__v_float(pv)
    it1, it3 = L.get_iterator_1__v_float(v, 4), L.get_iterator_3__v_const_float(v, 2)
    assert (it1.base, it1.idx, it1.numTileBits, it1.tileMask, it1.numChns) == (h, 4, 0, 0, 1)
    assert (it3.base, it3.idx, it3.numTileBits, it3.tileMask, it3.numChns) == (h, 2, 0, 0, 3)
    # reduce over the iterator: sum of elements [4, 12)
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    end = Port(it1.base, 12, 0, 0, 1)
    L.reduce_sum__rocm_float_1(pol.handle, it1, end, Port(out.data_ptr(), 0, 0, 0, 1))
    assert out.item() == src[4:].sum()
    L.del_container__v_float(v)


def test_virtual_allocator_keeps_pointer_on_resize(pol):
    """allocator_virtual (py_interop/Allocator.cpp:14-19): a Vector over reserved address space grows by mapping memory
    behind the SAME pointer; contents survive; the plain allocator moves."""
    from zpc_amd import lib
    from zpc_amd.containers import Allocator, Vector
    L = lib()
    va = Allocator(virtual_reserve=1 << 30)
    v = L.container__v_int_virtual(va._h, 1000)
    p0 = L.get_handle_container__v_int_virtual(v)
    src = np.arange(1000, dtype=np.int32)
    L.copy_to_container__v_int_virtual(v, src.ctypes.data)
    L.resize_container__v_int_virtual(v, 5_000_000)
    assert L.container_size__v_int_virtual(v) == 5_000_000 and L.container_capacity__v_int_virtual(v) >= 5_000_000
    p1 = L.get_handle_container__v_int_virtual(v)
    assert L.get_val_i_container__v_int_virtual(v, 999) == 999 and L.get_val_i_container__v_int_virtual(v, 17) == 17
    L.set_val_i_container__v_int_virtual(v, 4_999_999, 7)
    assert L.get_val_i_container__v_int_virtual(v, 4_999_999) == 7
    if p0 == p1:  # virtual memory management available on this runtime: nothing moved
        # the grown vector is usable by kernels through its (unchanged) handle
        t = torch.zeros(1, dtype=torch.int32, device="cuda")
        import zpc_amd as zs
        from zpc_amd import Port
        L.reduce_sum__rocm_int_1(pol.handle, Port(p1, 0, 0, 0, 1), Port(p1, 1000, 0, 0, 1), Port(t.data_ptr(), 0, 0, 0, 1))
        assert t.item() == src.sum()
    L.del_container__v_int_virtual(v)
    pv = Vector("int", 1000)
    q0 = pv.data()
    pv.resize(5_000_000)
    assert pv.data() != q0 or pv.capacity() >= 5_000_000


def test_tilevector_views_and_vec3_iterators(pol):
    """pyview__tv / pyview__tvn (device-side tag tables), get_iterator_3, property_tags_get_* --
    py_interop/TileVectorInstantiations.cpp:8-215."""
    import ctypes as C
    from zpc_amd import lib
    from zpc_amd.containers import TileVector
    L = lib()
    tv = TileVector("float", 32, [("m", 1), ("x", 3), ("v", 3)], 100)
    v = L.pyview__tv_float_32(tv._h)
    assert v.contents._vector == tv.data() and v.contents._numChannels == 7
    L.del_pyview__tv_float_32(v)
    nv = L.pyview__tvn_const_float_32(tv._h)
    c = nv.contents
    assert c._N == 3 and c._numChannels == 7
    hip = C.CDLL("libamdhip64.so")
    names = (C.c_char * 96)()
    offs, sizes = (C.c_int * 3)(), (C.c_int * 3)()
    hip.hipMemcpy(names, C.c_void_p(c._tagNames), 96, 2)
    hip.hipMemcpy(offs, C.c_void_p(c._tagOffsets), 12, 2)
    hip.hipMemcpy(sizes, C.c_void_p(c._tagSizes), 12, 2)
    assert [names.raw[32 * i:32 * i + 32].split(b"\0")[0] for i in range(3)] == [b"m", b"x", b"v"]
    assert list(offs) == [0, 1, 4] and list(sizes) == [1, 3, 3]
    L.del_pyview__tvn_const_float_32(nv)
    it = L.get_iterator_3__tv_float_32(tv._h, 5, 1)
    assert it.base == tv.data() + 1 * 32 * 4 and (it.idx, it.numTileBits, it.tileMask, it.numChns) == (5, 5, 31, 7)
    t = TileVector._make_tags([("a", 3), ("bb", 2)])
    assert L.property_tags_get_size(t) == 2
    nm, sz_ = C.c_char_p(), C.c_size_t()
    L.property_tags_get_item(t, 1, C.byref(nm), C.byref(sz_))
    assert nm.value == b"bb" and sz_.value == 2
    L.del_property_tags(t)
    # _virtual spelling creates the same kind of object
    from zpc_amd.containers import Allocator
    va = Allocator(virtual_reserve=1 << 28)
    t2 = TileVector._make_tags([("q", 2)])
    h = L.container__tv_int_8_virtual(va._h, t2, 100)
    assert L.container_size__tv_int_8_virtual(h) == 100 and L.property_size__tv_int_8_virtual(h, b"q") == 2
    L.resize_container__tv_int_8_virtual(h, 100_000)
    assert L.container_size__tv_int_8_virtual(h) == 100_000
    L.del_container__tv_int_8_virtual(h)
    L.del_property_tags(t2)
