"""zs::reduce / exclusive_scan / inclusive_scan / radix_sort / radix_sort_pair
(include/zensim/execution/ExecutionPolicy.hpp:684-781) on device arrays.

Arguments are "device arrays": anything with .data_ptr(), .numel() and .dtype (torch tensors) -- torch is
only the owner of the device memory, the work is done by libzsrocm's HIP kernels.  `Iter` wraps a
strided/AoSoA view exactly like the reference's aosoa_iterator (py_interop/GenericIterator.hpp).
"""
import ctypes as C

from ._lib import lib, Port

plus, multiplies, getmin, getmax = 0, 1, 2, 3  # ZpcFunctional.hpp functors -> op codes of zs_rocm.h

_SUFFIX = {"torch.int32": "i32", "torch.int64": "i64", "torch.float32": "f32", "torch.float64": "f64"}
_CT = {"i32": C.c_int32, "i64": C.c_int64, "f32": C.c_float, "f64": C.c_double}
_CNAME = {"torch.int32": "int", "torch.float32": "float", "torch.float64": "double"}
_ESIZE = {"torch.int32": 4, "torch.float32": 4, "torch.float64": 8, "torch.int64": 8, "torch.uint32": 4,
          "torch.uint64": 8}


class Iter:
    """aosoa_iterator over a device buffer: AoS (`Iter.aos`) or one TileVector channel (`Iter.aosoa`)."""

    def __init__(self, tensor, base_elem_offset, idx, num_tile_bits, tile_mask, num_chns):
        self.tensor = tensor
        es = _ESIZE[str(tensor.dtype)]
        self.port = Port(tensor.data_ptr() + base_elem_offset * es, idx, num_tile_bits, tile_mask, num_chns)
        self.dtype = tensor.dtype

    @staticmethod
    def aos(tensor, idx=0, num_chns=1, chn=0):
        # GenericIterator.hpp:71-72 (numTileBits = tileMask = 0, numChns = extent)
        return Iter(tensor, chn, idx, 0, 0, num_chns)

    @staticmethod
    def aosoa(tensor, idx, tile_size, chn_offset, num_chns):
        # GenericIterator.hpp:76-82: base = ptr + chnOffset * tileSize
        bits = tile_size.bit_length() - 1
        assert (1 << bits) == tile_size
        return Iter(tensor, chn_offset * tile_size, idx, bits, tile_size - 1, num_chns)

    def advanced(self, k):
        p = Port(self.port.base, self.port.idx + k, self.port.numTileBits, self.port.tileMask, self.port.numChns)
        it = Iter.__new__(Iter)
        it.tensor, it.port, it.dtype = self.tensor, p, self.dtype
        return it


def _as_iter(x, idx=0):
    if isinstance(x, Iter):
        return x
    return Iter.aos(x, idx)


def _check(pol):
    if pol is None:
        raise TypeError("an execution policy is required")
    return pol.handle


def reduce(pol, first, last_or_n, out, init=None, op=plus):
    """zs::reduce(pol, first, last, d_first, init, op) -- execution/ExecutionPolicy.hpp:721-728."""
    L = lib()
    if isinstance(first, Iter):
        n = last_or_n.port.idx - first.port.idx if isinstance(last_or_n, Iter) else int(last_or_n)
        if init is not None:
            raise ValueError("iterator form uses the C-ABI inits (0, 1, max, lowest)")
        cname = _CNAME[str(first.dtype)]
        name = {plus: "reduce_sum", multiplies: "reduce_prod", getmin: "reduce_min", getmax: "reduce_max"}[op]
        getattr(L, "%s__rocm_%s_1" % (name, cname))(_check(pol), first.port, first.advanced(n).port, _as_iter(out).port)
        return out
    n = first.numel() if last_or_n is None else int(last_or_n)
    S = _SUFFIX[str(first.dtype)]
    if init is None:
        import torch
        info = torch.finfo(first.dtype) if first.dtype.is_floating_point else torch.iinfo(first.dtype)
        init = {plus: 0, multiplies: 1, getmin: info.max, getmax: info.min}[op]
    getattr(L, "zs_rocm_reduce_" + S)(_check(pol), first.data_ptr(), n, out.data_ptr(), _CT[S](init), op)
    return out


def _scan(pol, src, n, out, init, op, exclusive):
    L = lib()
    if isinstance(src, Iter):
        cname = _CNAME[str(src.dtype)]
        kind = ("exclusive_scan_" if exclusive else "inclusive_scan_") + {plus: "sum", multiplies: "prod"}[op]
        getattr(L, "%s__rocm_%s_1" % (kind, cname))(_check(pol), src.port, src.advanced(n).port, _as_iter(out).port)
        return out
    S = _SUFFIX[str(src.dtype)]
    if init is None:
        init = {plus: 0, multiplies: 1}.get(op, 0)
    getattr(L, "zs_rocm_scan_" + S)(_check(pol), src.data_ptr(), n, out.data_ptr(), _CT[S](init), op, int(exclusive))
    return out


def exclusive_scan(pol, src, out, n=None, init=None, op=plus):
    """zs::exclusive_scan -- execution/ExecutionPolicy.hpp:706-716."""
    return _scan(pol, src, src.numel() if n is None else n, out, init, op, True)


def inclusive_scan(pol, src, out, n=None, op=plus):
    """zs::inclusive_scan -- execution/ExecutionPolicy.hpp:698-704."""
    return _scan(pol, src, src.numel() if n is None else n, out, None, op, False)


_SORT = {"torch.int32": "i32", "torch.uint32": "u32", "torch.int64": "i64", "torch.uint64": "u64"}


def radix_sort(pol, keys_in, keys_out, n=None, sbit=0, ebit=None):
    """zs::radix_sort(pol, first, last, d_first, sbit, ebit) -- execution/ExecutionPolicy.hpp:777-781."""
    L = lib()
    if isinstance(keys_in, Iter):
        cname = _CNAME[str(keys_in.dtype)]
        getattr(L, "radix_sort__rocm_%s_1" % cname)(_check(pol), keys_in.port, keys_in.advanced(n).port,
                                                    _as_iter(keys_out).port)
        return keys_out
    n = keys_in.numel() if n is None else n
    S = _SORT[str(keys_in.dtype)]
    if ebit is None:
        ebit = _ESIZE[str(keys_in.dtype)] * 8
    getattr(L, "zs_rocm_radix_sort_" + S)(_check(pol), keys_in.data_ptr(), None, keys_out.data_ptr(), None, n, sbit, ebit)
    return keys_out


def radix_sort_pair(pol, keys_in, vals_in, keys_out, vals_out, n=None, sbit=0, ebit=None):
    """zs::radix_sort_pair -- execution/ExecutionPolicy.hpp:765-775."""
    L = lib()
    if isinstance(keys_in, Iter):
        cname = _CNAME[str(keys_in.dtype)]
        getattr(L, "radix_sort_pair__rocm_%s_1" % cname)(_check(pol), keys_in.port, _as_iter(vals_in).port,
                                                         _as_iter(keys_out).port, _as_iter(vals_out).port, n)
        return keys_out, vals_out
    n = keys_in.numel() if n is None else n
    S = _SORT[str(keys_in.dtype)]
    if ebit is None:
        ebit = _ESIZE[str(keys_in.dtype)] * 8
    getattr(L, "zs_rocm_radix_sort_" + S)(_check(pol), keys_in.data_ptr(), vals_in.data_ptr(), keys_out.data_ptr(),
                                          vals_out.data_ptr(), n, sbit, ebit)
    return keys_out, vals_out


_MSORT = {"torch.int32": "i32", "torch.uint32": "u32", "torch.int64": "i64", "torch.uint64": "u64", "torch.float32": "f32",
          "torch.float64": "f64"}


def merge_sort(pol, keys, n=None, descending=False):
    """zs::merge_sort(pol, first, last[, comp]) -- stable, in place (execution/ExecutionPolicy.hpp:755-761).
    Iterator form = the C ABI `merge_sort__rocm_T_1(pol, first, last)` (comparator `<`)."""
    L = lib()
    if isinstance(keys, Iter):
        cname = _CNAME[str(keys.dtype)]
        getattr(L, "merge_sort__rocm_%s_1" % cname)(_check(pol), keys.port, keys.advanced(n).port)
        return keys
    n = keys.numel() if n is None else n
    getattr(L, "zs_rocm_merge_sort_" + _MSORT[str(keys.dtype)])(_check(pol), keys.data_ptr(), None, n, int(descending))
    return keys


def merge_sort_pair(pol, keys, vals, n=None, descending=False):
    """zs::merge_sort_pair(pol, keys, vals, count[, comp]) -- stable, in place, int32 values
    (execution/ExecutionPolicy.hpp:745-753; C ABI `merge_sort_pair__rocm_T_1(pol, keys, vals, count)`)."""
    L = lib()
    if isinstance(keys, Iter):
        cname = _CNAME[str(keys.dtype)]
        getattr(L, "merge_sort_pair__rocm_%s_1" % cname)(_check(pol), keys.port, _as_iter(vals).port, n)
        return keys, vals
    n = keys.numel() if n is None else n
    getattr(L, "zs_rocm_merge_sort_" + _MSORT[str(keys.dtype)])(_check(pol), keys.data_ptr(), vals.data_ptr(), n, int(descending))
    return keys, vals
