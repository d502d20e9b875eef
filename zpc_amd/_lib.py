"""ctypes binding of libzsrocm.so (include/zs_rocm.h).  The product path: if the HIP library is missing
this module raises -- there is no CPU fallback anywhere in zpc_amd."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzsrocm.so")


class Port(C.Structure):
    """aosoa_iterator_port (py_interop/GenericIterator.hpp:11-16), passed by value."""
    _fields_ = [("base", C.c_void_p), ("idx", C.c_uint32), ("numTileBits", C.c_uint32),
                ("tileMask", C.c_uint32), ("numChns", C.c_uint32)]


class BhtViewLite(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("indices", C.c_void_p), ("status", C.c_void_p), ("activeKeys", C.c_void_p),
                ("cnt", C.c_void_p), ("success", C.c_void_p), ("tableSize", C.c_size_t),
                ("hf0x", C.c_uint32), ("hf0y", C.c_uint32), ("hf1x", C.c_uint32), ("hf1y", C.c_uint32),
                ("hf2x", C.c_uint32), ("hf2y", C.c_uint32)]


class Particles(C.Structure):
    _fields_ = [("mass", Port), ("pos", Port), ("vel", Port), ("C", Port), ("F", Port), ("logJp", Port),
                ("n", C.c_size_t)]


class MpmParams(C.Structure):
    _fields_ = [("model", C.c_int), ("dx", C.c_float), ("dt", C.c_float), ("volume", C.c_float), ("E", C.c_float),
                ("nu", C.c_float), ("cohesion", C.c_float), ("beta", C.c_float), ("yieldSurface", C.c_float),
                ("volCorrection", C.c_int), ("side", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libzsrocm.so not built (%s): run `python -m zpc_amd.build` -- zpc_amd has no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    L.policy__device.restype = vp
    L.del_policy__device.argtypes = [vp]
    for n in ("sync", "profile", "device", "stream", "block"):
        getattr(L, "zs_rocm_policy_" + n).argtypes = [vp, i32]
    L.zs_rocm_policy_listen.argtypes = [vp, i32, i32]
    L.zs_rocm_policy_shmem.argtypes = [vp, sz]
    L.zs_rocm_policy_external_stream.argtypes = [vp, vp]
    L.zs_rocm_policy_get_stream.argtypes = [vp]
    L.zs_rocm_policy_get_stream.restype = vp
    L.zs_rocm_policy_should_sync.argtypes = [vp]
    L.zs_rocm_policy_sync_ctx.argtypes = [vp]
    L.zs_rocm_policy_last_elapsed_ms.argtypes = [vp]
    L.zs_rocm_policy_last_elapsed_ms.restype = f32
    L.zs_rocm_last_error.argtypes = [i32]
    L.zs_rocm_clear_error.argtypes = [i32]
    L.launch__device.argtypes = [vp, vp, sz, vp]
    for T in ("int", "float", "double"):
        for op in ("reduce_sum", "reduce_prod", "reduce_min", "reduce_max", "exclusive_scan_sum",
                   "exclusive_scan_prod", "inclusive_scan_sum", "inclusive_scan_prod"):
            getattr(L, "%s__rocm_%s_1" % (op, T)).argtypes = [vp, Port, Port, Port]
        getattr(L, "radix_sort__rocm_%s_1" % T).argtypes = [vp, Port, Port, Port]
        getattr(L, "radix_sort_pair__rocm_%s_1" % T).argtypes = [vp, Port, Port, Port, Port, sz]
    for S, ct in (("i32", C.c_int32), ("i64", C.c_int64), ("f32", C.c_float), ("f64", C.c_double)):
        getattr(L, "zs_rocm_reduce_" + S).argtypes = [vp, vp, sz, vp, ct, i32]
        getattr(L, "zs_rocm_scan_" + S).argtypes = [vp, vp, sz, vp, ct, i32, i32]
    for S in ("i32", "u32", "i64", "u64"):
        getattr(L, "zs_rocm_radix_sort_" + S).argtypes = [vp, vp, vp, vp, vp, sz, i32, i32]
