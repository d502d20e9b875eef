"""Host mirrors of zs::Vector / zs::TileVector / zs::bht as the reference's ctypes layer sees them
(include/zensim/py_interop/{Vector,TileVector,Bht}Instantiations.cpp).  Thin handles over the C ABI."""
import ctypes as C

from ._lib import lib, Port

memsrc_host, memsrc_device, memsrc_um = 0, 1, 2  # memsrc_e (types/Property.h:7)
_CT = {"int": C.c_int, "float": C.c_float, "double": C.c_double}
_ES = {"int": 4, "float": 4, "double": 8}


class Allocator:
    """allocator(memsrc_e, ProcID) -- py_interop/Allocator.cpp:5-21"""

    def __init__(self, memsrc=memsrc_device, devid=None, virtual_reserve=0):
        """virtual_reserve > 0: allocator_virtual(mre, devid, reservedSpace) -- containers keep their data pointer on resize.
        devid None = the calling thread's current device (one process per GPU: rank k must not allocate on device 0)."""
        if devid is None:
            devid = lib().zs_rocm_current_device() if memsrc != memsrc_host else -1
        self.memsrc, self.devid = memsrc, devid
        self._h = lib().allocator_virtual(memsrc, devid, virtual_reserve) if virtual_reserve else lib().allocator(memsrc, devid)

    def __del__(self):
        try:
            lib().del_allocator(self._h)
        except Exception:
            pass


class Vector:
    """zs::Vector<T> (container/Vector.hpp:11-421) through container__v_T & friends."""

    def __init__(self, dtype, n, alloc=None):
        self.T = dtype
        self.alloc = alloc or Allocator()
        self._h = getattr(lib(), "container__v_" + dtype)(self.alloc._h, n)

    def __del__(self):
        try:
            getattr(lib(), "del_container__v_" + self.T)(self._h)
        except Exception:
            pass

    def _f(self, name):
        return getattr(lib(), name % self.T)

    def size(self):
        return self._f("container_size__v_%s")(self._h)

    def capacity(self):
        return self._f("container_capacity__v_%s")(self._h)

    def resize(self, n):
        self._f("resize_container__v_%s")(self._h, n)

    def reset(self, byte):
        self._f("reset_container__v_%s")(self._h, byte)

    def relocate(self, memsrc, devid):
        self._f("relocate_container__v_%s")(self._h, memsrc, devid)

    def getVal(self, i=0):
        """getVal(i): get_val_container__v_T(v) / get_val_i_container__v_T(v, i) (VectorInstantiations.cpp:70-92)."""
        if i == 0:
            return self._f("get_val_container__v_%s")(self._h)
        return self._f("get_val_i_container__v_%s")(self._h, i)

    def setVal(self, v, i=0):
        if i == 0:
            self._f("set_val_container__v_%s")(self._h, v)
        else:
            self._f("set_val_i_container__v_%s")(self._h, i, v)

    def assignVals(self, host_array):
        """copy_to_container__v_T(v, src): size() elements from host memory (Vector::assignVals)."""
        self._f("copy_to_container__v_%s")(self._h, host_array.ctypes.data)

    def retrieveVals(self, host_array):
        self._f("copy_from_container__v_%s")(self._h, host_array.ctypes.data)

    def data(self):
        return self._f("container_data__v_%s")(self._h)

    def begin(self, idx=0):
        return Port(self.data(), idx, 0, 0, 1)


class TileVector:
    """zs::TileVector<T, L> (container/TileVector.hpp:14-561) through container__tv_T_L & friends."""

    def __init__(self, dtype, lane_width, tags, n, alloc=None):
        self.T, self.L = dtype, lane_width
        self.alloc = alloc or Allocator()
        self.s = "%s_%d" % (dtype, lane_width)
        t = self._make_tags(tags)
        self._h = getattr(lib(), "container__tv_" + self.s)(self.alloc._h, t, n)
        lib().del_property_tags(t)

    @staticmethod
    def _make_tags(tags):
        names = (C.c_char_p * len(tags))(*[k.encode() for k, _ in tags])
        sizes = (C.c_int * len(tags))(*[v for _, v in tags])
        return lib().property_tags(names, sizes, len(tags))

    def __del__(self):
        try:
            getattr(lib(), "del_container__tv_" + self.s)(self._h)
        except Exception:
            pass

    def _f(self, name):
        return getattr(lib(), name % self.s)

    def size(self):
        return self._f("container_size__tv_%s")(self._h)

    def capacity(self):
        return self._f("container_capacity__tv_%s")(self._h)

    def numChannels(self):
        return self._f("container_num_channels__tv_%s")(self._h)

    def getPropertyOffset(self, name):
        return self._f("property_offset__tv_%s")(self._h, name.encode())

    def getPropertySize(self, name):
        return self._f("property_size__tv_%s")(self._h, name.encode())

    def resize(self, n):
        self._f("resize_container__tv_%s")(self._h, n)

    def reset(self, byte):
        self._f("reset_container__tv_%s")(self._h, byte)

    def data(self):
        return self._f("container_data__tv_%s")(self._h)

    def iterator(self, prop_or_chn, idx=0):
        """get_iterator_1__tv_T_L(v, id, chnOffset) -- TileVectorInstantiations.cpp"""
        chn = self.getPropertyOffset(prop_or_chn) if isinstance(prop_or_chn, str) else int(prop_or_chn)
        return self._f("get_iterator_1__tv_%s")(self._h, idx, chn)

    def append_channels(self, pol, tags):
        t = self._make_tags(tags)
        self._f("append_properties__rocm_tv_%s")(pol.handle, self._h, t)
        lib().del_property_tags(t)

    def fill(self, pol, val):
        self._f("zs_rocm_fill__tv_%s")(pol.handle, self._h, val)

    def reorder(self, pol, map_ptr, gather=True):
        self._f("zs_rocm_reorder__tv_%s")(pol.handle, self._h, map_ptr, int(gather))

    def nbytes(self):
        tiles = (self.size() + self.L - 1) // self.L
        return tiles * self.L * self.numChannels() * _ES[self.T]


class Bht:
    """zs::bht<int, dim, int, B> (container/Bht.hpp:16-272), dim 1-4, B 16|32, through container__bht_int_D_int_B & friends."""

    def __init__(self, dim, n, alloc=None, bucket=16):
        self.dim = dim
        self.bucket = bucket
        self.alloc = alloc or Allocator()
        self.s = "bht_int_%d_int_%d" % (dim, bucket)
        self._h = getattr(lib(), "container__" + self.s)(self.alloc._h, n)

    def __del__(self):
        try:
            getattr(lib(), "del_container__" + self.s)(self._h)
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def size(self):
        return getattr(lib(), "container_size__" + self.s)(self._h)

    def success(self):
        """the table's `success` word (Bht.hpp:536-541: cleared by an insert that found its three buckets full / by the proximity guard)"""
        v = self.view()
        w = C.c_int(0)
        C.CDLL("libamdhip64.so").hipMemcpy(C.byref(w), C.c_void_p(v.success), C.c_size_t(4), 2)
        return bool(w.value)

    def tableSize(self):
        return getattr(lib(), "container_capacity__" + self.s)(self._h)

    def reset(self, clear_cnt=True):
        getattr(lib(), "reset_container__" + self.s)(self._h, int(clear_cnt))

    def view(self):
        """pyview__bht_...: BhtViewLite (py_interop/BhtView.hpp:96-111); returned as a plain copy."""
        p = getattr(lib(), "pyview__" + self.s)(self._h)
        v = type(p.contents)()
        C.pointer(v)[0] = p.contents
        getattr(lib(), "del_pyview__" + self.s)(p)
        return v

    def resize(self, pol, new_capacity):
        getattr(lib(), "resize_container__rocm_" + self.s)(pol.handle, self._h, new_capacity)

    def insert(self, pol, keys_ptr, n, ret_ptr=None):
        getattr(lib(), "zs_rocm_insert__" + self.s)(pol.handle, self._h, keys_ptr, n, ret_ptr)

    def assign(self, pol, keys_ptr, n):
        """table := {keys[i] -> i}: adopt a partition numbered elsewhere (e.g. a zs::HashTable's active keys)."""
        getattr(lib(), "zs_rocm_assign__" + self.s)(pol.handle, self._h, keys_ptr, n)

    def query(self, pol, keys_ptr, n, ret_ptr):
        getattr(lib(), "zs_rocm_query__" + self.s)(pol.handle, self._h, keys_ptr, n, ret_ptr)

    def reorder(self, pol, map_ptr, scatter=False):
        getattr(lib(), "zs_rocm_reorder__" + self.s)(pol.handle, self._h, map_ptr, int(scatter))

    def canonicalize(self, pol, axes=None, first=0):
        """renumber the entries in lexicographic key order; axes: the components from most to least significant (default 0, 1, ..);
        first: the entries [0, first) keep their numbers, only the rest is sorted"""
        if axes is None and not first:
            getattr(lib(), "zs_rocm_canonicalize__" + self.s)(pol.handle, self._h)
            return
        axes = [int(a) for a in (axes if axes is not None else range(self.dim))]
        if len(axes) != self.dim or getattr(lib(), "zs_rocm_canonicalize_tail__" + self.s)(pol.handle, self._h, (C.c_int * self.dim)(*axes), int(first)) != 0:
            raise ValueError("Bht.canonicalize: axes must be a permutation of 0 .. %d" % (self.dim - 1))

    def order_morton(self, pol):
        """renumber the entries along the Z-order curve of their keys (the launch order of the per-block MPM kernels)"""
        getattr(lib(), "zs_rocm_order_morton__" + self.s)(pol.handle, self._h)


class HashTable:
    """zs::HashTable<i32, dim, int> (container/HashTable.hpp:16-592): hash_combine hash, linear probing with stride 127 --
    the table `partition_for_particles` returns and the Grids-based MPM path keys its blocks with."""

    def __init__(self, dim, n, memsrc=1, devid=0):
        self.dim = dim
        self._h = lib().zs_rocm_hashtable_create(dim, n, memsrc, devid)
        if not self._h:
            raise ValueError("HashTable: dim must be 1..4")

    def __del__(self):
        try:
            lib().zs_rocm_hashtable_destroy(self._h)
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def size(self):
        return lib().zs_rocm_hashtable_size(self._h)

    def tableSize(self):
        return lib().zs_rocm_hashtable_table_size(self._h)

    def view(self):
        from ._lib import HashTableView
        v = HashTableView()
        lib().zs_rocm_hashtable_get_view(self._h, C.byref(v))
        return v

    def reset(self, pol, clear_cnt=True):
        lib().zs_rocm_hashtable_reset(pol.handle, self._h, int(clear_cnt))

    def insert(self, pol, keys_ptr, n, ret_ptr=None):
        lib().zs_rocm_hashtable_insert(pol.handle, self._h, keys_ptr, n, ret_ptr)

    def insert_ids(self, pol, keys_ptr, ids_ptr, n, ok_ptr=None):
        lib().zs_rocm_hashtable_insert_ids(pol.handle, self._h, keys_ptr, ids_ptr, n, ok_ptr)

    def query(self, pol, keys_ptr, n, ret_ptr):
        lib().zs_rocm_hashtable_query(pol.handle, self._h, keys_ptr, n, ret_ptr)

    def entry(self, pol, keys_ptr, n, ret_ptr):
        lib().zs_rocm_hashtable_entry(pol.handle, self._h, keys_ptr, n, ret_ptr)

    def resize(self, pol, n):
        lib().zs_rocm_hashtable_resize(pol.handle, self._h, n)

    def preserve(self, pol, n):
        lib().zs_rocm_hashtable_preserve(pol.handle, self._h, n)


class LBvh:
    """zs::LBvh<3, int, f32> (container/Bvh.hpp:86-1248): build / refit over [n][6] float boxes {min xyz, max xyz} held in a
    device tensor; bulk iter_neighbors as a count pass + exclusive scan + fill pass."""

    def __init__(self):
        self._h = lib().zs_rocm_lbvh_create()

    def __del__(self):
        try:
            lib().zs_rocm_lbvh_destroy(self._h)
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def build(self, pol, bvs, refit=True):
        n = bvs.numel() // 6
        lib().zs_rocm_lbvh_build(pol.handle, self._h, bvs.data_ptr(), n, int(refit))

    def refit(self, pol, bvs):
        if lib().zs_rocm_lbvh_refit(pol.handle, self._h, bvs.data_ptr(), bvs.numel() // 6) != 0:
            raise RuntimeError("bvh topology changes, require rebuild!")  # Bvh.hpp:1230-1231

    def numLeaves(self):
        return lib().zs_rocm_lbvh_num_leaves(self._h)

    def numNodes(self):
        return lib().zs_rocm_lbvh_num_nodes(self._h)

    def view(self):
        from ._lib import LBvhView
        v = LBvhView()
        lib().zs_rocm_lbvh_get_view(self._h, C.byref(v))
        return v

    def query(self, pol, queries):
        """(offsets[nq+1], ids): ids[offsets[q]:offsets[q+1]] = primitives overlapping queries[q], traversal order."""
        import torch
        from .primitives import exclusive_scan
        nq = queries.numel() // 6
        counts = torch.zeros(nq + 1, dtype=torch.int32, device=queries.device)
        lib().zs_rocm_lbvh_query_count(pol.handle, self._h, queries.data_ptr(), nq, counts.data_ptr())
        offsets = torch.empty_like(counts)
        exclusive_scan(pol, counts, offsets)
        pol.syncCtx()
        total = int(offsets[nq].item())
        out = torch.empty(max(total, 1), dtype=torch.int32, device=queries.device)
        lib().zs_rocm_lbvh_query_fill(pol.handle, self._h, queries.data_ptr(), nq, offsets.data_ptr(), out.data_ptr())
        return offsets, out[:total]

    def self_query(self, pol, device="cuda"):
        """Self-collision broadphase: (offsets[numLeaves+1], pairs[total, 2]) -- every unordered pair of overlapping primitive
        boxes once; offsets is indexed by the leaf's position in node order (self_iter_neighbors over every leaf)."""
        import torch
        from .primitives import exclusive_scan
        n = int(self.numLeaves())
        counts = torch.zeros(n + 1, dtype=torch.int32, device=device)
        lib().zs_rocm_lbvh_self_query_count(pol.handle, self._h, counts.data_ptr())
        offsets = torch.empty_like(counts)
        exclusive_scan(pol, counts, offsets)
        pol.syncCtx()
        total = int(offsets[n].item())
        pairs = torch.empty(max(total, 1) * 2, dtype=torch.int32, device=device)
        lib().zs_rocm_lbvh_self_query_fill(pol.handle, self._h, offsets.data_ptr(), pairs.data_ptr())
        return offsets, pairs[: 2 * total].view(total, 2)


class IndexBuckets:
    """zs::IndexBuckets<3, i32, i32> (container/IndexBuckets.hpp) built by index_buckets_for_particles
    (simulation/particle/Query.tpp:9-58)."""

    def __init__(self):
        self._h = lib().zs_rocm_index_buckets_create()

    def __del__(self):
        try:
            lib().zs_rocm_index_buckets_destroy(self._h)
        except Exception:
            pass

    def build(self, pol, pos_port, n, dx, displacement=0.5, expected_cells=0):
        lib().zs_rocm_index_buckets_for_particles(pol.handle, self._h, pos_port, n, dx, displacement, expected_cells)

    def build_for_partition(self, pol, pos_port, n, dx, table_handle, side, key_is_origin=False):
        """Buckets over the cells of a block partition (zs_rocm_index_buckets_for_partition): no hash table, bucket = block * side^3 + cell."""
        lib().zs_rocm_index_buckets_for_partition(pol.handle, self._h, pos_port, n, dx, table_handle, int(side), int(key_is_origin))

    def view(self):
        from ._lib import IndexBucketsView
        v = IndexBucketsView()
        lib().zs_rocm_index_buckets_get_view(self._h, C.byref(v))
        return v
