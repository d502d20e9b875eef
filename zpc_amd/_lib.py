"""ctypes binding of libzsrocm.so (include/zs_rocm.h).  The product path: if the HIP library is missing
this module raises -- there is no CPU fallback anywhere in zpc_amd."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzsrocm.so")
if os.environ.get("ZS_ROCM_LIB"):  # measurement builds of the SAME library (tools/ab_build.sh); announced, never silent
    LIB_PATH = os.environ["ZS_ROCM_LIB"]
    print("[zpc_amd] using HIP library %s" % LIB_PATH, flush=True)


class Port(C.Structure):
    """aosoa_iterator_port (py_interop/GenericIterator.hpp:11-16), passed by value."""
    _fields_ = [("base", C.c_void_p), ("idx", C.c_uint32), ("numTileBits", C.c_uint32),
                ("tileMask", C.c_uint32), ("numChns", C.c_uint32)]


class HashTableView(C.Structure):
    """zs_rocm_hashtable_view (HashTableView members, container/HashTable.hpp:472-476)."""
    _fields_ = [("keys", C.c_void_p), ("indices", C.c_void_p), ("status", C.c_void_p), ("activeKeys", C.c_void_p),
                ("cnt", C.c_void_p), ("tableSize", C.c_int)]


class LBvhView(C.Structure):
    """zs_rocm_lbvh_view (LBvhView members, container/Bvh.hpp:790-792)."""
    _fields_ = [("orderedBvs", C.c_void_p), ("parents", C.c_void_p), ("levels", C.c_void_p), ("leafInds", C.c_void_p),
                ("auxIndices", C.c_void_p), ("numNodes", C.c_int), ("numLeaves", C.c_int)]


class IndexBucketsView(C.Structure):
    """zs_rocm_index_buckets_view (IndexBucketsView members, container/IndexBuckets.hpp:112-114)."""
    _fields_ = [("table", C.c_void_p), ("indices", C.c_void_p), ("offsets", C.c_void_p), ("counts", C.c_void_p),
                ("numBuckets", C.c_int), ("numEntries", C.c_int), ("dx", C.c_float)]


class TvViewLite(C.Structure):
    """TileVectorViewLite<T, L> (py_interop/TileVectorView.hpp:9-131)."""
    _fields_ = [("_vector", C.c_void_p), ("_numChannels", C.c_int)]


class TvNamedViewLite(C.Structure):
    """TileVectorNamedViewLite<T, L> (py_interop/TileVectorView.hpp:133-287)."""
    _fields_ = [("_vector", C.c_void_p), ("_numChannels", C.c_int), ("_tagNames", C.c_void_p), ("_tagOffsets", C.c_void_p),
                ("_tagSizes", C.c_void_p), ("_N", C.c_int)]


class BhtViewLite(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("indices", C.c_void_p), ("status", C.c_void_p), ("activeKeys", C.c_void_p),
                ("cnt", C.c_void_p), ("success", C.c_void_p), ("tableSize", C.c_uint32), ("numBuckets", C.c_uint32),
                ("hf0x", C.c_uint32), ("hf0y", C.c_uint32), ("hf1x", C.c_uint32), ("hf1y", C.c_uint32),
                ("hf2x", C.c_uint32), ("hf2y", C.c_uint32)]


class Particles(C.Structure):
    _fields_ = [("mass", Port), ("pos", Port), ("vel", Port), ("C", Port), ("F", Port), ("logJp", Port),
                ("stress", Port), ("n", C.c_size_t)]


class SlotStorage(C.Structure):
    """zs_rocm_slot_storage: the slotted particle storage's device buffers (all owned by the caller)."""
    _fields_ = [("cellMask", C.c_void_p), ("K", C.c_int), ("nbr", C.c_void_p), ("nbr27", C.c_void_p), ("moverCount", C.c_void_p),
                ("claim", C.c_void_p), ("moverRec", C.c_void_p), ("outboxCap", C.c_int), ("status", C.c_void_p),
                ("blockEdge", C.c_void_p)]


class MpmStep(C.Structure):
    """zs_rocm_mpm_step: arguments of zs_rocm_mpm_step_slotted (one sub-step of the slotted MPM path in one call)."""
    _fields_ = [("params", C.c_void_p), ("particles", Particles), ("table", C.c_void_p), ("gridA", C.c_void_p), ("gridB", C.c_void_p),
                ("nblocks", C.c_size_t), ("storage", C.c_void_p), ("writeAll", C.c_int), ("extf", C.c_float * 3),
                ("maxVelSqr", C.c_void_p), ("collider", C.c_void_p), ("nBoundary", C.c_size_t), ("dist", C.c_void_p),
                ("plan", C.c_void_p), ("commPolicy", C.c_void_p), ("haloGrid", C.c_void_p), ("evTransferBegin", C.c_void_p),
                ("evTransferEnd", C.c_void_p), ("evBreakdown", C.POINTER(C.c_void_p)), ("haloChannels", C.c_int), ("rangeSchedule", C.c_int), ("handoverSnapshot", C.c_void_p)]


class MpmParams(C.Structure):
    _fields_ = [("model", C.c_int), ("dx", C.c_float), ("dt", C.c_float), ("volume", C.c_float), ("E", C.c_float),
                ("nu", C.c_float), ("cohesion", C.c_float), ("beta", C.c_float), ("yieldSurface", C.c_float),
                ("volCorrection", C.c_int), ("side", C.c_int), ("keyIsOrigin", C.c_int), ("yieldStress", C.c_float),
                ("xi", C.c_float), ("Msqr", C.c_float), ("hardeningOn", C.c_int), ("bulk", C.c_float), ("viscosity", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libzsrocm.so not built (%s): run `python -m zpc_amd.build` -- zpc_amd has no CPU fallback" % LIB_PATH)
        try:
            # PyTorch-ROCm bundles its own libamdhip64.so; it must be the first HIP runtime mapped into the process,
            # otherwise libzsrocm binds /opt/rocm's copy and torch tensors / streams live in a different runtime
            # instance (observed: kernels then read stale data).  torch is only memory + stream plumbing here.
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


class Collider(C.Structure):
    """zs_rocm_collider (include/zs_rocm.h): Collider<AnalyticLevelSet<...>, f32, 3> of geometry/Collider.h"""
    _fields_ = [("geometry", C.c_int), ("type", C.c_int), ("param", C.c_float * 8), ("s", C.c_float), ("dsdt", C.c_float),
                ("R", C.c_float * 9), ("omega", C.c_float * 3), ("b", C.c_float * 3), ("dbdt", C.c_float * 3)]


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
    L.zs_rocm_memset.argtypes = [vp, vp, i32, sz]
    L.zs_rocm_dist_unique_id_bytes.restype = sz
    L.zs_rocm_dist_unique_id.argtypes = [vp]
    L.zs_rocm_dist_create.argtypes = [i32, i32, vp, i32]
    L.zs_rocm_dist_create.restype = vp
    L.zs_rocm_dist_destroy.argtypes = [vp]
    L.zs_rocm_dist_rank.argtypes = [vp]
    L.zs_rocm_dist_world.argtypes = [vp]
    L.zs_rocm_dist_comm_count.argtypes = [vp]
    L.zs_rocm_dist_comm_count.restype = C.c_int
    L.zs_rocm_dist_halo_exchange.argtypes = [vp, vp, vp, i32, i32, i32, vp, sz, i32, vp, vp, vp, vp, vp]
    L.zs_rocm_dist_allreduce_f32.argtypes = [vp, vp, vp, sz, i32]
    L.zs_rocm_dist_allreduce_i64.argtypes = [vp, vp, vp, sz, i32]
    L.zs_rocm_dist_alltoall_i64.argtypes = [vp, vp, vp, vp]
    L.zs_rocm_dist_alltoallv_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.zs_rocm_dist_barrier.argtypes = [vp, vp]
    L.zs_rocm_halo_plan_from_keys.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.zs_rocm_halo_plan_from_keys.restype = sz
    L.zs_rocm_dist_halo_plan_create.argtypes = [vp, vp, vp, sz, i32]
    L.zs_rocm_dist_halo_plan_create.restype = vp
    L.zs_rocm_dist_halo_plan_destroy.argtypes = [vp]
    L.zs_rocm_dist_halo_plan_npeers.argtypes = [vp]
    L.zs_rocm_dist_halo_plan_npeers.restype = i32
    L.zs_rocm_dist_halo_plan_blocks.argtypes = [vp]
    L.zs_rocm_dist_halo_plan_blocks.restype = sz
    L.zs_rocm_dist_halo_plan_bytes.argtypes = [vp]
    L.zs_rocm_dist_halo_plan_bytes.restype = sz
    L.zs_rocm_dist_halo_plan_block_list.argtypes = [vp]
    L.zs_rocm_dist_halo_plan_block_list.restype = vp
    L.zs_rocm_dist_halo_plan_exchange.argtypes = [vp, vp, vp, vp, i32, i32]
    L.zs_rocm_dist_halo_plan_exchange.restype = i32
    L.zs_rocm_mpm_slot_outbox_bytes.argtypes = [sz, i32, i32]
    L.zs_rocm_mpm_slot_outbox_bytes.restype = sz
    L.zs_rocm_mpm_build_neighbors27.argtypes = [vp, vp, vp, i32]
    L.zs_rocm_mpm_slot_particles.argtypes = [vp, vp, Port, sz, f32, i32, i32, i32, vp, vp, i32, vp, vp]
    L.zs_rocm_mpm_slot_particles.restype = i32
    L.zs_rocm_mpm_slot_list.argtypes = [vp, vp, sz, i32, vp]
    L.zs_rocm_mpm_slot_list.restype = sz
    L.zs_rocm_policy_temporary.argtypes = [vp, sz]
    L.zs_rocm_policy_temporary.restype = vp
    L.zs_rocm_policy_temporary_free.argtypes = [vp, vp]
    L.zs_rocm_last_error.argtypes = [i32]
    L.zs_rocm_clear_error.argtypes = [i32]
    L.launch__device.argtypes = [vp, vp, sz, vp]
    L.rocm_compile_program.argtypes = [C.c_char_p, i32, C.c_char_p, C.c_bool, C.c_bool, C.c_bool, C.c_bool, C.c_char_p]
    L.rocm_compile_program.restype = sz
    L.rocm_load_module.argtypes = [vp, C.c_char_p]
    L.rocm_load_module.restype = vp
    L.rocm_unload_module.argtypes = [vp, vp]
    L.rocm_get_kernel.argtypes = [vp, vp, C.c_char_p]
    L.rocm_get_kernel.restype = vp
    L.rocm_launch_kernel.argtypes = [vp, vp, sz, vp, vp]
    L.rocm_launch_kernel.restype = sz
    for T in ("int", "float", "double"):
        for op in ("reduce_sum", "reduce_prod", "reduce_min", "reduce_max", "exclusive_scan_sum",
                   "exclusive_scan_prod", "inclusive_scan_sum", "inclusive_scan_prod"):
            getattr(L, "%s__rocm_%s_1" % (op, T)).argtypes = [vp, Port, Port, Port]
        getattr(L, "radix_sort__rocm_%s_1" % T).argtypes = [vp, Port, Port, Port]
        getattr(L, "merge_sort__rocm_%s_1" % T).argtypes = [vp, Port, Port]
        getattr(L, "merge_sort_pair__rocm_%s_1" % T).argtypes = [vp, Port, Port, sz]
        getattr(L, "radix_sort_pair__rocm_%s_1" % T).argtypes = [vp, Port, Port, Port, Port, sz]
    for S, ct in (("i32", C.c_int32), ("i64", C.c_int64), ("f32", C.c_float), ("f64", C.c_double)):
        getattr(L, "zs_rocm_reduce_" + S).argtypes = [vp, vp, sz, vp, ct, i32]
        getattr(L, "zs_rocm_scan_" + S).argtypes = [vp, vp, sz, vp, ct, i32, i32]
    for S in ("i32", "u32", "i64", "u64"):
        getattr(L, "zs_rocm_radix_sort_" + S).argtypes = [vp, vp, vp, vp, vp, sz, i32, i32]
    for S in ("i32", "u32", "i64", "u64", "f32", "f64"):
        getattr(L, "zs_rocm_merge_sort_" + S).argtypes = [vp, vp, vp, sz, i32]


def _declare_containers(L):
    vp, sz, i32, f32, i8 = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_int8
    L.allocator.argtypes = [i32, i8]
    L.allocator.restype = vp
    L.del_allocator.argtypes = [vp]
    L.property_tags.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), sz]
    L.property_tags.restype = vp
    L.del_property_tags.argtypes = [vp]
    L.allocator_virtual.argtypes = [i32, i8, sz]
    L.allocator_virtual.restype = vp
    L.del_allocator_virtual.argtypes = [vp]
    L.property_tags_get_item.argtypes = [vp, sz, C.POINTER(C.c_char_p), C.POINTER(sz)]
    L.property_tags_get_size.argtypes = [vp]
    L.property_tags_get_size.restype = sz
    for T, ct in (("int", C.c_int), ("float", C.c_float), ("double", C.c_double)):
        for sfx in ("", "_virtual"):  # ZSPmrAllocator<false> / <true> spellings of the same entry points
            g = lambda n: getattr(L, (n % T) + sfx)
            g("container__v_%s").argtypes = [vp, sz]
            g("container__v_%s").restype = vp
            g("del_container__v_%s").argtypes = [vp]
            g("relocate_container__v_%s").argtypes = [vp, i32, i8]
            g("resize_container__v_%s").argtypes = [vp, sz]
            g("reset_container__v_%s").argtypes = [vp, i32]
            g("container_size__v_%s").argtypes = [vp]
            g("container_size__v_%s").restype = sz
            g("container_capacity__v_%s").argtypes = [vp]
            g("container_capacity__v_%s").restype = sz
            g("get_val_container__v_%s").argtypes = [vp]
            g("get_val_container__v_%s").restype = ct
            g("set_val_container__v_%s").argtypes = [vp, ct]
            g("get_val_i_container__v_%s").argtypes = [vp, sz]
            g("get_val_i_container__v_%s").restype = ct
            g("set_val_i_container__v_%s").argtypes = [vp, sz, ct]
            g("copy_to_container__v_%s").argtypes = [vp, vp]
            g("copy_from_container__v_%s").argtypes = [vp, vp]
            g("get_handle_container__v_%s").argtypes = [vp]
            g("get_handle_container__v_%s").restype = vp
            g("pyview__v_%s").argtypes = [vp]
            g("pyview__v_%s").restype = C.POINTER(C.c_void_p)
            g("pyview__v_const_%s").argtypes = [vp]
            g("pyview__v_const_%s").restype = C.POINTER(C.c_void_p)
            for it in ("get_iterator_1__v_%s", "get_iterator_1__v_const_%s", "get_iterator_3__v_%s", "get_iterator_3__v_const_%s"):
                g(it).argtypes = [vp, C.c_uint32]
                g(it).restype = Port
        getattr(L, "del_pyview__v_%s" % T).argtypes = [vp]
        getattr(L, "del_pyview__v_const_%s" % T).argtypes = [vp]
        getattr(L, "container_data__v_%s" % T).argtypes = [vp]
        getattr(L, "container_data__v_%s" % T).restype = vp
        for Lw in (8, 32, 64, 512):
            s = "%s_%d" % (T, Lw)
            for sfx in ("", "_virtual"):
                h = lambda n: getattr(L, (n % s) + sfx)
                h("container__tv_%s").argtypes = [vp, vp, sz]
                h("container__tv_%s").restype = vp
                h("del_container__tv_%s").argtypes = [vp]
                h("relocate_container__tv_%s").argtypes = [vp, i32, i8]
                h("resize_container__tv_%s").argtypes = [vp, sz]
                h("reset_container__tv_%s").argtypes = [vp, i32]
                for q in ("container_size__tv_%s", "container_capacity__tv_%s"):
                    h(q).argtypes = [vp]
                    h(q).restype = sz
                h("property_offset__tv_%s").argtypes = [vp, C.c_char_p]
                h("property_size__tv_%s").argtypes = [vp, C.c_char_p]
                for it in ("get_iterator_1__tv_%s", "get_iterator_3__tv_%s"):
                    h(it).argtypes = [vp, C.c_uint32, C.c_uint32]
                    h(it).restype = Port
                for it in ("get_iterator_1__tv_const_%s", "get_iterator_3__tv_const_%s"):
                    h(it).argtypes = [vp, C.c_uint32, C.c_uint32]
                    h(it).restype = Port
                for pv, rt in (("pyview__tv_%s", TvViewLite), ("pyview__tv_const_%s", TvViewLite), ("pyview__tvn_%s", TvNamedViewLite),
                               ("pyview__tvn_const_%s", TvNamedViewLite)):
                    h(pv).argtypes = [vp]
                    h(pv).restype = C.POINTER(rt)
                h("append_properties__rocm_tv_%s").argtypes = [vp, vp, vp]
            k = lambda n: getattr(L, n % s)
            for dv in ("del_pyview__tv_%s", "del_pyview__tv_const_%s", "del_pyview__tvn_%s", "del_pyview__tvn_const_%s"):
                k(dv).argtypes = [vp]
            k("container_num_channels__tv_%s").argtypes = [vp]
            k("container_num_channels__tv_%s").restype = sz
            k("container_data__tv_%s").argtypes = [vp]
            k("container_data__tv_%s").restype = vp
            k("zs_rocm_fill__tv_%s").argtypes = [vp, vp, ct]
            k("zs_rocm_reorder__tv_%s").argtypes = [vp, vp, vp, i32]
    L.zs_rocm_tv_from_aos_f32.argtypes = [vp, vp, sz, i32, i32, vp]
    L.zs_rocm_tv_to_aos_f32.argtypes = [vp, vp, sz, i32, i32, vp]
    L.zs_rocm_tv_scale_f32.argtypes = [vp, vp, sz, i32, i32, f32]
    L.zs_rocm_tv_gather_f32.argtypes = [vp, vp, vp, sz, i32, i32, vp]
    L.zs_rocm_tv_gather_channels_f32.argtypes = [vp, vp, vp, sz, i32, i32, vp, C.c_uint64]
    L.zs_rocm_tv_gather_rows_f32.argtypes = [vp, vp, vp, sz, i32, i32, vp]
    L.zs_rocm_tv_scatter_rows_f32.argtypes = [vp, vp, sz, i32, i32, vp, sz]
    for D, B in ((d, b) for d in (1, 2, 3, 4) for b in (16, 32)):
        s = "bht_int_%d_int_%d" % (D, B)
        for sfx in ("", "_virtual"):
            getattr(L, "relocate_container__" + s + sfx).argtypes = [vp, i32, i8]
            getattr(L, "pyview__bht_const_int_%d_int_%d" % (D, B) + sfx).argtypes = [vp]
            getattr(L, "pyview__bht_const_int_%d_int_%d" % (D, B) + sfx).restype = C.POINTER(BhtViewLite)
            if sfx:
                getattr(L, "container__" + s + sfx).argtypes = [vp, sz]
                getattr(L, "container__" + s + sfx).restype = vp
                getattr(L, "del_container__" + s + sfx).argtypes = [vp]
                getattr(L, "container_size__" + s + sfx).argtypes = [vp]
                getattr(L, "container_size__" + s + sfx).restype = sz
        getattr(L, "container__" + s).argtypes = [vp, sz]
        getattr(L, "container__" + s).restype = vp
        getattr(L, "del_container__" + s).argtypes = [vp]
        getattr(L, "container_size__" + s).argtypes = [vp]
        getattr(L, "container_size__" + s).restype = sz
        getattr(L, "container_capacity__" + s).argtypes = [vp]
        getattr(L, "container_capacity__" + s).restype = sz
        getattr(L, "reset_container__" + s).argtypes = [vp, i32]
        getattr(L, "pyview__" + s).argtypes = [vp]
        getattr(L, "pyview__" + s).restype = C.POINTER(BhtViewLite)
        getattr(L, "del_pyview__" + s).argtypes = [C.POINTER(BhtViewLite)]
        getattr(L, "resize_container__rocm_" + s).argtypes = [vp, vp, sz]
        getattr(L, "zs_rocm_insert__" + s).argtypes = [vp, vp, vp, sz, vp]
        getattr(L, "zs_rocm_query__" + s).argtypes = [vp, vp, vp, sz, vp]
        getattr(L, "zs_rocm_assign__" + s).argtypes = [vp, vp, vp, sz]
        getattr(L, "zs_rocm_reorder__" + s).argtypes = [vp, vp, vp, i32]
        getattr(L, "zs_rocm_canonicalize__" + s).argtypes = [vp, vp]
        getattr(L, "zs_rocm_canonicalize_axes__" + s).argtypes = [vp, vp, C.POINTER(C.c_int)]
        getattr(L, "zs_rocm_canonicalize_axes__" + s).restype = i32
        getattr(L, "zs_rocm_canonicalize_tail__" + s).argtypes = [vp, vp, C.POINTER(C.c_int), sz]
        getattr(L, "zs_rocm_canonicalize_tail__" + s).restype = i32
        getattr(L, "zs_rocm_order_morton__" + s).argtypes = [vp, vp]
    L.zs_rocm_hashtable_create.argtypes = [i32, sz, i32, i32]
    L.zs_rocm_hashtable_create.restype = vp
    L.zs_rocm_hashtable_destroy.argtypes = [vp]
    L.zs_rocm_hashtable_dim.argtypes = [vp]
    L.zs_rocm_hashtable_table_size.argtypes = [vp]
    L.zs_rocm_hashtable_table_size.restype = sz
    L.zs_rocm_hashtable_size.argtypes = [vp]
    L.zs_rocm_hashtable_get_view.argtypes = [vp, C.POINTER(HashTableView)]
    L.zs_rocm_hashtable_reset.argtypes = [vp, vp, i32]
    L.zs_rocm_hashtable_insert.argtypes = [vp, vp, vp, sz, vp]
    L.zs_rocm_hashtable_insert_ids.argtypes = [vp, vp, vp, vp, sz, vp]
    L.zs_rocm_hashtable_query.argtypes = [vp, vp, vp, sz, vp]
    L.zs_rocm_hashtable_entry.argtypes = [vp, vp, vp, sz, vp]
    L.zs_rocm_hashtable_resize.argtypes = [vp, vp, sz]
    L.zs_rocm_hashtable_preserve.argtypes = [vp, vp, sz]
    L.zs_rocm_mpm_partition_for_particles.argtypes = [vp, vp, Port, sz, f32, i32]
    L.zs_rocm_mpm_enlarge_sparsity__hashtable.argtypes = [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.zs_rocm_lbvh_create.restype = vp
    L.zs_rocm_lbvh_destroy.argtypes = [vp]
    L.zs_rocm_lbvh_num_leaves.argtypes = [vp]
    L.zs_rocm_lbvh_num_leaves.restype = sz
    L.zs_rocm_lbvh_num_nodes.argtypes = [vp]
    L.zs_rocm_lbvh_num_nodes.restype = sz
    L.zs_rocm_lbvh_get_view.argtypes = [vp, C.POINTER(LBvhView)]
    L.zs_rocm_lbvh_build.argtypes = [vp, vp, vp, sz, i32]
    L.zs_rocm_lbvh_refit.argtypes = [vp, vp, vp, sz]
    L.zs_rocm_lbvh_total_box.argtypes = [vp, vp, vp]
    L.zs_rocm_lbvh_query_count.argtypes = [vp, vp, vp, sz, vp]
    L.zs_rocm_lbvh_query_fill.argtypes = [vp, vp, vp, sz, vp, vp]
    L.zs_rocm_lbvh_self_query_count.argtypes = [vp, vp, vp]
    L.zs_rocm_lbvh_self_query_fill.argtypes = [vp, vp, vp, vp]
    L.zs_rocm_index_buckets_create.restype = vp
    L.zs_rocm_index_buckets_destroy.argtypes = [vp]
    L.zs_rocm_index_buckets_get_view.argtypes = [vp, C.POINTER(IndexBucketsView)]
    L.zs_rocm_index_buckets_for_particles.argtypes = [vp, vp, Port, sz, f32, f32, sz]
    L.zs_rocm_index_buckets_for_partition.argtypes = [vp, vp, Port, sz, f32, vp, i32, i32]
    PP = C.POINTER(MpmParams)
    L.zs_rocm_mpm_compute_sparsity.argtypes = [vp, vp, Port, sz, f32, i32, i32]
    L.zs_rocm_mpm_enlarge_sparsity.argtypes = [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), i32]
    L.zs_rocm_mpm_bin_particles.argtypes = [vp, vp, Port, sz, f32, i32, i32, vp, vp, vp]
    L.zs_rocm_mpm_build_neighbors.argtypes = [vp, vp, vp, i32]
    L.zs_rocm_mpm_p2g.argtypes = [vp, PP, Particles, vp, vp, sz, vp, vp, vp]
    L.zs_rocm_mpm_owner_rank.argtypes = [vp, Port, sz, f32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), i32, vp]
    L.zs_rocm_mpm_owner_counts.argtypes = [vp, vp, sz, i32, vp]
    L.zs_rocm_mpm_owner_counts.restype = None
    L.zs_rocm_mpm_g2p2g.argtypes = [vp, PP, Particles, vp, vp, vp, sz, vp, vp, vp, i32]
    L.zs_rocm_mpm_g2p2g_range.argtypes = [vp, PP, Particles, vp, vp, vp, sz, vp, vp, vp, i32, sz, sz, vp]
    L.zs_rocm_mpm_g2p2g_slotted.argtypes = [vp, PP, Particles, vp, vp, vp, sz, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp]
    L.zs_rocm_mpm_g2p2g_slotted.restype = i32
    L.zs_rocm_mpm_g2p2g_slotted_range.argtypes = [vp, PP, Particles, vp, vp, vp, sz, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, sz, sz, i32]
    L.zs_rocm_mpm_g2p2g_slotted_range.restype = i32
    L.zs_rocm_mpm_g2p2g_slots.argtypes = [vp, PP, Particles, vp, vp, vp, sz, C.POINTER(SlotStorage), i32, sz, sz, i32]
    L.zs_rocm_mpm_g2p2g_slots.restype = i32
    L.zs_rocm_mpm_partition_edge.argtypes = [vp, vp, vp, i32, i32, i32]
    L.zs_rocm_mpm_slot_compute_sparsity.argtypes = [vp, vp, vp, sz, i32, i32, vp]
    L.zs_rocm_mpm_reslot.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.zs_rocm_mpm_reslot.restype = i32
    L.zs_rocm_mpm_step_slotted.argtypes = [vp, C.POINTER(MpmStep)]
    L.zs_rocm_mpm_step_slotted.restype = i32
    L.zs_rocm_dist_halo_plan_from_lists.argtypes = [vp, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                                    C.POINTER(C.c_int), sz]
    L.zs_rocm_dist_halo_plan_from_lists.restype = vp
    L.zs_rocm_mpm_g2p2g_reorder_range.argtypes = [vp, PP, Particles, Particles, vp, vp, vp, vp, sz, vp, vp, vp, i32, sz, sz, vp]
    L.zs_rocm_mpm_grid_update.argtypes = [vp, PP, vp, sz, C.POINTER(C.c_float), vp]
    L.zs_rocm_mpm_g2p.argtypes = [vp, PP, Particles, vp, vp, sz, vp, vp, vp]
    L.zs_rocm_mpm_p2c2g.argtypes = [vp, PP, Particles, vp, vp, vp, sz, i32]
    L.zs_rocm_mpm_p2c2g.restype = i32
    L.zs_rocm_mpm_pre_g2c2p.argtypes = [vp, Particles]
    L.zs_rocm_mpm_g2c2p.argtypes = [vp, PP, Particles, vp, vp, vp, sz]
    L.zs_rocm_mpm_g2c2p.restype = i32
    L.zs_rocm_mpm_post_g2c2p.argtypes = [vp, PP, Particles]
    L.zs_rocm_mpm_g2c2p_step.argtypes = [vp, PP, Particles, vp, vp, sz]
    L.zs_rocm_mpm_g2c2p_step.restype = i32
    L.zs_rocm_mpm_stress.argtypes = [vp, PP, vp, vp, sz, vp]
    L.zs_rocm_mpm_update_stress.argtypes = [vp, PP, Particles]
    L.zs_rocm_svd3.argtypes = [vp, vp, sz, vp, vp, vp]
    L.zs_rocm_nacc_msqr.argtypes = [f32]
    L.zs_rocm_nacc_msqr.restype = f32
    L.zs_rocm_collider_init.argtypes = [C.POINTER(Collider), i32, i32, C.POINTER(C.c_float), i32]
    L.zs_rocm_mpm_apply_boundary.argtypes = [vp, PP, vp, vp, sz, C.POINTER(Collider)]
    L.zs_rocm_collider_resolve.argtypes = [vp, C.POINTER(Collider), vp, vp, sz, vp]
    L.zs_rocm_mpm_halo_pack.argtypes = [vp, vp, vp, sz, i32, i32, i32, vp]
    L.zs_rocm_mpm_halo_unpack.argtypes = [vp, vp, vp, sz, i32, i32, i32, vp, i32]


_declare_base = _declare


def _declare(L):  # noqa: F811
    _declare_base(L)
    _declare_containers(L)
