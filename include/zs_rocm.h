/*
 * zs_rocm.h -- C ABI of libzsrocm.so, the MI355X (gfx950) backend for zpc's data-parallel hot path.
 *
 * Two groups of entry points:
 *
 *  (A) the reference's own C ABI (include/zensim/py_interop/, the `zpc_py_interop` target that Python
 *      `zpc_jit` binds through ctypes), re-exported 1:1 with `cuda` -> `rocm` in the symbol names:
 *      policy handles, launch__device, typed parallel primitives over `aosoa_iterator_port`s,
 *      Vector / TileVector / bht container handles.  Each declaration cites the file:line it replaces.
 *
 *  (B) `zs_rocm_*` entry points for work that in the reference only exists as C++ template
 *      instantiations inside user translation units (`pol(range(n), P2GTransfer{...})`,
 *      `tb.insert(key)` in a lambda, ...).  A C++ caller reaches them through the header-only face in
 *      include/zensim_rocm/; each declaration cites the functor it replaces.
 *
 * Conventions (same as the reference): every create returns a heap object owned by the
 * caller and released with the matching del_*; views are non-owning; functions return void and report
 * device errors by printing the first error per device to stderr and latching it
 * (zs_rocm_last_error), never by throwing across the ABI.  All device pointers are plain HIP device
 * pointers; no framework types appear in any signature.
 */
#ifndef ZS_ROCM_H
#define ZS_ROCM_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#  define ZS_ROCM_EXPORT __attribute__((visibility("default")))
#else
#  define ZS_ROCM_EXPORT
#endif

/* ======================================================================== (A) policy & launch */
/* zs::CudaExecutionPolicy (cuda/execution/ExecutionPolicy.cuh:345-911) -> RocmExecutionPolicy */
typedef struct zs_rocm_policy zs_rocm_policy;

/* py_interop/cuda/ExecutionPolicy.cpp:8-9 */
ZS_ROCM_EXPORT zs_rocm_policy *policy__device(void);
ZS_ROCM_EXPORT void del_policy__device(zs_rocm_policy *);
/* fluent setters of ExecutionPolicyInterface / CudaExecutionPolicy
 * (execution/ExecutionPolicy.hpp:110-126, cuda/execution/ExecutionPolicy.cuh:362-385).
 * Defaults: sync = 1, profile = 0, device = -1 (current), stream = -1 (null stream), block = 0. */
ZS_ROCM_EXPORT void zs_rocm_policy_sync(zs_rocm_policy *, int sync);
ZS_ROCM_EXPORT void zs_rocm_policy_profile(zs_rocm_policy *, int profile);
ZS_ROCM_EXPORT void zs_rocm_policy_device(zs_rocm_policy *, int procid);
ZS_ROCM_EXPORT void zs_rocm_policy_stream(zs_rocm_policy *, int streamid); /* -1 | 0..31 spare streams (cuda/Cuda.h:39,115-120) */
ZS_ROCM_EXPORT void zs_rocm_policy_listen(zs_rocm_policy *, int incomingProc, int incomingStreamid);
ZS_ROCM_EXPORT void zs_rocm_policy_shmem(zs_rocm_policy *, size_t bytes);
ZS_ROCM_EXPORT void zs_rocm_policy_block(zs_rocm_policy *, int tpb);
/* run on a stream owned by the caller (e.g. the framework's current stream); NULL restores streamid */
ZS_ROCM_EXPORT void zs_rocm_policy_external_stream(zs_rocm_policy *, void *hipStream);
ZS_ROCM_EXPORT void *zs_rocm_policy_get_stream(const zs_rocm_policy *); /* getStream(), :379 */
ZS_ROCM_EXPORT int zs_rocm_policy_should_sync(const zs_rocm_policy *);
ZS_ROCM_EXPORT void zs_rocm_policy_sync_ctx(const zs_rocm_policy *);   /* syncCtx(), :396-399 */
ZS_ROCM_EXPORT float zs_rocm_policy_last_elapsed_ms(const zs_rocm_policy *); /* profile(true): hipEvent pair */
/* latched error status of the device context (cuda/Cuda.h:291-312); 0 = hipSuccess */
ZS_ROCM_EXPORT int zs_rocm_last_error(int device);
ZS_ROCM_EXPORT void zs_rocm_clear_error(int device);
ZS_ROCM_EXPORT int zs_rocm_device_count(void);
ZS_ROCM_EXPORT int zs_rocm_current_device(void); /* the calling thread's HIP device (0 without a GPU) */
/* scratch memory for a caller-side kernel sequence on the policy's stream: the stream-ordered temporary of
 * get_temporary_memory_source(pol) (resource/cuda/ExecutionPolicy.cu:5-16, temporary_memory_resource::do_allocate /
 * do_deallocate, cuda/memory/Allocator.h:33-50 = cuMemAllocAsync / cuMemFreeAsync).  hipMallocAsync / hipFreeAsync on the
 * policy's stream: a block stays valid until it is handed back and never overlaps another live block or the scratch the
 * library's own primitives take. */
ZS_ROCM_EXPORT void *zs_rocm_policy_temporary(zs_rocm_policy *, size_t bytes);
ZS_ROCM_EXPORT void zs_rocm_policy_temporary_free(zs_rocm_policy *, void *ptr);
/* Vector::reset(byteVal) / TileVector::reset(pol, 0) on raw device memory: hipMemsetAsync on the policy's stream */
ZS_ROCM_EXPORT void zs_rocm_memset(zs_rocm_policy *, void *ptr, int byteVal, size_t bytes);
/* frees the grow-only per-stream arenas of the library's own per-call scratch (stand-in for streamMemFree, cuda/Cuda.cu:169-176) */
ZS_ROCM_EXPORT void zs_rocm_release_temporaries(void);

/* py_interop/cuda/ExecutionPolicy.cpp:11-39: launch a module function (hipFunction_t) over `dim`
 * threads, block 128, grid ceil(dim/128), on the policy's stream, sync if shouldSync() */
ZS_ROCM_EXPORT void launch__device(zs_rocm_policy *, void *kernel, size_t dim, void **args);

/* ======================================================================== (A) runtime compilation */
/* py_interop/cuda/Nvrtc.cpp:29-271 with hiprtc: cuda_compile_program / cuda_load_module / cuda_unload_module /
 * cuda_get_kernel / cuda_launch_kernel -> rocm_*.  `arch` is the gfx number (950); 0 takes it from the current device.
 * The output file is a code object; rocm_get_kernel returns the hipFunction_t that launch__device accepts.
 * rocm_compile_program returns 0 on success (hiprtcResult otherwise, (size_t)-1 if hiprtc cannot be loaded). */
ZS_ROCM_EXPORT size_t rocm_compile_program(const char *src, int arch, const char *include_dir, bool debug, bool verbose,
                                           bool verify_fp, bool fast_math, const char *output_path);
ZS_ROCM_EXPORT void *rocm_load_module(void *pol, const char *path);
ZS_ROCM_EXPORT void rocm_unload_module(void *pol, void *module);
ZS_ROCM_EXPORT void *rocm_get_kernel(void *pol, void *module, const char *name);
ZS_ROCM_EXPORT size_t rocm_launch_kernel(void *context, void *kernel, size_t dim, void **args, void *stream);

/* ======================================================================== (A) iterator ABI */
/* py_interop/GenericIterator.hpp:11-16: element address =
 *   base + ((((idx >> numTileBits) * numChns) << numTileBits) | (idx & tileMask));
 * AoS form: numTileBits = tileMask = 0.  Passed BY VALUE. */
#define ZS_ROCM_DECL_PORT(T, NAME)                                   \
  typedef struct {                                                   \
    T *base;                                                         \
    uint32_t idx, numTileBits, tileMask, numChns;                    \
  } NAME;
ZS_ROCM_DECL_PORT(int, aosoa_iterator_int_1)
ZS_ROCM_DECL_PORT(const int, aosoa_iterator_const_int_1)
ZS_ROCM_DECL_PORT(float, aosoa_iterator_float_1)
ZS_ROCM_DECL_PORT(const float, aosoa_iterator_const_float_1)
ZS_ROCM_DECL_PORT(double, aosoa_iterator_double_1)
ZS_ROCM_DECL_PORT(const double, aosoa_iterator_const_double_1)
/* aosoa_iterator_port<T, 3>: same POD, dereferences to 3 components `tileMask + 1` elements apart (GenericIterator.hpp:96-104);
 * the AoS form has numChns = 3 (element i at base + 3 i) */
ZS_ROCM_DECL_PORT(int, aosoa_iterator_int_3)
ZS_ROCM_DECL_PORT(const int, aosoa_iterator_const_int_3)
ZS_ROCM_DECL_PORT(float, aosoa_iterator_float_3)
ZS_ROCM_DECL_PORT(const float, aosoa_iterator_const_float_3)
ZS_ROCM_DECL_PORT(double, aosoa_iterator_double_3)
ZS_ROCM_DECL_PORT(const double, aosoa_iterator_const_double_3)

/* ======================================================================== (A) parallel primitives */
/* py_interop/cuda/ExecutionPolicy.cpp:41-131 (ZS_DEFINE_PARALLEL_PRIMITIVES for int, float, double).
 * reduce: out[0] = fold(op, init, [first,last)) with init = 0 / 1 / numeric max / numeric lowest;
 * scans with init = identity; merge sort = stable in-place sort by `<` (`:99-111`; pair form permutes int values);
 * radix sort over all key bits (integral T only, as in the reference). */
#define ZS_ROCM_DECL_PRIMITIVES(T)                                                                          \
  ZS_ROCM_EXPORT void reduce_sum__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1 first,        \
                                               aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1);  \
  ZS_ROCM_EXPORT void reduce_prod__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1 first,       \
                                                aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1); \
  ZS_ROCM_EXPORT void reduce_min__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1 first,        \
                                               aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1);  \
  ZS_ROCM_EXPORT void reduce_max__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1 first,        \
                                               aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1);  \
  ZS_ROCM_EXPORT void exclusive_scan_sum__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1,      \
                                                       aosoa_iterator_const_##T##_1, aosoa_iterator_##T##_1); \
  ZS_ROCM_EXPORT void exclusive_scan_prod__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1,     \
                                                        aosoa_iterator_const_##T##_1, aosoa_iterator_##T##_1); \
  ZS_ROCM_EXPORT void inclusive_scan_sum__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1,      \
                                                       aosoa_iterator_const_##T##_1, aosoa_iterator_##T##_1); \
  ZS_ROCM_EXPORT void inclusive_scan_prod__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_const_##T##_1,     \
                                                        aosoa_iterator_const_##T##_1, aosoa_iterator_##T##_1); \
  ZS_ROCM_EXPORT void merge_sort__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_##T##_1 first,              \
                                               aosoa_iterator_##T##_1 last);                                \
  ZS_ROCM_EXPORT void merge_sort_pair__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_##T##_1 keys,          \
                                                    aosoa_iterator_int_1 vals, size_t count);               \
  ZS_ROCM_EXPORT void radix_sort__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_##T##_1 first,              \
                                               aosoa_iterator_##T##_1 last, aosoa_iterator_##T##_1 out);    \
  ZS_ROCM_EXPORT void radix_sort_pair__rocm_##T##_1(zs_rocm_policy *, aosoa_iterator_##T##_1 keysIn,        \
                                                    aosoa_iterator_int_1 valsIn, aosoa_iterator_##T##_1 keysOut, \
                                                    aosoa_iterator_int_1 valsOut, size_t count);
ZS_ROCM_DECL_PRIMITIVES(int)
ZS_ROCM_DECL_PRIMITIVES(float)
ZS_ROCM_DECL_PRIMITIVES(double)

/* (B) primitives on contiguous device arrays with the full argument set of the C++ face
 * (execution/ExecutionPolicy.hpp:698-781: init, bit window [sbit, ebit), other key widths). */
ZS_ROCM_EXPORT void zs_rocm_reduce_i32(zs_rocm_policy *, const int32_t *in, size_t n, int32_t *out, int32_t init, int op);
ZS_ROCM_EXPORT void zs_rocm_reduce_i64(zs_rocm_policy *, const int64_t *in, size_t n, int64_t *out, int64_t init, int op);
ZS_ROCM_EXPORT void zs_rocm_reduce_f32(zs_rocm_policy *, const float *in, size_t n, float *out, float init, int op);
ZS_ROCM_EXPORT void zs_rocm_reduce_f64(zs_rocm_policy *, const double *in, size_t n, double *out, double init, int op);
/* op: 0 = plus, 1 = multiplies, 2 = getmin, 3 = getmax (ZpcFunctional.hpp) */
ZS_ROCM_EXPORT void zs_rocm_scan_i32(zs_rocm_policy *, const int32_t *in, size_t n, int32_t *out, int32_t init, int op, int exclusive);
ZS_ROCM_EXPORT void zs_rocm_scan_i64(zs_rocm_policy *, const int64_t *in, size_t n, int64_t *out, int64_t init, int op, int exclusive);
ZS_ROCM_EXPORT void zs_rocm_scan_f32(zs_rocm_policy *, const float *in, size_t n, float *out, float init, int op, int exclusive);
ZS_ROCM_EXPORT void zs_rocm_scan_f64(zs_rocm_policy *, const double *in, size_t n, double *out, double init, int op, int exclusive);
/* stable radix sort on bits [sbit, ebit) of the sign-flipped key -- the order of zs::radix_sort / radix_sort_pair
 * (execution/ExecutionPolicy.hpp:765-781; cuda: cub::DeviceRadixSort, cuda/execution/ExecutionPolicy.cuh:755-881); vals may be NULL for
 * keys-only; kout may alias kin (the reference stages through temporaries).  Temporaries come from the policy's stream arena.
 * Up to 2 048 000 4-byte keys in contiguous arrays take three launches (top-digit split + buckets finished in LDS), everything else
 * one onesweep pass per 8 bits; the result is the same stable order either way. */
ZS_ROCM_EXPORT void zs_rocm_radix_sort_i32(zs_rocm_policy *, const int32_t *kin, const int32_t *vin, int32_t *kout, int32_t *vout, size_t n, int sbit, int ebit);
ZS_ROCM_EXPORT void zs_rocm_radix_sort_u32(zs_rocm_policy *, const uint32_t *kin, const int32_t *vin, uint32_t *kout, int32_t *vout, size_t n, int sbit, int ebit);
ZS_ROCM_EXPORT void zs_rocm_radix_sort_i64(zs_rocm_policy *, const int64_t *kin, const int32_t *vin, int64_t *kout, int32_t *vout, size_t n, int sbit, int ebit);
ZS_ROCM_EXPORT void zs_rocm_radix_sort_u64(zs_rocm_policy *, const uint64_t *kin, const int32_t *vin, uint64_t *kout, int32_t *vout, size_t n, int sbit, int ebit);
/* stable in-place merge sort (zs::merge_sort / merge_sort_pair, execution/ExecutionPolicy.hpp:745-761,
 * cuda/execution/ExecutionPolicy.cuh:698-752); `descending` != 0 sorts with `>`; vals may be NULL for keys-only.
 * User comparators: include/zensim_rocm/merge_sort.hpp through the C++ face. */
ZS_ROCM_EXPORT void zs_rocm_merge_sort_i32(zs_rocm_policy *, int32_t *keys, int32_t *vals, size_t n, int descending);
ZS_ROCM_EXPORT void zs_rocm_merge_sort_u32(zs_rocm_policy *, uint32_t *keys, int32_t *vals, size_t n, int descending);
ZS_ROCM_EXPORT void zs_rocm_merge_sort_i64(zs_rocm_policy *, int64_t *keys, int32_t *vals, size_t n, int descending);
ZS_ROCM_EXPORT void zs_rocm_merge_sort_u64(zs_rocm_policy *, uint64_t *keys, int32_t *vals, size_t n, int descending);
ZS_ROCM_EXPORT void zs_rocm_merge_sort_f32(zs_rocm_policy *, float *keys, int32_t *vals, size_t n, int descending);
ZS_ROCM_EXPORT void zs_rocm_merge_sort_f64(zs_rocm_policy *, double *keys, int32_t *vals, size_t n, int descending);

/* ======================================================================== (A) allocators & Vector */
/* py_interop/Allocator.cpp:5-21; memsrc_e: 0 = host, 1 = device, 2 = um (types/Property.h:7).
 * allocator_virtual = ZSPmrAllocator<true> over get_virtual_memory_source(mre, devid, reservedSpace, "STACK"): containers
 * created with it reserve `reservedSpace` bytes of address space (hipMemAddressReserve) and map physical memory as they
 * grow (hipMemCreate / hipMemMap), so their data pointer survives resize.  Every container entry point below also
 * exists with the reference's `_virtual` suffix; both spellings accept the same handle types. */
typedef struct zs_rocm_allocator zs_rocm_allocator;
ZS_ROCM_EXPORT zs_rocm_allocator *allocator(int memsrc, int8_t devid);
ZS_ROCM_EXPORT zs_rocm_allocator *allocator_virtual(int memsrc, int8_t devid, size_t reservedSpace);
ZS_ROCM_EXPORT void del_allocator(zs_rocm_allocator *);
ZS_ROCM_EXPORT void del_allocator_virtual(zs_rocm_allocator *);
ZS_ROCM_EXPORT int mem_enum__host(void);
ZS_ROCM_EXPORT int mem_enum__device(void);
ZS_ROCM_EXPORT int mem_enum__um(void);

/* py_interop/VectorInstantiations.cpp:8-170: zs::Vector<T> (container/Vector.hpp:11-421) */
typedef struct { void *_vector; } zs_rocm_vector_view_lite; /* VectorViewLite<T>, py_interop/VectorView.hpp:6-30 */
#define ZS_ROCM_DECL_VECTOR(T, SFX)                                                                       \
  ZS_ROCM_EXPORT zs_rocm_vector_##T *container__v_##T##SFX(zs_rocm_allocator *, size_t n);                \
  ZS_ROCM_EXPORT void del_container__v_##T##SFX(zs_rocm_vector_##T *);                                    \
  ZS_ROCM_EXPORT void relocate_container__v_##T##SFX(zs_rocm_vector_##T *, int memsrc, int8_t devid);     \
  ZS_ROCM_EXPORT void resize_container__v_##T##SFX(zs_rocm_vector_##T *, size_t n);                       \
  ZS_ROCM_EXPORT void reset_container__v_##T##SFX(zs_rocm_vector_##T *, int byteVal);                     \
  ZS_ROCM_EXPORT size_t container_size__v_##T##SFX(const zs_rocm_vector_##T *);                           \
  ZS_ROCM_EXPORT size_t container_capacity__v_##T##SFX(const zs_rocm_vector_##T *);                       \
  ZS_ROCM_EXPORT T get_val_container__v_##T##SFX(zs_rocm_vector_##T *);            /* getVal()      */   \
  ZS_ROCM_EXPORT void set_val_container__v_##T##SFX(zs_rocm_vector_##T *, T v);    /* setVal(v)     */   \
  ZS_ROCM_EXPORT T get_val_i_container__v_##T##SFX(zs_rocm_vector_##T *, size_t i);                       \
  ZS_ROCM_EXPORT void set_val_i_container__v_##T##SFX(zs_rocm_vector_##T *, size_t i, T v);              \
  ZS_ROCM_EXPORT void copy_to_container__v_##T##SFX(zs_rocm_vector_##T *, void *hostSrc);   /* assignVals   */ \
  ZS_ROCM_EXPORT void copy_from_container__v_##T##SFX(zs_rocm_vector_##T *, void *hostDst); /* retrieveVals */ \
  ZS_ROCM_EXPORT T *get_handle_container__v_##T##SFX(zs_rocm_vector_##T *);                               \
  ZS_ROCM_EXPORT zs_rocm_vector_view_lite *pyview__v_##T##SFX(zs_rocm_vector_##T *);                      \
  ZS_ROCM_EXPORT zs_rocm_vector_view_lite *pyview__v_const_##T##SFX(const zs_rocm_vector_##T *);          \
  ZS_ROCM_EXPORT aosoa_iterator_##T##_1 get_iterator_1__v_##T##SFX(zs_rocm_vector_##T *, uint32_t id);    \
  ZS_ROCM_EXPORT aosoa_iterator_const_##T##_1 get_iterator_1__v_const_##T##SFX(const zs_rocm_vector_##T *, uint32_t id); \
  ZS_ROCM_EXPORT aosoa_iterator_##T##_3 get_iterator_3__v_##T##SFX(zs_rocm_vector_##T *, uint32_t id);    \
  ZS_ROCM_EXPORT aosoa_iterator_const_##T##_3 get_iterator_3__v_const_##T##SFX(const zs_rocm_vector_##T *, uint32_t id);
#define ZS_ROCM_DECL_VECTOR_BOTH(T)                                                    \
  typedef struct zs_rocm_vector_##T zs_rocm_vector_##T;                                \
  ZS_ROCM_DECL_VECTOR(T, )                                                             \
  ZS_ROCM_DECL_VECTOR(T, _virtual)                                                     \
  ZS_ROCM_EXPORT void del_pyview__v_##T(zs_rocm_vector_view_lite *);                   \
  ZS_ROCM_EXPORT void del_pyview__v_const_##T(zs_rocm_vector_view_lite *);             \
  ZS_ROCM_EXPORT T *container_data__v_##T(zs_rocm_vector_##T *);
ZS_ROCM_DECL_VECTOR_BOTH(int)
ZS_ROCM_DECL_VECTOR_BOTH(float)
ZS_ROCM_DECL_VECTOR_BOTH(double)

/* ======================================================================== (A) TileVector */
/* py_interop/TileVectorInstantiations.cpp:8-215: property tags + zs::TileVector<T, L>
 * (container/TileVector.hpp:14-561).  Storage: element (chn, i) at (i/L*C + chn)*L + i%L. */
typedef struct zs_rocm_property_tags zs_rocm_property_tags;
ZS_ROCM_EXPORT zs_rocm_property_tags *property_tags(const char *const *names, const int *sizes, size_t n);
ZS_ROCM_EXPORT void del_property_tags(zs_rocm_property_tags *);
ZS_ROCM_EXPORT void property_tags_get_item(zs_rocm_property_tags *, size_t index, const char **name, size_t *size);
ZS_ROCM_EXPORT size_t property_tags_get_size(zs_rocm_property_tags *);
typedef struct {
  void *_vector;
  int _numChannels;
} zs_rocm_tv_view_lite; /* TileVectorViewLite<T, L>, py_interop/TileVectorView.hpp:9-131 */
typedef struct {
  void *_vector;
  int _numChannels;
  const char *_tagNames; /* [N] SmallString = char[32], in the container's memory space (TileVector.hpp:503-509) */
  const int *_tagOffsets;
  const int *_tagSizes;
  int _N;
} zs_rocm_tv_named_view_lite; /* TileVectorNamedViewLite<T, L>, py_interop/TileVectorView.hpp:133-287 */
#define ZS_ROCM_DECL_TILEVECTOR(T, L, SFX)                                                                      \
  ZS_ROCM_EXPORT zs_rocm_tv_##T##_##L *container__tv_##T##_##L##SFX(zs_rocm_allocator *,                        \
                                                                   const zs_rocm_property_tags *, size_t n);   \
  ZS_ROCM_EXPORT void del_container__tv_##T##_##L##SFX(zs_rocm_tv_##T##_##L *);                                 \
  ZS_ROCM_EXPORT void relocate_container__tv_##T##_##L##SFX(zs_rocm_tv_##T##_##L *, int memsrc, int8_t devid);  \
  ZS_ROCM_EXPORT void resize_container__tv_##T##_##L##SFX(zs_rocm_tv_##T##_##L *, size_t n);                    \
  ZS_ROCM_EXPORT void reset_container__tv_##T##_##L##SFX(zs_rocm_tv_##T##_##L *, int byteVal);                  \
  ZS_ROCM_EXPORT size_t container_size__tv_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *);                        \
  ZS_ROCM_EXPORT size_t container_capacity__tv_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *);                    \
  ZS_ROCM_EXPORT int property_offset__tv_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *, const char *name);        \
  ZS_ROCM_EXPORT int property_size__tv_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *, const char *name);          \
  ZS_ROCM_EXPORT aosoa_iterator_##T##_1 get_iterator_1__tv_##T##_##L##SFX(zs_rocm_tv_##T##_##L *, uint32_t id,  \
                                                                          uint32_t chnOffset);                 \
  ZS_ROCM_EXPORT aosoa_iterator_const_##T##_1 get_iterator_1__tv_const_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *, \
                                                                                      uint32_t id, uint32_t chnOffset); \
  ZS_ROCM_EXPORT aosoa_iterator_##T##_3 get_iterator_3__tv_##T##_##L##SFX(zs_rocm_tv_##T##_##L *, uint32_t id,  \
                                                                          uint32_t chnOffset);                 \
  ZS_ROCM_EXPORT aosoa_iterator_const_##T##_3 get_iterator_3__tv_const_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *, \
                                                                                      uint32_t id, uint32_t chnOffset); \
  ZS_ROCM_EXPORT zs_rocm_tv_view_lite *pyview__tv_##T##_##L##SFX(zs_rocm_tv_##T##_##L *);                       \
  ZS_ROCM_EXPORT zs_rocm_tv_view_lite *pyview__tv_const_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *);           \
  ZS_ROCM_EXPORT zs_rocm_tv_named_view_lite *pyview__tvn_##T##_##L##SFX(zs_rocm_tv_##T##_##L *);                \
  ZS_ROCM_EXPORT zs_rocm_tv_named_view_lite *pyview__tvn_const_##T##_##L##SFX(const zs_rocm_tv_##T##_##L *);    \
  /* py_interop/cuda/TileVectorUtility.cpp:7-30 -> append_channels (TileVector.hpp:583-623) */                 \
  ZS_ROCM_EXPORT void append_properties__rocm_tv_##T##_##L##SFX(zs_rocm_policy *, zs_rocm_tv_##T##_##L *,       \
                                                               const zs_rocm_property_tags *);
#define ZS_ROCM_DECL_TILEVECTOR_BOTH(T, L)                                                                 \
  typedef struct zs_rocm_tv_##T##_##L zs_rocm_tv_##T##_##L;                                                \
  ZS_ROCM_DECL_TILEVECTOR(T, L, )                                                                          \
  ZS_ROCM_DECL_TILEVECTOR(T, L, _virtual)                                                                  \
  ZS_ROCM_EXPORT void del_pyview__tv_##T##_##L(zs_rocm_tv_view_lite *);                                    \
  ZS_ROCM_EXPORT void del_pyview__tv_const_##T##_##L(zs_rocm_tv_view_lite *);                              \
  ZS_ROCM_EXPORT void del_pyview__tvn_##T##_##L(zs_rocm_tv_named_view_lite *);                             \
  ZS_ROCM_EXPORT void del_pyview__tvn_const_##T##_##L(zs_rocm_tv_named_view_lite *);                       \
  ZS_ROCM_EXPORT size_t container_num_channels__tv_##T##_##L(const zs_rocm_tv_##T##_##L *);                \
  ZS_ROCM_EXPORT T *container_data__tv_##T##_##L(zs_rocm_tv_##T##_##L *);                                  \
  /* (B) TileVector::reset(pol, val) TileVector.hpp:624-640 */                                            \
  ZS_ROCM_EXPORT void zs_rocm_fill__tv_##T##_##L(zs_rocm_policy *, zs_rocm_tv_##T##_##L *, T val);         \
  /* (B) TileVector::reorderTiles-style element reorder (TileVector.hpp:641-691): dst(:, i) = src(:, map[i]) \
     when gather != 0, dst(:, map[i]) = src(:, i) otherwise; all channels */                              \
  ZS_ROCM_EXPORT void zs_rocm_reorder__tv_##T##_##L(zs_rocm_policy *, zs_rocm_tv_##T##_##L *,              \
                                                   const int *map, int gather);
ZS_ROCM_DECL_TILEVECTOR_BOTH(int, 8)
ZS_ROCM_DECL_TILEVECTOR_BOTH(int, 32)
ZS_ROCM_DECL_TILEVECTOR_BOTH(int, 64)
ZS_ROCM_DECL_TILEVECTOR_BOTH(int, 512)
ZS_ROCM_DECL_TILEVECTOR_BOTH(float, 8)
ZS_ROCM_DECL_TILEVECTOR_BOTH(float, 32)
ZS_ROCM_DECL_TILEVECTOR_BOTH(float, 64)
ZS_ROCM_DECL_TILEVECTOR_BOTH(float, 512)
ZS_ROCM_DECL_TILEVECTOR_BOTH(double, 8)
ZS_ROCM_DECL_TILEVECTOR_BOTH(double, 32)
ZS_ROCM_DECL_TILEVECTOR_BOTH(double, 64)
ZS_ROCM_DECL_TILEVECTOR_BOTH(double, 512)

/* (B) raw AoSoA kernels on plain device pointers (what `pol(range(n), [tv = view<space>(tv)](i){...})`
 * load/store lambdas do, test/cuda/basic.cu:118-144).  tv has C channels of tile width L (power of 2). */
ZS_ROCM_EXPORT void zs_rocm_tv_from_aos_f32(zs_rocm_policy *, const float *aos, size_t n, int C, int L, float *tv);
ZS_ROCM_EXPORT void zs_rocm_tv_to_aos_f32(zs_rocm_policy *, const float *tv, size_t n, int C, int L, float *aos);
/* load every channel, scale by `alpha`, store back (BASELINE config 2 "AoSoA load/store": 8*C bytes/element) */
ZS_ROCM_EXPORT void zs_rocm_tv_scale_f32(zs_rocm_policy *, float *tv, size_t n, int C, int L, float alpha);
/* rows map[j] of an AoSoA buffer -> AoS rows j, and AoS rows j -> elements dstOffset + j of an AoSoA buffer: the pack / unpack
 * of the inter-rank particle migration ("ncclSend/Recv of AoSoA tiles after a per-destination radix partition") */
ZS_ROCM_EXPORT void zs_rocm_tv_gather_rows_f32(zs_rocm_policy *, const float *tv, const int *map, size_t m, int C, int L, float *aos);
ZS_ROCM_EXPORT void zs_rocm_tv_scatter_rows_f32(zs_rocm_policy *, const float *aos, size_t m, int C, int L, float *tv, size_t dstOffset);
ZS_ROCM_EXPORT void zs_rocm_tv_gather_f32(zs_rocm_policy *, const float *src, float *dst, size_t n, int C, int L, const int *map);
/* the same for the channels whose bit is set in channelMask only (C <= 64); the other channels of dst are left untouched */
ZS_ROCM_EXPORT void zs_rocm_tv_gather_channels_f32(zs_rocm_policy *, const float *src, float *dst, size_t n, int C, int L, const int *map,
                                                   unsigned long long channelMask);

/* ======================================================================== (A) bht */
/* py_interop/BhtInstantiations.cpp:6-128: zs::bht<int, dim, int, B>, dim 1-4, B = 16 | 32 (container/Bht.hpp:16-272).
 * Table layout identical to the reference: keys [tableSize] of next_2pow(dim) ints (unused/padding
 * ints hold 0x3f3f3f3f), indices [tableSize], status [tableSize] (all -1 outside of a build),
 * activeKeys [tableSize][dim], cnt, buildSuccess; tableSize = evaluateTableSize(n) (Bht.hpp:154-158);
 * hash functions seeded from std::mt19937(2) (Bht.hpp:165-169). */
typedef struct {
  void *keys;       /* storage_key_type* */
  int *indices;
  int *status;
  void *activeKeys; /* key_type* */
  int *cnt;
  int *success;
  uint32_t tableSize, numBuckets; /* size_type = u32 (BhtView.hpp:13); numBuckets = tableSize / B (:107) */
  uint32_t hf0x, hf0y, hf1x, hf1y, hf2x, hf2y;
} zs_rocm_bht_view_lite; /* BhtViewLite members, py_interop/BhtView.hpp:1054-1062; byte layout checked against
                            tests/golden/abi_layout.json (derived from the reference headers by tools/gen_abi_layout.sh) */
#define ZS_ROCM_DECL_BHT_A(D, B, SFX)                                                                         \
  ZS_ROCM_EXPORT zs_rocm_bht_##D *container__bht_int_##D##_int_##B##SFX(zs_rocm_allocator *, size_t n);        \
  ZS_ROCM_EXPORT void del_container__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *);                            \
  ZS_ROCM_EXPORT void relocate_container__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *, int memsrc, int8_t devid); \
  ZS_ROCM_EXPORT size_t container_size__bht_int_##D##_int_##B##SFX(const zs_rocm_bht_##D *);                   \
  ZS_ROCM_EXPORT size_t container_capacity__bht_int_##D##_int_##B##SFX(const zs_rocm_bht_##D *);               \
  ZS_ROCM_EXPORT void reset_container__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *, int clearCnt);            \
  ZS_ROCM_EXPORT zs_rocm_bht_view_lite *pyview__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *);                 \
  ZS_ROCM_EXPORT zs_rocm_bht_view_lite *pyview__bht_const_int_##D##_int_##B##SFX(const zs_rocm_bht_##D *);     \
  /* py_interop/cuda/BhtUtility.cpp:7-26 -> bht::resize (Bht.hpp:320-340) */                                  \
  ZS_ROCM_EXPORT void resize_container__rocm_bht_int_##D##_int_##B##SFX(zs_rocm_policy *, zs_rocm_bht_##D *,   \
                                                                       size_t newCapacity);
#define ZS_ROCM_DECL_BHT(D, B)                                                                           \
  ZS_ROCM_DECL_BHT_A(D, B, )                                                                             \
  ZS_ROCM_DECL_BHT_A(D, B, _virtual)                                                                     \
  ZS_ROCM_EXPORT void del_pyview__bht_int_##D##_int_##B(zs_rocm_bht_view_lite *);                         \
  /* (B) pol(range(n), [tb](i){ ret[i] = tb.insert(keys[i]); })  BHTView::insert, Bht.hpp:490-542 */     \
  ZS_ROCM_EXPORT void zs_rocm_insert__bht_int_##D##_int_##B(zs_rocm_policy *, zs_rocm_bht_##D *,          \
                                                          const int *keys, size_t n, int *ret);         \
  /* (B) table := { keys[i] -> index i }, cnt = n: adopt a partition numbered elsewhere (e.g. the _activeKeys of a  \
     zs::HashTable, container/HashTable.hpp, which the reference's in-tree P2G/G2P use); insert(key, i, enqueue)   \
     of Bht.hpp:490-542 with a fixed index */                                                                     \
  ZS_ROCM_EXPORT void zs_rocm_assign__bht_int_##D##_int_##B(zs_rocm_policy *, zs_rocm_bht_##D *,          \
                                                          const int *keys, size_t n);                   \
  /* (B) pol(range(n), [tb](i){ ret[i] = tb.query(keys[i]); })  BHTView::query, Bht.hpp:667-698 */       \
  ZS_ROCM_EXPORT void zs_rocm_query__bht_int_##D##_int_##B(zs_rocm_policy *, const zs_rocm_bht_##D *,     \
                                                         const int *keys, size_t n, int *ret);          \
  /* (B) bht::reorder(pol, map, scatter|gather), Bht.hpp:377-400 */                                      \
  ZS_ROCM_EXPORT void zs_rocm_reorder__bht_int_##D##_int_##B(zs_rocm_policy *, zs_rocm_bht_##D *,         \
                                                           const int *map, int scatter);                \
  /* (B) canonical form for bit-exact comparison (SURVEY.md 8a note): sort active keys               \
     lexicographically and renumber */                                                                  \
  ZS_ROCM_EXPORT void zs_rocm_canonicalize__bht_int_##D##_int_##B(zs_rocm_policy *, zs_rocm_bht_##D *);  \
  /* ... with the significance of the key's components chosen: axes = a permutation of 0 .. D-1 (host array), axes[0] \
     the most significant component, axes[D-1] the one that changes fastest along the numbering; NULL = 0, 1, ..       \
     Returns 0, -1 if axes is not a permutation */                                                                     \
  ZS_ROCM_EXPORT int zs_rocm_canonicalize_axes__bht_int_##D##_int_##B(zs_rocm_policy *, zs_rocm_bht_##D *, const int *axes);  \
  /* ... and only the entries [first, size): the head keeps its numbers (an MPM partition: the blocks that hold particles, numbered \
     before EnlargeSparsity appended the apron blocks -- sorting the tail makes the whole numbering reproducible) */              \
  ZS_ROCM_EXPORT int zs_rocm_canonicalize_tail__bht_int_##D##_int_##B(zs_rocm_policy *, zs_rocm_bht_##D *, const int *axes,     \
                                                                     size_t first);                                             \
  /* (B) renumber the active keys along the Z-order (Morton) curve of (key - min key): consecutive  \
     entries are spatial neighbours -- the launch order the per-block MPM kernels want (no reference  \
     counterpart: the reference's numbering is insertion order, Bht.hpp:612-664, and any numbering is \
     a valid table) */                                                                                  \
  ZS_ROCM_EXPORT void zs_rocm_order_morton__bht_int_##D##_int_##B(zs_rocm_policy *, zs_rocm_bht_##D *);
typedef struct zs_rocm_bht_1 zs_rocm_bht_1; /* one handle type per dim; the bucket size is a field of the object */
typedef struct zs_rocm_bht_2 zs_rocm_bht_2;
typedef struct zs_rocm_bht_3 zs_rocm_bht_3;
typedef struct zs_rocm_bht_4 zs_rocm_bht_4;
ZS_ROCM_DECL_BHT(1, 16)
ZS_ROCM_DECL_BHT(2, 16)
ZS_ROCM_DECL_BHT(3, 16)
ZS_ROCM_DECL_BHT(4, 16)
ZS_ROCM_DECL_BHT(1, 32)
ZS_ROCM_DECL_BHT(2, 32)
ZS_ROCM_DECL_BHT(3, 32)
ZS_ROCM_DECL_BHT(4, 32)

/* ======================================================================== (B) HashTable */
/* zs::HashTable<i32, dim, int> (container/HashTable.hpp:16-592), dim 1-4: the open-addressing table (64-bit hash_combine
 * chain over the raw coordinates `:496-500`, home = ((h % size) + size) % size, linear probing with stride 127 `:362`)
 * that `partition_for_particles` returns (simulation/sparsity/SparsityCompute.tpp:5-24) and the Grids-based MPM path keys
 * its blocks with (simulation/mpm/Simulator.cpp:122).  The reference exposes it only as a C++ template; these entry
 * points are the bulk forms of `pol(range(n), [t = proxy<space>(table)](i){ ... t.insert(key_i) ... })`.
 * Layout as in the reference: keys [tableSize][dim] packed ints (empty = INT_MAX), indices (-1 = empty), status (-1),
 * activeKeys [tableSize][dim], cnt; tableSize = next_2pow(n) * 16 (`:87-90`). */
typedef struct zs_rocm_hashtable zs_rocm_hashtable;
typedef struct {
  int *keys, *indices, *status, *activeKeys, *cnt;
  int tableSize;
} zs_rocm_hashtable_view; /* HashTableView members, HashTable.hpp:472-476 */
ZS_ROCM_EXPORT zs_rocm_hashtable *zs_rocm_hashtable_create(int dim, size_t numExpectedEntries, int memsrc, int devid);
ZS_ROCM_EXPORT void zs_rocm_hashtable_destroy(zs_rocm_hashtable *);
ZS_ROCM_EXPORT int zs_rocm_hashtable_dim(const zs_rocm_hashtable *);
ZS_ROCM_EXPORT size_t zs_rocm_hashtable_table_size(const zs_rocm_hashtable *); /* _tableSize */
ZS_ROCM_EXPORT int zs_rocm_hashtable_size(const zs_rocm_hashtable *);          /* size(): device -> host copy of cnt (:152) */
ZS_ROCM_EXPORT void zs_rocm_hashtable_get_view(const zs_rocm_hashtable *, zs_rocm_hashtable_view *out);
/* HashTable::reset(pol, clearCnt) (:294-299) == CleanSparsity with clearCnt = 1 (sparsity/SparsityOp.hpp:42-57) */
ZS_ROCM_EXPORT void zs_rocm_hashtable_reset(zs_rocm_policy *, zs_rocm_hashtable *, int clearCnt);
/* HashTableView::insert(key) (:353-374): ret[i] = dense index for the one inserter of a new key, -1 (sentinel_v) otherwise */
ZS_ROCM_EXPORT void zs_rocm_hashtable_insert(zs_rocm_policy *, zs_rocm_hashtable *, const int *keys, size_t n, int *ret);
/* HashTableView::insert(key, id) (:405-421): indices[slot] = ids[i] (ids == NULL: i); ok[i] = 1 if this call created the entry */
ZS_ROCM_EXPORT void zs_rocm_hashtable_insert_ids(zs_rocm_policy *, zs_rocm_hashtable *, const int *keys, const int *ids, size_t n, int *ok);
/* HashTableView::query / entry (:445-470): index / slot of a key, -1 if absent (table quiescent) */
ZS_ROCM_EXPORT void zs_rocm_hashtable_query(zs_rocm_policy *, const zs_rocm_hashtable *, const int *keys, size_t n, int *ret);
ZS_ROCM_EXPORT void zs_rocm_hashtable_entry(zs_rocm_policy *, const zs_rocm_hashtable *, const int *keys, size_t n, int *ret);
/* HashTable::resize / preserve (:258-292) */
ZS_ROCM_EXPORT void zs_rocm_hashtable_resize(zs_rocm_policy *, zs_rocm_hashtable *, size_t numExpectedEntries);
ZS_ROCM_EXPORT void zs_rocm_hashtable_preserve(zs_rocm_policy *, zs_rocm_hashtable *, size_t numExpectedEntries);

/* ======================================================================== (B) LBvh */
/* zs::LBvh<3, int, f32> (container/Bvh.hpp:86-492): linear BVH over AABBs, Karras (2012) topology on sorted 30-bit morton
 * codes, stored in depth-first pre-order with escape indices for stack-less traversal.  Boxes are AABBBox<3, f32> =
 * [n][6] floats {min xyz, max xyz} in device memory.  The reference exposes it only as a C++ template
 * (`bvh.build(pol, primBvs)`, `pol(range(n), [bvh = proxy<space>(bvh)](i){ bvh.iter_neighbors(bv, f); })`). */
typedef struct zs_rocm_lbvh zs_rocm_lbvh;
typedef struct {
  float *orderedBvs; /* [numNodes][6] */
  int *parents, *levels, *leafInds, *auxIndices;
  int numNodes, numLeaves;
} zs_rocm_lbvh_view; /* LBvhView members, Bvh.hpp:790-792 */
ZS_ROCM_EXPORT zs_rocm_lbvh *zs_rocm_lbvh_create(void);
ZS_ROCM_EXPORT void zs_rocm_lbvh_destroy(zs_rocm_lbvh *);
ZS_ROCM_EXPORT size_t zs_rocm_lbvh_num_leaves(const zs_rocm_lbvh *); /* getNumLeaves, :125 */
ZS_ROCM_EXPORT size_t zs_rocm_lbvh_num_nodes(const zs_rocm_lbvh *);  /* getNumNodes, :126-129: 2n-1, or n when n <= 2 */
ZS_ROCM_EXPORT void zs_rocm_lbvh_get_view(const zs_rocm_lbvh *, zs_rocm_lbvh_view *out);
/* LBvh::build(pol, primBvs, wrapv<Refit>) (:810-1082); n == 0 is a no-op, n <= 2 stores the boxes themselves */
ZS_ROCM_EXPORT void zs_rocm_lbvh_build(zs_rocm_policy *, zs_rocm_lbvh *, const float *primBvs, size_t n, int refit);
/* LBvh::refit (:1219-1248); returns -1 (the reference throws) when n differs from the built leaf count */
ZS_ROCM_EXPORT int zs_rocm_lbvh_refit(zs_rocm_policy *, zs_rocm_lbvh *, const float *primBvs, size_t n);
/* LBvh::getTotalBox (:152-171) -> box6 (device pointer, 6 floats) */
ZS_ROCM_EXPORT void zs_rocm_lbvh_total_box(zs_rocm_policy *, const zs_rocm_lbvh *, float *box6);
/* bulk LBvhView::iter_neighbors (:644-680): counts[q] = number of primitives whose box overlaps queryBvs[q]; then, with
 * offsets = exclusive_scan(counts), out[offsets[q] ...] = their ids in traversal order */
/* (r04: 16384 or more queries are walked in Morton order of their centres -- codes + one pair sort inside the call; every query's
 * result still goes to its own slot, hits in the walk's order) */
ZS_ROCM_EXPORT void zs_rocm_lbvh_query_count(zs_rocm_policy *, const zs_rocm_lbvh *, const float *queryBvs, size_t nq, int *counts);
ZS_ROCM_EXPORT void zs_rocm_lbvh_query_fill(zs_rocm_policy *, const zs_rocm_lbvh *, const float *queryBvs, size_t nq, const int *offsets,
                                            int *out);
/* self-collision broadphase (LBvhView::self_iter_neighbors, container/Bvh.hpp:695-728, driven over every leaf): thread k walks
 * from the k-th leaf in node order; counts[k] = overlapping leaves AFTER it (the leaf itself is skipped), so every unordered
 * pair of overlapping primitives appears exactly once.  fill: pairs[2*(offsets[k]+c)] = {primitive of leaf k, other primitive}.
 * The count pass remembers the first 32 hits of every leaf inside the LBvh object (132 B per leaf: 1.3 GB for 10 M leaves, allocated on
 * the first self query); a fill pass on the unchanged tree copies them instead of walking again (the memory is released with the
 * object, invalidated by build / refit).  `pairs` must be 8-byte aligned (any hipMalloc'ed array is).  `offsets` may be any per-leaf start
 * (the exclusive scan of `counts` is the fast case: a wave then writes its leaves' runs as one contiguous stream).  r04: one walk per WAVE over the
 * union of its 64 leaves' walks; every leaf still reports the same ids in the same order. */
ZS_ROCM_EXPORT void zs_rocm_lbvh_self_query_count(zs_rocm_policy *, const zs_rocm_lbvh *, int *counts /* [numLeaves] */);
ZS_ROCM_EXPORT void zs_rocm_lbvh_self_query_fill(zs_rocm_policy *, const zs_rocm_lbvh *, const int *offsets, int *pairs);

/* ======================================================================== (B) MPM transfers */
/* A particle attribute stored in an AoS zs::Vector<vec<T,N>> (geometry/Structurefree.hpp:21-237) or in
 * a TileVector channel group: component d of particle i lives at
 *   base + ((((i >> numTileBits) * numChns) << numTileBits) | (i & tileMask)) + d * (tileMask + 1)
 * (py_interop/GenericIterator.hpp:95-101). */
typedef aosoa_iterator_float_1 zs_rocm_attr;

typedef struct {
  zs_rocm_attr mass;  /* 1 */
  zs_rocm_attr pos;   /* 3 */
  zs_rocm_attr vel;   /* 3 */
  zs_rocm_attr C;     /* 9, column-major */
  zs_rocm_attr F;     /* 9, column-major */
  zs_rocm_attr logJp; /* 1, plastic models only (base may be NULL otherwise) */
  zs_rocm_attr stress; /* 6 = {xx, xy, xz, yy, yz, zz} of the symmetric P F^T * volume (the Kirchhoff stress times the volume:
                          symmetric for every isotropic model of P2G.hpp:82-101; r04, was 9), optional (base NULL = off): the
                          cached result of the constitutive update.  When present,
                          zs_rocm_mpm_g2p evaluates the model on the F it has just updated (and updates logJp) and stores the
                          result here, and zs_rocm_mpm_p2g reads it instead of re-running the 3x3 SVD: the per-particle
                          constitutive work of the reference's P2G (P2G.hpp:60-101) moves to the tail of the previous G2P, where
                          the VALU is idle.  Call zs_rocm_mpm_update_stress once before the first P2G and after any host-side
                          edit of F / logJp. */
  size_t n;
} zs_rocm_particles;

enum {
  ZS_MPM_FIXED_COROTATED = 0, ZS_MPM_DRUCKER_PRAGER = 1, ZS_MPM_VONMISES_FIXED_COROTATED = 2, ZS_MPM_NACC = 3,
  /* EquationOfStateConfig (weakly compressible fluid): the particles carry the volume ratio J instead of F -- pass the
   * 1-channel `J` attribute as zs_rocm_particles::F (only component 0 is read / written: G2P does J <- (1 + tr(C) dt) J,
   * simulation/transfer/G2P.hpp:70-74; P2G.hpp:60-81 forms the stress from J and C) */
  ZS_MPM_EQUATION_OF_STATE = 4
};
typedef struct {
  int model;          /* FixedCorotatedConfig | DruckerPragerConfig (physics/ConstitutiveModel.hpp:739-757) */
  float dx, dt;
  float volume, E, nu;
  float cohesion, beta, yieldSurface;
  int volCorrection;
  int side;           /* grid block side: 4 = Grids<f32,3,4> (geometry/Structure.hpp), 8 = SparseGrid<3,f32,8> */
  int keyIsOrigin;    /* 0: partition keys are block coordinates cell/side (Grids + HashTable/bht, simulation/Utils.hpp:24-31);
                         1: keys are block origins in cells, multiples of side (SparseGrid, geometry/SparseGrid.hpp:305-309) */
  float yieldStress;  /* VonMisesFixedCorotatedConfig::yieldStress (physics/ConstitutiveModel.hpp:745-749) */
  float xi, Msqr;     /* NACCConfig::xi, NACCConfig::Msqr() (:759-785; zs_rocm_nacc_msqr); NACC also uses E, nu, beta, logJp */
  int hardeningOn;    /* NACCConfig::hardeningOn */
  float bulk, viscosity; /* EquationOfStateConfig::bulk, ::viscosity (gamma is fixed to 7 by the formula, P2G.hpp:64-69) */
} zs_rocm_mpm_params;
/* NACCConfig::Msqr() for friction angle `fa` and dimension 3, evaluated as the reference does (physics/ConstitutiveModel.hpp:771-785) */
ZS_ROCM_EXPORT float zs_rocm_nacc_msqr(float fa);

/* grid: TileVector<f32, side^3> with 7 channels {m:1, v:3, rhs:3} (simulation/mpm/Simulator.cpp:116-122),
 * block b channel c cell k at grid[(b*7 + c)*side^3 + k], cell id = (x*side + y)*side + z
 * (geometry/Structure.hpp:323-333).  The partition is a bht<int,3,int,16> keyed by block coordinate
 * (cell coordinate / side), block number = table index. */

/* pol(range(n), ComputeSparsity{dx, side, table, X}) (sparsity/SparsityOp.hpp:59-87) */
ZS_ROCM_EXPORT void zs_rocm_mpm_compute_sparsity(zs_rocm_policy *, zs_rocm_bht_3 *, zs_rocm_attr pos, size_t n,
                                                 float dx, int side, int keyIsOrigin);
/* pol(range(nblocks), EnlargeSparsity{table, lo, hi}) (sparsity/SparsityOp.hpp:89-115) */
/* keyStride: 1 for block-coordinate keys, side for block-origin keys */
ZS_ROCM_EXPORT void zs_rocm_mpm_enlarge_sparsity(zs_rocm_policy *, zs_rocm_bht_3 *, const int lo[3], const int hi[3],
                                                 int keyStride);
/* ======================================================================== (B) IndexBuckets */
/* zs::IndexBuckets<3, i32, i32> (container/IndexBuckets.hpp:9-67) and `index_buckets_for_particles(pol, particles, dx,
 * displacement)` (simulation/particle/Query.tpp:9-58): a HashTable<i32,3,int> of occupied cells (key = floor(x/dx +
 * displacement), CleanSparsity + ComputeSparsity with blockLen 1, offset 0), counts[numBuckets+1] (SpatiallyCount,
 * simulation/sparsity/SparsityOp.hpp:117-152), offsets = exclusive_scan(counts), indices[n] = particle ids grouped by
 * bucket (SpatiallyDistribute, :154-196).  The reference fills a bucket in atomic race order; here ids are ascending
 * inside a bucket (= the sequential policy's result), so the structure is reproducible. */
typedef struct zs_rocm_index_buckets zs_rocm_index_buckets;
typedef struct {
  zs_rocm_hashtable *table; /* _table: bucket number = table.query(cell) */
  int *indices;             /* [numEntries] */
  int *offsets, *counts;    /* [numBuckets + 1] */
  int numBuckets, numEntries;
  float dx;
} zs_rocm_index_buckets_view; /* IndexBucketsView members, IndexBuckets.hpp:112-114 */
ZS_ROCM_EXPORT zs_rocm_index_buckets *zs_rocm_index_buckets_create(void);
ZS_ROCM_EXPORT void zs_rocm_index_buckets_destroy(zs_rocm_index_buckets *);
ZS_ROCM_EXPORT void zs_rocm_index_buckets_get_view(const zs_rocm_index_buckets *, zs_rocm_index_buckets_view *out);
/* `expectedCells` sizes the hash table (0: the particle count, as the reference does: tableSize = next_2pow(n) * 16) */
ZS_ROCM_EXPORT void zs_rocm_index_buckets_for_particles(zs_rocm_policy *, zs_rocm_index_buckets *, zs_rocm_attr pos, size_t n,
                                                        float dx, float displacement, size_t expectedCells);
/* Buckets over the cells of a block partition, for the gather-style transfers of a time loop: bucket number = block * side^3 + cell id
 * (the cell that contains the particle, displacement 0), counts / offsets / indices as above, but NO hash table is built (view.table ==
 * NULL, numBuckets = nblocks * side^3): zs_rocm_mpm_p2c2g finds a cell's bucket through the partition it is given anyway, which must be
 * this `table`.  Particles whose cell is not in the partition are listed in one extra bucket at index numBuckets that no transfer visits.
 * 4.4 -> 1.5 ms per rebuild at 16.7 M particles (no hash inserts). */
ZS_ROCM_EXPORT void zs_rocm_index_buckets_for_partition(zs_rocm_policy *, zs_rocm_index_buckets *, zs_rocm_attr pos, size_t n, float dx,
                                                        const zs_rocm_bht_3 *table, int side, int keyIsOrigin);
/* The same two functors on a zs::HashTable<i32,3,int> partition (their in-tree form: `partition_for_particles`,
 * simulation/sparsity/SparsityCompute.tpp:5-24 = CleanSparsity + ComputeSparsity(dx, blocklen, table, x), offset -2,
 * displacement 0.5; EnlargeSparsity sparsity/SparsityOp.hpp:89-115).  `zs_rocm_assign__bht_int_3_int_16(pol, bht,
 * view.activeKeys, size)` then adopts the block numbering for the binned transfers. */
ZS_ROCM_EXPORT void zs_rocm_mpm_partition_for_particles(zs_rocm_policy *, zs_rocm_hashtable *, zs_rocm_attr pos, size_t n,
                                                        float dx, int blocklen);
ZS_ROCM_EXPORT void zs_rocm_mpm_enlarge_sparsity__hashtable(zs_rocm_policy *, zs_rocm_hashtable *, const int lo[3],
                                                            const int hi[3]);

/* particle -> bin binning (the role of IndexBuckets / SpatiallyCount+Distribute,
 * sparsity/SparsityOp.hpp:117-196, simulation/particle/Query.tpp:9-58).  A bin is a 4x4x4 group of cells
 * (side 4: bin == grid block b; side 8: bin = 8*b + sub-cube).  Output: `order` [n], a permutation listing for
 * every bin, in [binStart[k], binStart[k+1]), the particles whose base node floor(x/dx - 0.5) lies in bin k,
 * round-robin over the bin's 64 cells (round r = the r-th particle of every cell that has one, cells in
 * x-major order); `cellCount` [nbins*64] particles per cell; binStart [nbins+1].  nbins = nblocks * (side/4)^3. */
ZS_ROCM_EXPORT void zs_rocm_mpm_bin_particles(zs_rocm_policy *, const zs_rocm_bht_3 *, zs_rocm_attr pos, size_t n,
                                              float dx, int side, int keyIsOrigin, int *order, int *binStart,
                                              unsigned *cellCount);
/* nbr[b][8]: block numbers of b + {0,1}^3 (x-major), -1 when absent */
ZS_ROCM_EXPORT void zs_rocm_mpm_build_neighbors(zs_rocm_policy *, const zs_rocm_bht_3 *, int *nbr, int keyStride);

/* pol(range(n), P2GTransfer{apic, dt, model, particles, table, grids})
 * (simulation/transfer/P2G.hpp:27-132, cuda/simulation/transfer/P2G.hpp:13-123).
 * binStart/cellCount/nbr == NULL: particle-order path (hash query + global float atomics per node, the
 * reference algorithm); otherwise the binned path: particles must be physically ordered by `order` of
 * zs_rocm_mpm_bin_particles; one wavefront per bin accumulates in registers + LDS and flushes once.  Particles
 * that moved out of their cell since binning are detected and take the particle-order path: results do not
 * depend on the freshness of the bins.  nblocks = number of grid blocks (table size). */
ZS_ROCM_EXPORT void zs_rocm_mpm_p2g(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles,
                                    const zs_rocm_bht_3 *, float *grid, size_t nblocks, const int *binStart,
                                    const unsigned *cellCount, const int *nbr);
/* pol(Collapse{nblocks, side^3}, ComputeGridBlockVelocity{grids, dt, extf, maxVel})
 * (simulation/grid/GridOp.hpp:71-108).  maxVelSqr may be NULL. */
ZS_ROCM_EXPORT void zs_rocm_mpm_grid_update(zs_rocm_policy *, const zs_rocm_mpm_params *, float *grid, size_t nblocks,
                                            const float extf[3], float *maxVelSqr);
/* Collider<AnalyticLevelSet<Plane|Cuboid|Sphere|Cylinder, f32, 3>> (geometry/Collider.h:10-206; the analytic members of
 * GeneralBoundary, :246-252).  param: plane {origin xyz, normal xyz}; cuboid {min xyz, max xyz}; sphere {centre xyz, radius};
 * cylinder {bottom centre xyz, radius, length, axis 0|1|2}.  x = R s X + b maps material to world space (R row-major). */
enum { ZS_ROCM_GEOM_PLANE = 0, ZS_ROCM_GEOM_CUBOID = 1, ZS_ROCM_GEOM_SPHERE = 2, ZS_ROCM_GEOM_CYLINDER = 3 };
enum { ZS_ROCM_COLLIDER_STICKY = 0, ZS_ROCM_COLLIDER_SLIP = 1, ZS_ROCM_COLLIDER_SEPARATE = 2 }; /* collider_e */
typedef struct zs_rocm_collider {
  int geometry, type;
  float param[8];
  float s, dsdt;
  float R[9];
  float omega[3];
  float b[3], dbdt[3];
} zs_rocm_collider;
/* identity transform, sticky */
ZS_ROCM_EXPORT void zs_rocm_collider_init(zs_rocm_collider *c, int geometry, int type, const float *param, int nparam);
/* pol(Collapse{nblocks, side^3}, ApplyBoundaryConditionOnGridBlocks{collider, partition, grids}) (simulation/grid/GridOp.hpp:111-164):
 * every node with mass > 0 at world position (block key * side + cell) * dx gets collider.resolveCollision(pos, vel).
 * Runs after zs_rocm_mpm_grid_update (the grid holds velocities). */
ZS_ROCM_EXPORT void zs_rocm_mpm_apply_boundary(zs_rocm_policy *, const zs_rocm_mpm_params *, const zs_rocm_bht_3 *, float *grid,
                                               size_t nblocks, const zs_rocm_collider *collider);
/* bulk Collider::resolveCollision on n points (x, v: [n][3] device arrays, v updated; inside[n] may be NULL): test entry */
ZS_ROCM_EXPORT void zs_rocm_collider_resolve(zs_rocm_policy *, const zs_rocm_collider *collider, const float *x, float *v, size_t n,
                                             int *inside);
/* Gather-style transfers: linear particle <-> cell-centre weights, 1/8 cell <-> node weights.
 * pol(Collapse{nblocks, side^3}, P2C2GTransfer{scheme, dt, model, buckets, particles, table, grids}): kind 0 = P2C2GTransfer
 * (simulation/transfer/P2C2G.hpp:29-305: mass, momentum with the affine term, and -dt * stress), 1 = P2C2GTransferMomentum (:321-506: mass and
 * momentum only), 2 = P2C2GTransferForce (:522-790: -dt * stress only).  ADDS into grid channels 0..3 (m, mv) like the reference's atomics.
 * `buckets` = zs_rocm_index_buckets_for_particles(pos, n, dx, displacement 0) or zs_rocm_index_buckets_for_partition(...) over this table;
 * particles.C is the reference's `B` (Structurefree.hpp:268-271: one storage).  Built from gathers only (per bucket -> per particle -> per
 * 2x2x2 cells -> per node): no float atomics, reproducible bit for bit; each particle's constitutive update runs once (the reference functor repeats it, and its logJp store, for every cell that sees the particle).
 * Returns 0, -1 on bad arguments. */
ZS_ROCM_EXPORT int zs_rocm_mpm_p2c2g(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_index_buckets *buckets,
                                     const zs_rocm_bht_3 *, float *grid, size_t nblocks, int kind);
/* pol(range(n), PreG2C2PTransfer{particles}) (simulation/transfer/G2C2P.hpp:208-221): v = 0, C = 0 */
ZS_ROCM_EXPORT void zs_rocm_mpm_pre_g2c2p(zs_rocm_policy *, zs_rocm_particles);
/* pol(Collapse{nblocks, side^3}, G2C2PTransfer{scheme, dt, model, buckets, grids, table, particles}) (G2C2P.hpp:38-204): v_p += sum W v_c,
 * B_p += sum W (v_c (x) x_i - v_c (x) x_p); `buckets` may be NULL (the particle side is a gather here). */
ZS_ROCM_EXPORT int zs_rocm_mpm_g2c2p(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_index_buckets *buckets,
                                     const zs_rocm_bht_3 *, const float *grid, size_t nblocks);
/* pol(range(n), PostG2C2PTransfer{scheme, dt, dx, model, particles}) (G2C2P.hpp:224-275): C = B Dinv; F <- (I + dt C) F (J for the fluid);
 * x += v dt */
ZS_ROCM_EXPORT void zs_rocm_mpm_post_g2c2p(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles);
/* The three G2C2P functors above in one pass over the particles (v and B start from 0 instead of being zeroed, re-read and re-written):
 * same bits as zs_rocm_mpm_pre_g2c2p + zs_rocm_mpm_g2c2p + zs_rocm_mpm_post_g2c2p. */
ZS_ROCM_EXPORT int zs_rocm_mpm_g2c2p_step(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_bht_3 *,
                                          const float *grid, size_t nblocks);
/* pol(range(n), G2PTransfer{apic, dt, model, grids, table, particles}) (simulation/transfer/G2P.hpp:24-90) */
ZS_ROCM_EXPORT void zs_rocm_mpm_g2p(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles,
                                    const zs_rocm_bht_3 *, const float *grid, size_t nblocks, const int *binStart,
                                    const unsigned *cellCount, const int *nbr);
/* ---- slotted particle storage: the motion-robust form of the fused step (zpc_amd/csrc/mpm_slotted.hip).  Storage = bins x K rounds x
 * 64 lanes in ONE TileVector<f32, 64> (slot (bin, r, lane) = element (bin K + r) 64 + lane), cellMask[bin][lane] = occupied rounds of the
 * cell; a particle is always stored under the cell of its base node, and the step keeps it so.  A particle that changes cell is finished
 * by the workgroup that moves it: inside the bin it draws a ticket for the lowest free round of its new cell and is scattered by that
 * cell's lane; across bins its grid terms are added by global float atomics and a record goes to the bin's outbox (outboxCap records
 * per bin), from where a second, small kernel copies it into a free round of the destination cell (one ticket per record on claim[]),
 * and a third folds tickets and departures into cellMask.  No re-bin, no exact-path queues in the time loop.  Buffers: moverCount (int),
 * claim (unsigned; zeroed by the caller ONCE, every step leaves it zero), moverRec (float) -- their sizes in bytes come from
 * zs_rocm_mpm_slot_outbox_bytes(nbins, outboxCap, which = 0 / 1 / 2).  status: int[ZS_ROCM_SLOT_STATUS_WORDS], zeroed by the caller,
 * latched: [0] an outbox was full, [1] a
 * cell ran out of rounds, [2] mass or a mover for a block outside the partition, [3] early warning: a particle lives in a block flagged by
 * blockEdge (re-partition within `side` steps), [4] a particle was not stored under its cell, or moved more than one cell in a step;
 * [8, 264) and [264, 520): movers sent / re-homed (running sums spread over 256 words each: a single device-wide word would serialise one
 * atomic per bin); the totals are equal after every step.  NO PARTICLE IS EVER DROPPED (the reference writes every particle back,
 * G2P.hpp:67-82): a mover that finds its outbox full, its new cell full (inside the bin or, r04, in another bin: slot_rehome_kernel puts
 * it back), its destination block missing from the partition (r04), or that has moved more than a cell KEEPS its old slot with its new
 * state; it is then stored under the wrong cell and skipped ([4]) until the caller re-slots the storage (unslot / re-partition / slot),
 * which the flags [0], [1], [2] ask for.  Only GRID contributions can be dropped ([2]: a stencil node whose block is missing; the
 * reference does not check that either, P2G.hpp:109-110).
 * The per-particle arithmetic is that of zs_rocm_mpm_g2p2g (G2P.hpp:44-83 + P2G.hpp:51-125). */
#define ZS_ROCM_SLOT_STATUS_WORDS (8 + 2 * 256)
ZS_ROCM_EXPORT size_t zs_rocm_mpm_slot_outbox_bytes(size_t nbins, int outboxCap, int which); /* 0 moverCount, 1 claim words, 2 moverRec */
ZS_ROCM_EXPORT void zs_rocm_mpm_build_neighbors27(zs_rocm_policy *, const zs_rocm_bht_3 *, int *nbr27 /* [nblocks][27] */, int keyStride);
/* src: TileVector<f32,64> with C channels and n particles in any order -> dst: nbins*K tiles; status: int[ZS_ROCM_SLOT_STATUS_WORDS] ([1], [2] are used here) */
ZS_ROCM_EXPORT int zs_rocm_mpm_slot_particles(zs_rocm_policy *, const zs_rocm_bht_3 *, zs_rocm_attr pos, size_t n, float dx, int side,
                                              int keyIsOrigin, int K, const float *src, float *dst, int C, unsigned *cellMask, int *status);
/* occupied slots in slot order (slots may be NULL: count only); synchronises */
ZS_ROCM_EXPORT size_t zs_rocm_mpm_slot_list(zs_rocm_policy *, const unsigned *cellMask, size_t nbins, int K, int *slots);
/* the slotted storage's buffers in one POD (all device pointers, owned by the caller) */
typedef struct zs_rocm_slot_storage {
  unsigned *cellMask; /* [nbins][64] */
  int K;
  const int *nbr;     /* [nblocks][8]  zs_rocm_mpm_build_neighbors */
  const int *nbr27;   /* [nblocks][27] zs_rocm_mpm_build_neighbors27 */
  int *moverCount;
  unsigned *claim;
  float *moverRec;
  int outboxCap;
  int *status;        /* [ZS_ROCM_SLOT_STATUS_WORDS] */
  const unsigned char *blockEdge; /* [nblocks] from zs_rocm_mpm_partition_edge, or NULL (no early warning in status[3]) */
} zs_rocm_slot_storage;
/* edge[i] = 1 if one of the blocks at offsets [lo, hi]^3 of block i is not in the partition (lo = -1, hi = 2: a particle of a block with
 * edge == 0 can move into any neighbouring block and still scatter its whole stencil) */
ZS_ROCM_EXPORT void zs_rocm_mpm_partition_edge(zs_rocm_policy *, const zs_rocm_bht_3 *, unsigned char *edge /* [nblocks] */, int keyStride,
                                               int lo, int hi);
/* Re-partition of slotted storage in place (r05; no reference counterpart -- the reference never orders particles): the partition a fused time
 * loop runs on has to follow the particles (closed-loop trigger: status word [3]).  (1) the reference's ComputeSparsity
 * (simulation/sparsity/SparsityOp.hpp:65-86) taken from the occupancy words instead of the particle positions -- a particle is stored under the
 * cell of its base node -- into `newTab` (a freshly reset table); enlarge it with zs_rocm_mpm_enlarge_sparsity as usual.  (2) every bin that
 * holds particles moves, as whole tile rows, to the number its block has in `newTab`; `newMask` and (optionally) `newGrid` -- the node
 * velocities the next step gathers from -- are written for the whole new partition.  No particle is read, re-binned or re-slotted:
 * ~4 ms instead of ~50 ms per re-partition of the 64 Mi-particle column.  Buffers: newBuf = nbins(new) * K * 64 * C floats, newMask = nbins(new) * 64 words,
 * newGrid = nblocks(new) * 7 * side^3 floats.  Returns 0 / -1 (bad arguments); status[2] if a populated block is missing from `newTab`.
 * Only the slots that `newMask` selects are written in newBuf: every other row keeps whatever the buffer held (zs_rocm_mpm_slot_particles
 * zero-fills empty slots, this call does not) -- after a re-partition in place "a slot is defined iff its occupancy bit is set" is the
 * storage's invariant; every kernel of the path reads masked slots only. */
ZS_ROCM_EXPORT void zs_rocm_mpm_slot_compute_sparsity(zs_rocm_policy *, const zs_rocm_bht_3 *oldTab, const unsigned *cellMask, size_t nblocksOld,
                                                      int side, int keyIsOrigin, zs_rocm_bht_3 *newTab);
ZS_ROCM_EXPORT int zs_rocm_mpm_reslot(zs_rocm_policy *, const zs_rocm_bht_3 *oldTab, const zs_rocm_bht_3 *newTab, int side, int K, int C,
                                      const float *oldBuf, float *newBuf, const unsigned *oldMask, unsigned *newMask, const float *oldGrid,
                                      float *newGrid, int *status);
/* the step over blocks [blockBegin, blockEnd) on the storage `st` (see zs_rocm_mpm_g2p2g_slotted_range for finish) */
ZS_ROCM_EXPORT int zs_rocm_mpm_g2p2g_slots(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_bht_3 *,
                                           const float *gridA, float *gridB, size_t nblocks, const zs_rocm_slot_storage *st, int writeAll,
                                           size_t blockBegin, size_t blockEnd, int finish);
ZS_ROCM_EXPORT int zs_rocm_mpm_g2p2g_slotted(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_bht_3 *,
                                             const float *gridA, float *gridB, size_t nblocks, unsigned *cellMask, int K, const int *nbr,
                                             const int *nbr27, int *moverCount, unsigned *claim, float *moverRec, int outboxCap,
                                             int writeAll, int *status);
/* the same over blocks [blockBegin, blockEnd); finish != 0 (ONCE per step, with or after the last range): the step's outbox records get
 * their slots and departures / arrivals enter the occupancy words.  Multi-GPU: boundary blocks first, their ghost sums travel while the
 * interior range computes (zs_rocm_mpm_g2p2g_range is the compact-storage counterpart) */
ZS_ROCM_EXPORT int zs_rocm_mpm_g2p2g_slotted_range(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_bht_3 *,
                                                   const float *gridA, float *gridB, size_t nblocks, unsigned *cellMask, int K, const int *nbr,
                                                   const int *nbr27, int *moverCount, unsigned *claim, float *moverRec, int outboxCap,
                                                   int writeAll, int *status, size_t blockBegin, size_t blockEnd, int finish);
/* particles.stress := model(F, logJp) * volume, logJp updated (no-op when particles.stress.base == NULL) */
/* Fused transfer: G2P of step n (from gridA: velocities after zs_rocm_mpm_grid_update) and P2G of step n+1 (into gridB,
 * zeroed by the caller) in one pass over the binned particles -- the reference's G2P2GTransfer idea
 * (simulation/transfer/G2P2G.hpp:20-148).  Same results as zs_rocm_mpm_g2p followed by zs_rocm_mpm_p2g; HBM traffic per
 * particle drops from 296.5 B to 116.5 B because v, C and the cached stress stay on chip.  Requires binned particles and
 * the `stress` attribute (state of the particles that take the exact path is parked there).  writeAll != 0 also writes
 * v, C, stress of every particle (e.g. on the last step).  Returns 0, or -1 if the requirements are not met. */
ZS_ROCM_EXPORT int zs_rocm_mpm_g2p2g(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_bht_3 *,
                                     const float *gridA, float *gridB, size_t nblocks, const int *binStart,
                                     const unsigned *cellCount, const int *nbr, int writeAll);
/* The same over blocks [blockBegin, blockEnd) only (multi-GPU step: the blocks near a rank boundary are numbered first and
 * launched first, their ghost sums travel while the interior range computes).  driftFlag (device int[3], may be NULL):
 * [0] is set to 1 when an exact-path particle sits more than one 4^3 bin away from the bin it is stored in (the margin that
 * keeps interior blocks from touching shared blocks no longer holds: re-bin); [1] accumulates the number of particles the
 * call handled on the exact path (moved out of their cell since the last re-bin) -- the caller's re-bin trigger; [2] is set
 * to 1 when a particle's stencil reached a node whose block is not in the partition (the reference does not check,
 * P2G.hpp:109-110; here the contribution is dropped and the caller is told to rebuild the partition). */
ZS_ROCM_EXPORT int zs_rocm_mpm_g2p2g_range(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles, const zs_rocm_bht_3 *,
                                           const float *gridA, float *gridB, size_t nblocks, const int *binStart,
                                           const unsigned *cellCount, const int *nbr, int writeAll, size_t blockBegin,
                                           size_t blockEnd, int *driftFlag);
/* Fused step that re-bins on the fly.  binStart / cellCount describe the NEW binned order (zs_rocm_mpm_bin_particles on the
 * current positions, `order` = its permutation: slot i holds the particle stored at order[i]); the step's inputs m, x, F, logJp
 * are read from slot order[i] of particlesIn, everything the step stores (incl. m) goes to slot i of particlesOut.  Both
 * particle sets must be two buffers of one layout.  The physical re-bin thus costs no pass of its own, and every particle
 * starts the step in the cell it is binned in. */
ZS_ROCM_EXPORT int zs_rocm_mpm_g2p2g_reorder_range(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles particlesOut,
                                                   zs_rocm_particles particlesIn, const int *order, const zs_rocm_bht_3 *,
                                                   const float *gridA, float *gridB, size_t nblocks, const int *binStart,
                                                   const unsigned *cellCount, const int *nbr, int writeAll, size_t blockBegin,
                                                   size_t blockEnd, int *driftFlag);
ZS_ROCM_EXPORT void zs_rocm_mpm_update_stress(zs_rocm_policy *, const zs_rocm_mpm_params *, zs_rocm_particles);
/* channels the library expects behind zs_rocm_particles.stress (6 since r04; 9 before): a caller that allocates the attribute checks this
 * once -- a 9-channel attribute handed to a 6-channel library would be misread silently */
ZS_ROCM_EXPORT int zs_rocm_mpm_stress_channels(void);
/* per-particle constitutive update alone (physics/ConstitutiveModel_Vol_dP.hpp:10-47,246-326):
 * PF[9n] AoS out; F (and logJp) updated in place for plastic models.  Test/diagnostic entry point. */
ZS_ROCM_EXPORT void zs_rocm_mpm_stress(zs_rocm_policy *, const zs_rocm_mpm_params *, float *F, float *logJp, size_t n, float *PF);
ZS_ROCM_EXPORT void zs_rocm_svd3(zs_rocm_policy *, const float *F, size_t n, float *U, float *S, float *V);

/* ghost-block halo exchange helpers for the multi-GPU domain decomposition (no reference counterpart:
 * zpc has no collective layer).  pack: buf[i][c][k] = grid[blocks[i]][chn0+c][k];
 * unpack: add = 0 set, 1 add (every block listed once), 2 atomic add (a block may be listed several times: concatenated
 * messages of several peers). */
/* owner rank of every particle for the block-aligned spatial split of the multi-GPU driver: cell = floor(x / dx) clamped to
 * [lo, hi); axis d is cut into dims[d] slabs at lo + (len k / dims[d]) rounded down to a multiple of `align` cells;
 * rank = (r0 * dims[1] + r1) * dims[2] + r2 */
ZS_ROCM_EXPORT void zs_rocm_mpm_owner_rank(zs_rocm_policy *, zs_rocm_attr pos, size_t n, float dx, const int lo[3], const int hi[3],
                                           const int dims[3], int align, int *owner);
/* counts[r] (device, `world` ints, world <= 1024) = number of entries of owner[0 .. n) equal to r */
ZS_ROCM_EXPORT void zs_rocm_mpm_owner_counts(zs_rocm_policy *, const int *owner, size_t n, int world, int *counts);
ZS_ROCM_EXPORT void zs_rocm_mpm_halo_pack(zs_rocm_policy *, const float *grid, const int *blocks, size_t nb, int side,
                                          int chn0, int nchn, float *buf);
ZS_ROCM_EXPORT void zs_rocm_mpm_halo_unpack(zs_rocm_policy *, float *grid, const int *blocks, size_t nb, int side,
                                            int chn0, int nchn, const float *buf, int add);


/* ======================================================================== (B) multi-GPU exchange steps on RCCL (zpc_amd/csrc/dist.hip)
 * One process per GPU, one communicator per process; the reference has no collective layer, these are the exchange
 * steps the spatially sharded MPM path has: ghost-block halo exchange (grouped ncclSend / ncclRecv over the point-to-point xGMI
 * links), allreduce(max) for the CFL step, counts + uneven all-to-all for particle migration.  All calls enqueue on the policy's
 * stream.  Return 0, or -1 after an RCCL error (also latched in zs_rocm_last_error). */
typedef struct zs_rocm_dist zs_rocm_dist;
ZS_ROCM_EXPORT size_t zs_rocm_dist_unique_id_bytes(void);
ZS_ROCM_EXPORT int zs_rocm_dist_unique_id(void *out); /* rank 0; the launcher distributes the bytes */
ZS_ROCM_EXPORT zs_rocm_dist *zs_rocm_dist_create(int rank, int world, const void *uniqueId, int device);
ZS_ROCM_EXPORT void zs_rocm_dist_destroy(zs_rocm_dist *);
ZS_ROCM_EXPORT int zs_rocm_dist_rank(const zs_rocm_dist *);
ZS_ROCM_EXPORT int zs_rocm_dist_world(const zs_rocm_dist *);
/* ncclCommCount of the communicator (-1: no communicator): zs_rocm_dist_create already fails when it differs from `world` */
ZS_ROCM_EXPORT int zs_rocm_dist_comm_count(const zs_rocm_dist *);
ZS_ROCM_EXPORT int zs_rocm_dist_halo_exchange(zs_rocm_dist *, zs_rocm_policy *, float *grid, int side, int chn0, int nchn, const int *blocks,
                                              size_t totalBlocks, int npeers, const int *peerRank, const size_t *peerOffset,
                                              const size_t *peerCount, float *sendbuf, float *recvbuf);
ZS_ROCM_EXPORT int zs_rocm_dist_allreduce_f32(zs_rocm_dist *, zs_rocm_policy *, float *buf, size_t n, int op); /* 0 sum, 1 max, 2 min */
ZS_ROCM_EXPORT int zs_rocm_dist_allreduce_i64(zs_rocm_dist *, zs_rocm_policy *, long long *buf, size_t n, int op);
ZS_ROCM_EXPORT int zs_rocm_dist_alltoall_i64(zs_rocm_dist *, zs_rocm_policy *, const long long *send, long long *recv);
ZS_ROCM_EXPORT int zs_rocm_dist_alltoallv_f32(zs_rocm_dist *, zs_rocm_policy *, const float *send, const size_t *sendCounts,
                                              const size_t *sendOffsets, float *recv, const size_t *recvCounts, const size_t *recvOffsets);
ZS_ROCM_EXPORT int zs_rocm_dist_barrier(zs_rocm_dist *, zs_rocm_policy *);
/* The halo plan: which of this rank's grid blocks do other ranks hold as well (set at partition-build time).  _from_keys is the pure host
 * part (callable without a GPU): keysAll = the ranks' key lists back to back (counts[r] keys of 3 ints, each list in its rank's block-number
 * order); per peer that shares blocks with `rank`, in rank order: the shared keys in lexicographic order as positions in this rank's list
 * (both sides of a pair derive the same order).  peerRank / peerOffset / peerCount: capacity world - 1; blocks: capacity = the return value
 * (call with NULL outputs to size).  _create is collective: RCCL all-gather of the key lists (keys = the partition's activeKeys, device),
 * plan on the host, exchange buffers allocated; _exchange = zs_rocm_dist_halo_exchange over the plan's lists and buffers. */
typedef struct zs_rocm_halo_plan zs_rocm_halo_plan;
ZS_ROCM_EXPORT size_t zs_rocm_halo_plan_from_keys(const int *keysAll, const size_t *counts, int world, int rank, int *npeers, int *peerRank,
                                                  size_t *peerOffset, size_t *peerCount, int *blocks);
ZS_ROCM_EXPORT zs_rocm_halo_plan *zs_rocm_dist_halo_plan_create(zs_rocm_dist *, zs_rocm_policy *, const int *keys, size_t nblocks, int side);
ZS_ROCM_EXPORT void zs_rocm_dist_halo_plan_destroy(zs_rocm_halo_plan *);
ZS_ROCM_EXPORT int zs_rocm_dist_halo_plan_npeers(const zs_rocm_halo_plan *);
ZS_ROCM_EXPORT size_t zs_rocm_dist_halo_plan_blocks(const zs_rocm_halo_plan *);
ZS_ROCM_EXPORT size_t zs_rocm_dist_halo_plan_bytes(const zs_rocm_halo_plan *);   /* bytes sent (= received) per exchange of all 7 channels */
ZS_ROCM_EXPORT const int *zs_rocm_dist_halo_plan_block_list(const zs_rocm_halo_plan *);   /* device */
ZS_ROCM_EXPORT int zs_rocm_dist_halo_plan_exchange(zs_rocm_halo_plan *, zs_rocm_dist *, zs_rocm_policy *, float *grid, int chn0, int nchn);
/* a plan from explicit lists (host arrays): slice k of blocks[total] = local block numbers shared with rank peerRank[k] */
ZS_ROCM_EXPORT zs_rocm_halo_plan *zs_rocm_dist_halo_plan_from_lists(zs_rocm_dist *, int side, int npeers, const int *peerRank,
                                                                    const size_t *peerOffset, const size_t *peerCount, const int *blocks,
                                                                    size_t total);
/* One sub-step of the slotted MPM path in ONE call (zpc_amd/csrc/dist.hip): gridB := 0; fused G2P (gridA) + P2G (gridB) over the boundary
 * blocks [0, nBoundary), then the interior blocks (+ re-home / commit); the ghost-block exchange of the plan on commPolicy's stream behind
 * the boundary range, overlapping the interior (rangeSchedule: as two launches, or as one whose boundary workgroups count themselves off);
 * grid update of gridB (extf, maxVelSqr) [+ collider]; allreduce(max) of maxVelSqr.  Nothing
 * of the caller runs between the kernels.  dist / plan / commPolicy / maxVelSqr / collider / haloGrid may be NULL (single rank: no exchange).
 * haloGrid: the grid whose shared blocks are exchanged (NULL: gridB).  The reference's building blocks for such a schedule are
 * pol.device(i) / .stream(i) / .listen() (cuda/execution/ExecutionPolicy.cuh:364-399).
 * A non-zero return is FATAL for the object being stepped: when the exchange fails in the overlapped schedule the particle side of the step
 * has still been committed (both ranges ran, re-home and commit included, so that the slot storage stays consistent) but gridB has neither
 * been completed nor updated -- the particles are one step ahead of the grids.  Do not retry the step (it would advance the particles
 * twice) and do not continue from this state. */
typedef struct zs_rocm_mpm_step {
  const zs_rocm_mpm_params *params;
  zs_rocm_particles particles;
  const zs_rocm_bht_3 *table;
  const float *gridA;
  float *gridB;
  size_t nblocks;
  const zs_rocm_slot_storage *storage;
  int writeAll;
  float extf[3];
  float *maxVelSqr;
  const zs_rocm_collider *collider;
  size_t nBoundary;
  zs_rocm_dist *dist;
  zs_rocm_halo_plan *plan;
  zs_rocm_policy *commPolicy;
  float *haloGrid;
  void *evTransferBegin, *evTransferEnd; /* hipEvent_t or NULL: recorded on the policy's stream around the transfer kernels (timing) */
  void **evBreakdown;                    /* NULL, or ZS_ROCM_STEP_EVENTS hipEvent_t (timing enabled, created on the policy's device) recorded along
                                            the step -- on the policy's stream: [0] start, [1] boundary range done, [2] interior range + re-home +
                                            commit done, [5] exchange waited for, [6] grid update done, [7] CFL allreduce done; on commPolicy's
                                            stream: [3] exchange started, [4] exchange done (without overlap [3], [4] are recorded on the policy's
                                            stream around the exchange and [1] == [0]).  A rank's step time splits into
                                            boundary [0,1], interior [1,2], waiting for the exchange [2,5], grid update [5,6], allreduce [6,7];
                                            the exchange itself is [3,4] */
  int haloChannels;                      /* grid channels [0, haloChannels) of the shared blocks are exchanged; 0 = all 7.  4 = {m, mv}: all a step
                                            ever reads of a ghost block -- the grid update forms v = mv / m + extf dt and G2P gathers v, exactly as
                                            ComputeGridBlockVelocity / G2PTransfer of the reference do (simulation/grid/GridOp.hpp:90-104); the rhs
                                            channels 4..6 of a shared block then keep this rank's partial sums */
  int rangeSchedule;                     /* overlapped schedule only (commPolicy set, 0 < nBoundary < nblocks): how the boundary blocks' sums get to
                                            the exchange stream early.
                                            ZS_ROCM_RANGES_IN_TURN (0): two launches on the policy's stream, boundary range then interior range.
                                            ZS_ROCM_RANGES_SIDE_BY_SIDE (1): the boundary range on commPolicy's stream (give it the higher
                                            priority) in front of the exchange, the interior range on the policy's stream at the same time; re-home /
                                            commit wait for both.  The interior's workgroups fill the CUs the boundary range's last workgroups
                                            leave idle, but the two launches share the CUs: the boundary range ends later than it would alone.
                                            ZS_ROCM_RANGES_ONE_LAUNCH (2; 8^3 blocks): ONE launch over all blocks -- workgroups are dispatched in
                                            block order, the boundary blocks come first -- whose boundary workgroups count themselves off on a
                                            device word; a one-wave gate kernel on commPolicy's stream waits for the count and the exchange runs
                                            behind it.  No launch boundary inside the step: no tail, no sharing.
                                            evBreakdown[1] (boundary done) is recorded on commPolicy's stream for 1 and 2; [0,1] and [0,2] overlap. */
  float *handoverSnapshot;               /* NULL, or (test hook, overlapped schedule) plan-blocks x side^3 floats: the MASS channel of gridB's shared blocks, packed like the
                                            exchange buffer (zs_rocm_mpm_halo_pack, chn0 = 0, nchn = 1) on commPolicy's stream at the moment the exchange starts.  Only the
                                            boundary blocks write those nodes, so it must equal the same pack taken after the step: the check that the hand-over of a
                                            schedule lets the exchange see complete sums (tests/test_dist_gpu.py) */
} zs_rocm_mpm_step;
#define ZS_ROCM_STEP_EVENTS 8
#define ZS_ROCM_RANGES_IN_TURN 0
#define ZS_ROCM_RANGES_SIDE_BY_SIDE 1
#define ZS_ROCM_RANGES_ONE_LAUNCH 2
ZS_ROCM_EXPORT int zs_rocm_mpm_step_slotted(zs_rocm_policy *, const zs_rocm_mpm_step *);

#ifdef __cplusplus
}
#endif
#endif /* ZS_ROCM_H */
