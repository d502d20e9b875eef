// zs_rocm.hpp -- header-only C++ face of the MI355X backend: the subset of zpc's `zs::` API that user translation
// units (Zeno nodes, tests) touch on the execution-policy hot path, filling the `execspace_e::rocm` slot the
// reference reserves (types/Property.h:28-47, TypeAlias.hpp:117-121, execution/ExecutionPolicy.hpp:43-44).
// Compile user TUs with hipcc --offload-arch=gfx950 and link libzsrocm.so.
//
//   zs::rocm_exec()                         -> RocmExecutionPolicy  (cuda/execution/ExecutionPolicy.cuh:345-918)
//   pol(zs::range(n), f) / pol(zs::Collapse{n}, f) / {nb, nt} / {nb, ntiles, tileSize}
//                                            the launcher shapes of ExecutionPolicy.cuh:212-343, incl. the
//                                            (shmem*, ...) arity dispatch of :40-155
//   zs::reduce / exclusive_scan / inclusive_scan / radix_sort / radix_sort_pair  (execution/ExecutionPolicy.hpp:684-781)
//   zs::Vector<T>, zs::TileVector<T,L>, zs::bht<int,dim,int,16> + zs::view<zs::execspace_e::rocm>(c) / proxy
//   zs::atomic_* / shfl / ballot / thread_fence overloads on rocm_exec_tag  (execution/Atomics.hpp, Intrinsics.hpp)
//
// Names, argument meaning and defaults follow the reference; wave-level intrinsics are 64 lanes wide with 64-bit
// masks (the reference's cuda overloads assume 32).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <initializer_list>
#include <limits>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../zs_rocm.h"
#include "bht_device.hpp"
#include "merge_sort.hpp"
#include "hashtable_device.hpp"
#include "lbvh_device.hpp"

#define ZS_LAMBDA __device__
#define ZS_FUNCTION __forceinline__ __host__ __device__

namespace zs {

// ------------------------------------------------------------------------------------ tags (types/Property.h)
template <auto V> struct wrapv {
  static constexpr auto value = V;
  constexpr operator decltype(V)() const noexcept { return V; }
};
enum struct execspace_e : unsigned char { host = 0, seq = 0, openmp, cuda, musa, rocm, sycl };
using rocm_exec_tag = wrapv<execspace_e::rocm>;
constexpr auto rocm_c = rocm_exec_tag{};
constexpr auto exec_rocm = rocm_c;
enum struct memsrc_e : unsigned char { host = 0, device, um };
using ProcID = signed char;
using StreamID = int;

// ------------------------------------------------------------------------------------ functors (ZpcFunctional.hpp)
template <class T = void> struct plus { ZS_FUNCTION T operator()(T a, T b) const { return a + b; } };
template <class T = void> struct multiplies { ZS_FUNCTION T operator()(T a, T b) const { return a * b; } };
template <class T = void> struct less {
  template <class A, class B> ZS_FUNCTION bool operator()(const A &a, const B &b) const { return a < b; }
};
template <class T = void> struct greater {
  template <class A, class B> ZS_FUNCTION bool operator()(const A &a, const B &b) const { return a > b; }
};
template <class T = void> struct getmin { ZS_FUNCTION T operator()(T a, T b) const { return a < b ? a : b; } };
template <class T = void> struct getmax { ZS_FUNCTION T operator()(T a, T b) const { return a > b ? a : b; } };
namespace detail {
  template <class Op> struct op_code;
  template <class T> struct op_code<plus<T>> { static constexpr int value = 0; };
  template <class T> struct op_code<multiplies<T>> { static constexpr int value = 1; };
  template <class T> struct op_code<getmin<T>> { static constexpr int value = 2; };
  template <class T> struct op_code<getmax<T>> { static constexpr int value = 3; };
}  // namespace detail

// ------------------------------------------------------------------------------------ zs::tuple (ZpcTuple.hpp:119-365)
// The parameter pack of pol(range, params, f) travels to the device BY VALUE as a kernel argument (cuda/execution/ExecutionPolicy.cuh:
// 281-322), so it is an aggregate of its elements: one indexed holder per element, reached by get<I>() / zs::get<I>(t) / structured
// bindings.  make_tuple decays its arguments like the reference's (arrays and functions excepted: not kernel arguments).
namespace detail {
  template <std::size_t I, class T> struct tuple_slot { T v; };
  template <class Seq, class... Ts> struct tuple_slots;
  template <std::size_t... Is, class... Ts> struct tuple_slots<std::index_sequence<Is...>, Ts...> : tuple_slot<Is, Ts>... {
    constexpr tuple_slots() = default;
    template <class... Us, std::enable_if_t<sizeof...(Us) == sizeof...(Ts) && (sizeof...(Us) > 0), int> = 0>
    ZS_FUNCTION constexpr tuple_slots(Us &&...us) : tuple_slot<Is, Ts>{static_cast<Us &&>(us)}... {}
  };
  template <std::size_t I, class T> ZS_FUNCTION constexpr T &slot_of(tuple_slot<I, T> &s) { return s.v; }
  template <std::size_t I, class T> ZS_FUNCTION constexpr const T &slot_of(const tuple_slot<I, T> &s) { return s.v; }
  template <std::size_t I, class T> T slot_type(const tuple_slot<I, T> &);
}  // namespace detail
template <class... Ts> struct tuple : detail::tuple_slots<std::index_sequence_for<Ts...>, Ts...> {
  using base_t = detail::tuple_slots<std::index_sequence_for<Ts...>, Ts...>;
  static constexpr std::size_t tuple_size = sizeof...(Ts);
  constexpr tuple() = default;
  template <class... Us, std::enable_if_t<sizeof...(Us) == sizeof...(Ts) && (sizeof...(Us) > 0), int> = 0>
  ZS_FUNCTION constexpr tuple(Us &&...us) : base_t(static_cast<Us &&>(us)...) {}
  template <std::size_t I> ZS_FUNCTION constexpr auto &get() { return detail::slot_of<I>(*this); }
  template <std::size_t I> ZS_FUNCTION constexpr const auto &get() const { return detail::slot_of<I>(*this); }
};
template <class... Ts> tuple(Ts...) -> tuple<Ts...>;
template <class... Args> ZS_FUNCTION constexpr tuple<std::decay_t<Args>...> make_tuple(Args &&...args) {
  return tuple<std::decay_t<Args>...>(static_cast<Args &&>(args)...);
}
template <std::size_t I, class... Ts> ZS_FUNCTION constexpr auto &get(tuple<Ts...> &t) { return t.template get<I>(); }
template <std::size_t I, class... Ts> ZS_FUNCTION constexpr const auto &get(const tuple<Ts...> &t) { return t.template get<I>(); }
template <class T> struct tuple_size;
template <class... Ts> struct tuple_size<tuple<Ts...>> : std::integral_constant<std::size_t, sizeof...(Ts)> {};
template <class T> inline constexpr std::size_t tuple_size_v = tuple_size<T>::value;
template <std::size_t I, class T> struct tuple_element;
template <std::size_t I, class... Ts> struct tuple_element<I, tuple<Ts...>> {
  using type = decltype(detail::slot_type<I>(std::declval<const tuple<Ts...> &>()));
};
template <std::size_t I, class T> using tuple_element_t = typename tuple_element<I, T>::type;
template <class T> struct is_tuple : std::false_type {};
template <class... Ts> struct is_tuple<tuple<Ts...>> : std::true_type {};

// ------------------------------------------------------------------------------------ ranges (ZpcIterator.hpp:504-704)
template <int N> struct CollapseN { long long n[N]; };
struct Collapse {
  long long n[3];
  int dim;
  constexpr Collapse(long long a) : n{a, 1, 1}, dim(1) {}
  constexpr Collapse(long long a, long long b) : n{a, b, 1}, dim(2) {}
  constexpr Collapse(long long a, long long b, long long c) : n{a, b, c}, dim(3) {}
};
struct index_range { long long b, e; };
constexpr index_range range(long long n) { return {0, n}; }
constexpr index_range range(long long b, long long e) { return {b, e}; }

// ------------------------------------------------------------------------------------ atomics (execution/Atomics.hpp:27-392)
template <class T> __device__ __forceinline__ T atomic_add(rocm_exec_tag, T *dst, T val) { return atomicAdd(dst, val); }
__device__ __forceinline__ float atomic_add(rocm_exec_tag, float *dst, float val) { return unsafeAtomicAdd(dst, val); }
template <class T> __device__ __forceinline__ T atomic_cas(rocm_exec_tag, T *dst, T expected, T desired) {
  return atomicCAS(dst, expected, desired);
}
template <class T> __device__ __forceinline__ T atomic_exch(rocm_exec_tag, T *dst, T val) { return atomicExch(dst, val); }
template <class T> __device__ __forceinline__ T atomic_or(rocm_exec_tag, T *dst, T val) { return atomicOr(dst, val); }
template <class T> __device__ __forceinline__ T atomic_and(rocm_exec_tag, T *dst, T val) { return atomicAnd(dst, val); }
template <class T> __device__ __forceinline__ T atomic_xor(rocm_exec_tag, T *dst, T val) { return atomicXor(dst, val); }
template <class T> __device__ __forceinline__ void atomic_min(rocm_exec_tag, T *dst, T val) {
  if constexpr (std::is_floating_point_v<T>) {  // CAS loop as in the reference (Atomics.hpp:330-360)
    using U = std::conditional_t<sizeof(T) == 4, unsigned, unsigned long long>;
    U *p = (U *)dst, old = *p, assumed;
    do {
      assumed = old;
      T cur;
      __builtin_memcpy(&cur, &assumed, sizeof(T));
      if (!(val < cur)) break;
      U want;
      __builtin_memcpy(&want, &val, sizeof(T));
      old = atomicCAS(p, assumed, want);
    } while (assumed != old);
  } else
    atomicMin(dst, val);
}
template <class T> __device__ __forceinline__ void atomic_max(rocm_exec_tag, T *dst, T val) {
  if constexpr (std::is_floating_point_v<T>) {
    using U = std::conditional_t<sizeof(T) == 4, unsigned, unsigned long long>;
    U *p = (U *)dst, old = *p, assumed;
    do {
      assumed = old;
      T cur;
      __builtin_memcpy(&cur, &assumed, sizeof(T));
      if (!(val > cur)) break;
      U want;
      __builtin_memcpy(&want, &val, sizeof(T));
      old = atomicCAS(p, assumed, want);
    } while (assumed != old);
  } else
    atomicMax(dst, val);
}
// wave-aggregated increment (Atomics.hpp:113-130): one atomic per wave, 64-bit ballot
__device__ __forceinline__ int atomic_inc(rocm_exec_tag, int *dst) {
  const unsigned long long m = __ballot(1);
  const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(dst, __popcll(m));
  base = __shfl(base, leader, 64);
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

// ------------------------------------------------------------------------------------ intrinsics (execution/Intrinsics.hpp:40-196)
__device__ __forceinline__ void thread_fence(rocm_exec_tag) { __threadfence(); }
__device__ __forceinline__ void sync_threads(rocm_exec_tag) { __syncthreads(); }
__device__ __forceinline__ unsigned long long active_mask(rocm_exec_tag) { return __ballot(1); }
__device__ __forceinline__ unsigned long long ballot_sync(rocm_exec_tag, unsigned long long, int pred) { return __ballot(pred); }
template <class T> __device__ __forceinline__ T shfl_sync(rocm_exec_tag, unsigned long long, T v, int src, int width = 64) { return __shfl(v, src, width); }
template <class T> __device__ __forceinline__ T shfl_up_sync(rocm_exec_tag, unsigned long long, T v, int d, int width = 64) { return __shfl_up(v, d, width); }
template <class T> __device__ __forceinline__ T shfl_down_sync(rocm_exec_tag, unsigned long long, T v, int d, int width = 64) { return __shfl_down(v, d, width); }
template <class T> __device__ __forceinline__ T shfl_xor_sync(rocm_exec_tag, unsigned long long, T v, int m, int width = 64) { return __shfl_xor(v, m, width); }
__device__ __forceinline__ int count_lz(rocm_exec_tag, unsigned x) { return __clz((int)x); }
__device__ __forceinline__ int count_lz(rocm_exec_tag, unsigned long long x) { return __clzll((long long)x); }
__device__ __forceinline__ int count_ones(rocm_exec_tag, unsigned long long x) { return __popcll(x); }

// ------------------------------------------------------------------------------------ containers
namespace detail {
  template <class T> struct c_name;  // C-ABI type tag
  template <> struct c_name<int> { static constexpr int id = 0; };
  template <> struct c_name<float> { static constexpr int id = 1; };
  template <> struct c_name<double> { static constexpr int id = 2; };
}  // namespace detail

// memory locations and allocators (resource/Resource.h:29-146, 199-245; types/Property.h): a container is built as
// Container{allocator, n}; get_memory_source(memsrc, devid) names plain device / unified / host memory,
// get_temporary_memory_source(pol) the stream-ordered scratch of a policy's stream
struct MemoryLocation {
  memsrc_e _memsrc = memsrc_e::device;
  ProcID _devid = 0;
  constexpr memsrc_e memspace() const { return _memsrc; }
  constexpr ProcID devid() const { return _devid; }
};
struct ZSPmrAllocator {
  MemoryLocation location{};
};
inline ZSPmrAllocator get_memory_source(memsrc_e mre, ProcID devid = 0) { return {MemoryLocation{mre, devid}}; }
// get_temporary_memory_source(pol) (resource/Resource.h:52-58 -> temporary_memory_resource<device_mem_tag>, cuda/memory/Allocator.h:33-50):
// stream-ordered allocate / deallocate on the policy's stream; a block stays valid until deallocate and never aliases another live one
struct TemporaryMemorySource {
  zs_rocm_policy *_h;
  void *allocate(std::size_t bytes, std::size_t = 256) const { return zs_rocm_policy_temporary(_h, bytes); }
  void deallocate(void *p, std::size_t, std::size_t = 256) const { zs_rocm_policy_temporary_free(_h, p); }
};
// zs::Vector<T> (container/Vector.hpp:11-421) for device / um memory; T in {int, float, double} own their storage through
// the C ABI, any other trivially copyable T through hipMalloc directly.
template <class T> struct Vector {
  using value_type = T;
  Vector(std::size_t n = 0, memsrc_e mre = memsrc_e::device, ProcID devid = 0) : _size(n), _cap(n), _mre(mre) {
    if (n) {
      if (mre == memsrc_e::um) (void)hipMallocManaged((void **)&_data, n * sizeof(T));
      else if (mre == memsrc_e::device) (void)hipMalloc((void **)&_data, n * sizeof(T));
      else _data = (T *)std::malloc(n * sizeof(T));
    }
    (void)devid;
  }
  // Vector{allocator, n} (Vector.hpp:47-60)
  Vector(const ZSPmrAllocator &a, std::size_t n) : Vector(n, a.location.memspace(), a.location.devid()) {}
  Vector(const MemoryLocation &l, std::size_t n) : Vector(n, l.memspace(), l.devid()) {}
  Vector(const TemporaryMemorySource &a, std::size_t n) : _size(n), _cap(n), _mre(memsrc_e::device), _tmp(a._h) {
    if (n) _data = (T *)zs_rocm_policy_temporary(a._h, n * sizeof(T));
  }
  ~Vector() { release(); }
  Vector(const Vector &o) : Vector(o._size, o._mre) {
    if (_size) (void)hipMemcpy(_data, o._data, _size * sizeof(T), hipMemcpyDefault);
  }
  Vector(Vector &&o) noexcept { swap(o); }
  Vector &operator=(Vector o) { swap(o); return *this; }
  void swap(Vector &o) {
    std::swap(_data, o._data); std::swap(_size, o._size); std::swap(_cap, o._cap); std::swap(_mre, o._mre); std::swap(_tmp, o._tmp);
  }
  // element access on the host side: host / unified memory only (device vectors go through getVal / setVal)
  T &operator[](std::size_t i) { return _data[i]; }
  const T &operator[](std::size_t i) const { return _data[i]; }
  const T *begin() const { return _data; }
  const T *end() const { return _data + _size; }
  Vector clone(const MemoryLocation &l) const { return clone(l.memspace()); }
  std::size_t size() const { return _size; }
  std::size_t capacity() const { return _cap; }
  T *data() { return _data; }
  const T *data() const { return _data; }
  T *begin() { return _data; }
  T *end() { return _data + _size; }
  memsrc_e memspace() const { return _mre; }
  T getVal(std::size_t i = 0) const {  // 1-element copy (Vector.hpp:189-200)
    T r;
    (void)hipMemcpy(&r, _data + i, sizeof(T), hipMemcpyDefault);
    return r;
  }
  void setVal(const T &v, std::size_t i = 0) { (void)hipMemcpy(_data + i, &v, sizeof(T), hipMemcpyDefault); }
  void reset(int ch) { if (_size) (void)hipMemset(_data, ch, _size * sizeof(T)); }
  void resize(std::size_t n) {  // geometric growth x1.5 (Vector.hpp:228-256,407-416)
    if (n <= _cap) { _size = n; return; }
    std::size_t g = _cap + _cap / 2;
    Vector tmp(g > n ? g : n, _mre);
    if (_size) (void)hipMemcpy(tmp._data, _data, _size * sizeof(T), hipMemcpyDefault);
    tmp._size = n;
    swap(tmp);
  }
  // retrieveVals / assignVals (Vector.hpp: whole-range copies out of / into the container), push_back / append (host-side growth)
  void retrieveVals(T *dst) const { if (_size) (void)hipMemcpy(dst, _data, _size * sizeof(T), hipMemcpyDefault); }
  void assignVals(const T *src) { if (_size) (void)hipMemcpy(_data, src, _size * sizeof(T), hipMemcpyDefault); }
  void push_back(const T &v) { resize(_size + 1); setVal(v, _size - 1); }
  void append(const Vector &o) {
    const std::size_t old = _size;
    resize(old + o._size);
    if (o._size) (void)hipMemcpy(_data + old, o._data, o._size * sizeof(T), hipMemcpyDefault);
  }
  Vector clone(memsrc_e mre) const {
    Vector r(_size, mre);
    if (_size) (void)hipMemcpy(r._data, _data, _size * sizeof(T), hipMemcpyDefault);
    return r;
  }
private:
  void release() {
    if (!_data) return;
    if (_tmp) zs_rocm_policy_temporary_free(_tmp, _data);
    else if (_mre == memsrc_e::host) std::free(_data);
    else (void)hipFree(_data);
    _data = nullptr;
  }
  T *_data = nullptr;
  std::size_t _size = 0, _cap = 0;
  memsrc_e _mre = memsrc_e::device;
  zs_rocm_policy *_tmp = nullptr;  // set: stream-ordered temporary of that policy (get_temporary_memory_source)
};
// ZS_ENABLE_OFB_ACCESS_CHECK (container/Vector.hpp:471-480, TileVector.hpp:738-767; off by default like the reference's build option):
// an out-of-range access prints the reference's message and returns a reference to the top of the address space, so that the access
// faults instead of corrupting a neighbour
#ifndef ZS_ENABLE_OFB_ACCESS_CHECK
#define ZS_ENABLE_OFB_ACCESS_CHECK 0
#endif
namespace detail {
template <class T> ZS_FUNCTION T &ofb_sentinel() { return *reinterpret_cast<T *>(~std::uintptr_t(0) - sizeof(T) + 1); }
}  // namespace detail
template <class T> struct VectorView {
  T *_p;
  std::size_t _n;
  ZS_FUNCTION T &operator[](std::size_t i) const {
#if ZS_ENABLE_OFB_ACCESS_CHECK
    if (i >= _n) {
      printf("vector [%s] ofb! accessing %lld out of [0, %lld)\n", "", (long long)i, (long long)_n);
      return detail::ofb_sentinel<T>();
    }
#endif
    return _p[i];
  }
  ZS_FUNCTION T &operator()(std::size_t i) const { return (*this)[i]; }
  ZS_FUNCTION std::size_t size() const { return _n; }
};

// ------------------------------------------------------------------------------------ iterators and ranges
// (ZpcIterator.hpp:504-560, 637-704; container/TileVector.hpp:299-322; container/Vector.hpp begin / end)
// POD random-access iterators that are passed to kernels by value; zip / enumerate ranges whose dereference is a tuple of
// references; range(container), range(tilevector, "property") and range(n).
template <class T, int L> struct TileVectorIterator {  // one channel of a TileVector: element i at ((i / L) C + chn) L + i % L
  using iterator_category = std::random_access_iterator_tag;
  using value_type = std::remove_const_t<T>;
  using difference_type = long long;
  using pointer = T *;
  using reference = T &;
  T *_base;  // channel offset folded in
  long long _i;
  int _C;
  ZS_FUNCTION reference operator*() const { return (*this)[0]; }
  ZS_FUNCTION reference operator[](difference_type k) const {
    const unsigned long long j = (unsigned long long)(_i + k);
    return _base[(j / L) * (unsigned long long)_C * L + j % L];
  }
  ZS_FUNCTION TileVectorIterator operator+(difference_type k) const { return {_base, _i + k, _C}; }
  ZS_FUNCTION TileVectorIterator operator-(difference_type k) const { return {_base, _i - k, _C}; }
  ZS_FUNCTION difference_type operator-(const TileVectorIterator &o) const { return _i - o._i; }
  ZS_FUNCTION TileVectorIterator &operator++() { ++_i; return *this; }
  ZS_FUNCTION TileVectorIterator &operator+=(difference_type k) { _i += k; return *this; }
  ZS_FUNCTION bool operator==(const TileVectorIterator &o) const { return _i == o._i && _base == o._base; }
  ZS_FUNCTION bool operator!=(const TileVectorIterator &o) const { return !(*this == o); }
};
struct IndexIterator {  // range(n): dereferences to the index itself
  using iterator_category = std::random_access_iterator_tag;
  using value_type = long long;
  using difference_type = long long;
  using pointer = const long long *;
  using reference = long long;
  long long _i;
  ZS_FUNCTION long long operator*() const { return _i; }
  ZS_FUNCTION long long operator[](long long k) const { return _i + k; }
  ZS_FUNCTION IndexIterator operator+(long long k) const { return {_i + k}; }
  ZS_FUNCTION long long operator-(const IndexIterator &o) const { return _i - o._i; }
  ZS_FUNCTION bool operator!=(const IndexIterator &o) const { return _i != o._i; }
};
template <class It> struct iterator_range {
  It _b, _e;
  It begin() const { return _b; }
  It end() const { return _e; }
};
template <class... Its> struct zip_iterator {  // dereference: std::tuple of the member iterators' references
  using iterator_category = std::random_access_iterator_tag;
  using difference_type = long long;
  using reference = std::tuple<typename std::iterator_traits<Its>::reference...>;
  using value_type = reference;
  using pointer = void;
  std::tuple<Its...> iters;
  template <std::size_t... Is> ZS_FUNCTION reference deref(long long k, std::index_sequence<Is...>) const {
    return reference{std::get<Is>(iters)[k]...};
  }
  ZS_FUNCTION reference operator[](long long k) const { return deref(k, std::index_sequence_for<Its...>{}); }
  ZS_FUNCTION reference operator*() const { return (*this)[0]; }
  ZS_FUNCTION difference_type operator-(const zip_iterator &o) const { return std::get<0>(iters) - std::get<0>(o.iters); }
};
template <class T> struct is_zip_iterator : std::false_type {};
template <class... Its> struct is_zip_iterator<zip_iterator<Its...>> : std::true_type {};
namespace detail {
  template <class R> auto range_begin(R &&r) { return std::begin(r); }
  template <class R> auto range_end(R &&r) { return std::end(r); }
}  // namespace detail
inline iterator_range<IndexIterator> range(index_range r) { return {{r.b}, {r.e}}; }
template <class T> iterator_range<T *> range(Vector<T> &v) { return {v.data(), v.data() + v.size()}; }
template <class T> iterator_range<const T *> range(const Vector<T> &v) { return {v.data(), v.data() + v.size()}; }
template <class It> std::size_t range_size(const iterator_range<It> &r) { return (std::size_t)(r.end() - r.begin()); }
template <class T> std::size_t range_size(const Vector<T> &v) { return v.size(); }
// zip(r0, r1, ...): ranges or containers with begin() / end(); the length is that of the first
template <class... Rs> auto zip(Rs &&...rs) {
  using ZI = zip_iterator<std::decay_t<decltype(std::begin(rs))>...>;
  return iterator_range<ZI>{ZI{{std::begin(rs)...}}, ZI{{std::end(rs)...}}};
}
// enumerate(r0, ...) = zip(range(n), r0, ...)
template <class R0, class... Rs> auto enumerate(R0 &&r0, Rs &&...rs) {
  const long long n = (long long)(std::end(r0) - std::begin(r0));
  return zip(iterator_range<IndexIterator>{{0}, {n}}, std::forward<R0>(r0), std::forward<Rs>(rs)...);
}

struct PropertyTag {
  std::string name;
  int numChannels;
};
template <class T, int N> struct small_vec {
  T v[N];
  ZS_FUNCTION T &operator[](int i) { return v[i]; }
  ZS_FUNCTION const T &operator[](int i) const { return v[i]; }
  ZS_FUNCTION T &operator()(int i) { return v[i]; }
  ZS_FUNCTION const T &operator()(int i) const { return v[i]; }
};
// zs::ndrange<d>(n) (ZpcIterator.hpp): the index tuples of {0..n-1}^d, first index slowest -- `for (auto loc : ndrange<3>(3))` walks a
// stencil in the order the transfers use (simulation/transfer/P2G.hpp:106, P2C2G.hpp:79); get<I>(loc) or loc[I] reads a component
template <int d> struct ndrange_t {
  int n;
  struct iterator {
    int i, n;
    ZS_FUNCTION small_vec<int, d> operator*() const {
      small_vec<int, d> r{};
      int k = i;
      for (int a = d - 1; a >= 0; --a) { r.v[a] = k % n; k /= n; }
      return r;
    }
    ZS_FUNCTION iterator &operator++() { ++i; return *this; }
    ZS_FUNCTION bool operator!=(const iterator &o) const { return i != o.i; }
  };
  ZS_FUNCTION iterator begin() const { return {0, n}; }
  ZS_FUNCTION iterator end() const {
    int t = 1;
    for (int a = 0; a < d; ++a) t *= n;
    return {t, n};
  }
};
template <int d> ZS_FUNCTION constexpr ndrange_t<d> ndrange(int n) { return {n}; }
template <int I, class T, int N> ZS_FUNCTION constexpr const T &get(const small_vec<T, N> &v) { return v.v[I]; }
template <int... Ns> struct dim_t {};
template <int... Ns> constexpr dim_t<Ns...> dim_c{};

// zs::TileVector<T, L> (container/TileVector.hpp): AoSoA, element (chn, i) at (i/L*C + chn)*L + i%L
template <class T, int L> struct TileVectorView {
  T *_p;
  std::size_t _n;
  int _C;
  static constexpr int lane_width = L;
  ZS_FUNCTION T &operator()(int chn, std::size_t i) const {
#if ZS_ENABLE_OFB_ACCESS_CHECK
    if (chn < 0 || chn >= _C) {
      printf("tilevector [%s] ofb! accessing chn [%d] out of [0, %d)\n", "", chn, _C);
      return detail::ofb_sentinel<T>();
    }
    if (i >= _n) {
      printf("tilevector [%s] ofb! global accessing ele [%lld] out of [0, %lld)\n", "", (long long)i, (long long)_n);
      return detail::ofb_sentinel<T>();
    }
#endif
    return _p[(i / L * _C + chn) * L + i % L];
  }
  ZS_FUNCTION T &operator()(int chn, std::size_t tile, int lane) const {
#if ZS_ENABLE_OFB_ACCESS_CHECK
    if (chn < 0 || chn >= _C) {
      printf("tilevector [%s] ofb! accessing chn [%d] out of [0, %d)\n", "", chn, _C);
      return detail::ofb_sentinel<T>();
    }
    if (lane < 0 || lane >= L) {
      printf("tilevector [%s] ofb! in-tile accessing ele [%lld] out of [0, %lld)\n", "", (long long)lane, (long long)L);
      return detail::ofb_sentinel<T>();
    }
#endif
    return _p[(tile * _C + chn) * L + lane];
  }
  template <int N> ZS_FUNCTION small_vec<T, N> pack(dim_t<N>, int chn, std::size_t i) const {  // TileVector.hpp:897-943
    small_vec<T, N> r;
    const T *b = &(*this)(chn, i);
#pragma unroll
    for (int d = 0; d < N; ++d) r.v[d] = b[d * L];
    return r;
  }
  template <int N> ZS_FUNCTION void set(int chn, std::size_t i, const small_vec<T, N> &v) const {  // tuple(...) = vec
    T *b = &(*this)(chn, i);
#pragma unroll
    for (int d = 0; d < N; ++d) b[d * L] = v.v[d];
  }
  ZS_FUNCTION std::size_t size() const { return _n; }
  ZS_FUNCTION int numChannels() const { return _C; }
};
template <class T, int L> struct TileVector {
  static_assert((L & (L - 1)) == 0, "lane width must be a power of two");
  TileVector(const std::vector<PropertyTag> &tags, std::size_t n, memsrc_e mre = memsrc_e::device) : _tags(tags), _size(n), _mre(mre) {
    for (auto &t : _tags) { _offsets.push_back(_C); _C += t.numChannels; }
    _buf = Vector<T>(tiles() * L * (std::size_t)_C, mre);
  }
  std::size_t size() const { return _size; }
  std::size_t tiles() const { return (_size + L - 1) / L; }
  int numChannels() const { return _C; }
  int getPropertyOffset(const std::string &name) const {  // -1 if absent (TileVector.hpp:528-535)
    for (std::size_t i = 0; i < _tags.size(); ++i) if (_tags[i].name == name) return _offsets[i];
    return -1;
  }
  T *data() { return _buf.data(); }
  const T *data() const { return _buf.data(); }
  void reset(int ch) { _buf.reset(ch); }
  bool hasProperty(const std::string &name) const { return getPropertyOffset(name) >= 0; }
  int getPropertySize(const std::string &name) const {
    for (auto &t : _tags) if (t.name == name) return t.numChannels;
    return 0;
  }
  // begin / end of one channel of a property (TileVector.hpp:285-322)
  auto begin(const std::string &prop, int d = 0);
  auto end(const std::string &prop, int d = 0);
  // maintenance ops that take a policy (TileVector.hpp:583-640); defined after RocmExecutionPolicy below
  template <class Pol> void append_channels(const Pol &pol, const std::vector<PropertyTag> &tags);
  template <class Pol> void reset(const Pol &pol, T val);
  // reorderTiles(pol, map[, wrapv<Scatter>]) (TileVector.hpp:641-691): whole tiles move; gather: tile i := old tile map[i], scatter:
  // tile map[i] := old tile i.  `map`: a zs::Vector of integers or any iterator_range over integers, one entry per tile
  template <class Pol, class MapRange, bool Scatter = false> void reorderTiles(const Pol &pol, MapRange &&map, wrapv<Scatter> = {});
  std::size_t numTiles() const { return tiles(); }
  TileVector clone(memsrc_e mre) const {
    TileVector r(_tags, _size, mre);
    if (_buf.size()) (void)hipMemcpy(r._buf.data(), _buf.data(), _buf.size() * sizeof(T), hipMemcpyDefault);
    return r;
  }
  aosoa_iterator_float_1 port(int chn, unsigned idx = 0) {  // py_interop/GenericIterator.hpp:76-82 (float instantiation)
    static_assert(sizeof(T) == 4, "");
    int bits = 0;
    while ((1 << bits) < L) ++bits;
    return {(float *)_buf.data() + (std::size_t)chn * L, idx, (unsigned)bits, (unsigned)(L - 1), (unsigned)_C};
  }
  std::vector<PropertyTag> _tags;
  std::vector<int> _offsets;
  int _C = 0;
  std::size_t _size;
  memsrc_e _mre;
  Vector<T> _buf;
};

template <class T, int L> auto TileVector<T, L>::begin(const std::string &prop, int d) {
  return TileVectorIterator<T, L>{_buf.data() + (std::size_t)(getPropertyOffset(prop) + d) * L, 0, _C};
}
template <class T, int L> auto TileVector<T, L>::end(const std::string &prop, int d) {
  return TileVectorIterator<T, L>{_buf.data() + (std::size_t)(getPropertyOffset(prop) + d) * L, (long long)_size, _C};
}
// range(tv, "prop"): the first channel of a property as a random-access range (TileVector.hpp:299-322 + ZpcIterator.hpp range())
template <class T, int L> iterator_range<TileVectorIterator<T, L>> range(TileVector<T, L> &tv, const std::string &prop) {
  if (tv.getPropertyOffset(prop) < 0) throw std::runtime_error("range(TileVector, \"" + prop + "\"): no such property");
  return {tv.begin(prop), tv.end(prop)};
}
// TileVectorNamedView (container/TileVector.hpp:1150-1540): tv("name", d, i) / tv("name", i) / tv.pack(dim_c<N>, "name", i); the
// property table travels by value (up to 16 properties, names up to 23 characters) and is searched linearly, as in the reference
template <class T, int L> struct TileVectorNamedView : TileVectorView<T, L> {
  static constexpr int max_props = 16, max_name = 24;
  char _names[max_props][max_name];
  int _offs[max_props], _sizes[max_props], _np = 0;
  ZS_FUNCTION int propertyOffset(const char *name) const {
    for (int k = 0; k < _np; ++k) {
      int c = 0;
      while (c < max_name && _names[k][c] == name[c] && name[c]) ++c;
      if (c < max_name && _names[k][c] == name[c]) return _offs[k];
    }
    return -1;
  }
  ZS_FUNCTION bool hasProperty(const char *name) const { return propertyOffset(name) >= 0; }
  using TileVectorView<T, L>::operator();
  using TileVectorView<T, L>::pack;
  using TileVectorView<T, L>::set;
  ZS_FUNCTION T &operator()(const char *name, int d, std::size_t i) const { return (*this)(propertyOffset(name) + d, i); }
  ZS_FUNCTION T &operator()(const char *name, std::size_t i) const { return (*this)(propertyOffset(name), i); }
  template <int N> ZS_FUNCTION small_vec<T, N> pack(dim_t<N> t, const char *name, std::size_t i) const { return this->pack(t, propertyOffset(name), i); }
  template <int N> ZS_FUNCTION void set(const char *name, std::size_t i, const small_vec<T, N> &v) const { this->set(propertyOffset(name), i, v); }
};
// zs::bht<int, dim, int, B> (container/Bht.hpp), dim 1-4, B 16|32 -- owning handle over the C ABI, device view = zsr::BhtDev
template <int dim, int B> struct bht_traits;
#define ZS_ROCM_BHT_TRAITS(D, B)                                                                          \
  template <> struct bht_traits<D, B> {                                                                  \
    using handle = zs_rocm_bht_##D;                                                                      \
    static handle *create(std::size_t n) { return container__bht_int_##D##_int_##B(nullptr, n); }       \
    static void destroy(handle *h) { del_container__bht_int_##D##_int_##B(h); }                          \
    static std::size_t size(const handle *h) { return container_size__bht_int_##D##_int_##B(h); }        \
    static void reset(handle *h, bool c) { reset_container__bht_int_##D##_int_##B(h, c); }               \
    static zs_rocm_bht_view_lite *view(handle *h) { return pyview__bht_int_##D##_int_##B(h); }           \
    static void delview(zs_rocm_bht_view_lite *v) { del_pyview__bht_int_##D##_int_##B(v); }              \
    static void resize(zs_rocm_policy *p, handle *h, std::size_t n) { resize_container__rocm_bht_int_##D##_int_##B(p, h, n); } \
  };
ZS_ROCM_BHT_TRAITS(1, 16)
ZS_ROCM_BHT_TRAITS(2, 16)
ZS_ROCM_BHT_TRAITS(3, 16)
ZS_ROCM_BHT_TRAITS(4, 16)
ZS_ROCM_BHT_TRAITS(1, 32)
ZS_ROCM_BHT_TRAITS(2, 32)
ZS_ROCM_BHT_TRAITS(3, 32)
ZS_ROCM_BHT_TRAITS(4, 32)
#undef ZS_ROCM_BHT_TRAITS

template <int dim> struct BHTView {  // BHTView (Bht.hpp:403-1072): insert / query inside kernels
  zsr::BhtDev t;
  static constexpr int sentinel_v = -1;
  static constexpr int failure_token_v = zsr::BHT_FAIL;
  // insert(key[, index, enqueue]) (Bht.hpp:490-542): index == sentinel_v takes the next dense index
  __device__ __forceinline__ int insert(const small_vec<int, dim> &key, int index = -1, bool enqueue = true) const {
    return zsr::bht_insert<dim>(t, key.v, index, enqueue);
  }
  __device__ __forceinline__ int query(const small_vec<int, dim> &key) const { return zsr::bht_query<dim>(t, key.v); }
  __device__ __forceinline__ int entry(const small_vec<int, dim> &key) const { return zsr::bht_query<dim, true>(t, key.v); }  // slot (:700-731)
  __device__ __forceinline__ int size() const { return *t.cnt; }
  // tile_insert / tile_query (Bht.hpp:547-608, 703-736): every lane of the tile carries the same key and gets the same answer; the
  // tile's lanes examine one bucket together (lane r = slot r, ballots for match / first empty slot, rank 0 claims: bht_device.hpp).
  // Tile = a cooperative-groups thread_block_tile (thread_rank(), size(), ballot(), any(), shfl()).
  template <class Tile> __device__ __forceinline__ int tile_insert(Tile &tile, const small_vec<int, dim> &key, int index = -1, bool enqueue = true) const {
    return zsr::bht_tile_insert<dim>(t, key.v, tile, index, enqueue);
  }
  template <class Tile> __device__ __forceinline__ int tile_query(Tile &tile, const small_vec<int, dim> &key) const {
    return zsr::bht_tile_query<dim>(t, key.v, tile);
  }
  int *_activeKeys() const { return t.activeKeys; }
};
// template head of the reference (container/Bht.hpp:16-18: Tn, dim, Index, B = 32, allocator); the C ABI instantiates Tn = Index = int,
// dim 1-4, B 16 | 32 (py_interop/BhtInstantiations.cpp)
template <class Tn_, int dim_, class Index = int, int B = 32> struct bht {
  static_assert(std::is_same_v<Tn_, int> && std::is_same_v<Index, int>, "the rocm backend instantiates bht<int, dim, int, B>");
  static constexpr int dim = dim_;
  static constexpr int bucket_size = B;
  using traits = bht_traits<dim, B>;
  explicit bht(std::size_t n) : _h(traits::create(n)) {}
  template <class Alloc> bht(const Alloc &, std::size_t n) : _h(traits::create(n)) {}  // bht{allocator, numExpectedEntries} (Bht.hpp:160-171)
  ~bht() { traits::destroy(_h); }
  bht(const bht &) = delete;
  std::size_t size() const { return traits::size(_h); }
  void reset(bool clearCnt = true) { traits::reset(_h, clearCnt); }
  void resize(const struct RocmExecutionPolicy &pol, std::size_t newCapacity);  // Bht.hpp:320-340
  BHTView<dim> view() {
    zs_rocm_bht_view_lite *v = traits::view(_h);
    BHTView<dim> r;
    r.t.keys = (int *)v->keys; r.t.indices = v->indices; r.t.status = v->status; r.t.activeKeys = (int *)v->activeKeys;
    r.t.cnt = v->cnt; r.t.success = v->success; r.t.tableSize = (unsigned)v->tableSize;
    r.t.bucket = (unsigned)B;
    r.t.numBuckets = v->numBuckets;
    r.t.hf[0] = v->hf0x; r.t.hf[1] = v->hf0y; r.t.hf[2] = v->hf1x; r.t.hf[3] = v->hf1y; r.t.hf[4] = v->hf2x; r.t.hf[5] = v->hf2y;
    traits::delview(v);
    return r;
  }
  typename traits::handle *_h;
};

// zs::HashTable<i32, dim, int> (container/HashTable.hpp): hash_combine hash, linear probing (stride 127)
template <int dim> struct HashTableView {  // HashTableView (HashTable.hpp:313-592)
  zsr::HtDev t;
  static constexpr int sentinel_v = -1;
  __device__ __forceinline__ int insert(const small_vec<int, dim> &key) const { return zsr::ht_insert<dim>(t, key.v); }           // :353-374
  __device__ __forceinline__ bool insert(const small_vec<int, dim> &key, int id) const { return zsr::ht_insert_id<dim>(t, key.v, id); }  // :405-421
  __device__ __forceinline__ int query(const small_vec<int, dim> &key) const { return zsr::ht_query<dim>(t, key.v); }             // :445-457
  __device__ __forceinline__ int entry(const small_vec<int, dim> &key) const { return zsr::ht_query<dim, true>(t, key.v); }       // :458-470
  __device__ __forceinline__ int size() const { return *t.cnt; }
  int *_activeKeys() const { return t.activeKeys; }
};
template <int dim> struct HashTable {
  explicit HashTable(std::size_t numExpectedEntries, memsrc_e mre = memsrc_e::device, ProcID devid = 0)
      : _h(zs_rocm_hashtable_create(dim, numExpectedEntries, (int)mre, devid)) {}
  ~HashTable() { zs_rocm_hashtable_destroy(_h); }
  HashTable(const HashTable &) = delete;
  int size() const { return zs_rocm_hashtable_size(_h); }
  std::size_t tableSize() const { return zs_rocm_hashtable_table_size(_h); }
  template <class Pol> void reset(const Pol &pol, bool clearCnt) { zs_rocm_hashtable_reset(pol.handle(), _h, clearCnt); }
  template <class Pol> void resize(const Pol &pol, std::size_t n) { zs_rocm_hashtable_resize(pol.handle(), _h, n); }
  template <class Pol> void preserve(const Pol &pol, std::size_t n) { zs_rocm_hashtable_preserve(pol.handle(), _h, n); }
  HashTableView<dim> view() const {
    zs_rocm_hashtable_view v;
    zs_rocm_hashtable_get_view(_h, &v);
    HashTableView<dim> r;
    r.t.keys = v.keys; r.t.indices = v.indices; r.t.status = v.status; r.t.activeKeys = v.activeKeys; r.t.cnt = v.cnt;
    r.t.tableSize = v.tableSize;
    return r;
  }
  zs_rocm_hashtable *_h;
};

// zs::SparseGrid<dim, T, Side>, SparseGridView, GridArena<view, kernel, deriv_order> (geometry/SparseGrid.hpp, math/curve/InterpolationKernel.hpp)
}  // namespace zs
#include "sparse_grid.hpp"
namespace zs {

// zs::LBvh<3, int, f32> (container/Bvh.hpp:86-1248) over a zs::Vector<AABBBox<3, f32>> of primitive boxes
using AABBBox3f = zsr::AABB3;  // {lo[3], hi[3]} == AABBBox<3, f32> {_min, _max}
using LBvhView = zsr::LBvhDev; // iter_neighbors(bv, f) / self_iter_neighbors(leaf, f) / find_nearest(p, f, cap) inside kernels
struct LBvh {
  LBvh() : _h(zs_rocm_lbvh_create()) {}
  ~LBvh() { zs_rocm_lbvh_destroy(_h); }
  LBvh(const LBvh &) = delete;
  template <class Pol> void build(const Pol &pol, const Vector<AABBBox3f> &primBvs, bool refit = true) {  // :810-1082
    zs_rocm_lbvh_build(pol.handle(), _h, (const float *)primBvs.data(), primBvs.size(), refit);
  }
  template <class Pol> void refit(const Pol &pol, const Vector<AABBBox3f> &primBvs) {  // :1219-1248
    if (zs_rocm_lbvh_refit(pol.handle(), _h, (const float *)primBvs.data(), primBvs.size()) != 0)
      throw std::runtime_error("bvh topology changes, require rebuild!");
  }
  std::size_t getNumLeaves() const { return zs_rocm_lbvh_num_leaves(_h); }
  std::size_t getNumNodes() const { return zs_rocm_lbvh_num_nodes(_h); }
  LBvhView view() const {
    zs_rocm_lbvh_view v;
    zs_rocm_lbvh_get_view(_h, &v);
    LBvhView r;
    r.orderedBvs = (const zsr::AABB3 *)v.orderedBvs; r.parents = v.parents; r.levels = v.levels; r.leafInds = v.leafInds;
    r.auxIndices = v.auxIndices; r.numNodes = v.numNodes;
    return r;
  }
  zs_rocm_lbvh *_h;
};

// view<space>(container) / proxy<space>(container)  (container/Vector.hpp:455-615, TileVector.hpp:693-1540, Bht.hpp:403)
template <execspace_e space, class T> VectorView<T> view(Vector<T> &v) {
  static_assert(space == execspace_e::rocm, "this header provides the rocm space only");
  return {v.data(), v.size()};
}
template <execspace_e space, class T, int L> TileVectorView<T, L> view(TileVector<T, L> &v) { return {v.data(), v.size(), v.numChannels()}; }
// view<space>({"a", "b"}, tv) / proxy<space>({...}, tv): the named view over the listed properties (TileVector.hpp:1513-1540)
template <execspace_e space, class T, int L> TileVectorNamedView<T, L> view(std::initializer_list<const char *> names, TileVector<T, L> &v) {
  TileVectorNamedView<T, L> r{};
  static_cast<TileVectorView<T, L> &>(r) = view<space>(v);
  for (const char *nm : names) {
    if (r._np == r.max_props) throw std::runtime_error("named TileVector view: more than 16 properties");
    const int off = v.getPropertyOffset(nm);
    if (off < 0) throw std::runtime_error(std::string("named TileVector view: no property \"") + nm + "\"");
    int c = 0;
    for (; nm[c] && c < r.max_name - 1; ++c) r._names[r._np][c] = nm[c];
    r._names[r._np][c] = 0;
    r._offs[r._np] = off;
    r._sizes[r._np] = v.getPropertySize(nm);
    ++r._np;
  }
  return r;
}
template <execspace_e space, class Tn, int dim, class Ix, int B> BHTView<dim> view(bht<Tn, dim, Ix, B> &t) { return t.view(); }
template <execspace_e space, int dim> HashTableView<dim> view(HashTable<dim> &t) { return t.view(); }
template <execspace_e space, int dim, class T, int Side> SparseGridView<dim, T, Side> view(SparseGrid<dim, T, Side> &g) { return g.view(); }
template <execspace_e space> LBvhView view(const LBvh &b) { return b.view(); }
template <execspace_e space, class C> auto proxy(C &c) { return view<space>(c); }
template <execspace_e space, class T, int L> auto proxy(std::initializer_list<const char *> l, TileVector<T, L> &v) { return view<space>(l, v); }
}  // namespace zs
#include "structures.hpp"  // zs::Particles / ParticlesView, zs::Grids / GridsView
namespace zs {

// ------------------------------------------------------------------------------------ launch kernels
namespace detail {
// functor signature deduction (the role of detail::deduce_fts, cuda/execution/ExecutionPolicy.cuh:40-155): arity and
  // the type of the first parameter (a pointer => the dynamic-LDS base is passed first)
  template <class F> struct fn_traits : fn_traits<decltype(&F::operator())> {};
  template <class C, class R, class... A> struct fn_traits<R (C::*)(A...) const> {
    static constexpr int arity = sizeof...(A);
    using first = std::tuple_element_t<0, std::tuple<A...>>;
  };
  template <class C, class R, class... A> struct fn_traits<R (C::*)(A...)> : fn_traits<R (C::*)(A...) const> {};
  extern __shared__ __attribute__((aligned(16))) char zs_rocm_dyn_shmem[];

  // thread_launch (ExecutionPolicy.cuh:212-227): f(i) or f(shmem*, i)
  template <class F> __global__ void thread_launch(long long b, long long n, F f) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    if constexpr (fn_traits<F>::arity == 1) f(b + id);
    else f((typename fn_traits<F>::first)zs_rocm_dyn_shmem, b + id);
  }
  // range_launch / range_launch_with_params (:245-322): one thread per element of a zip range; the functor is called with the
  // dereferenced members, optionally preceded by the element index and / or the dynamic-LDS base, optionally followed by the
  // parameter tuple -- the arity dispatch of detail::deduce_fts (:40-155):
  //   arity == members            f(refs...)                 members + 1, first integral   f(i, refs...)
  //   members + 1, first pointer  f(shmem*, refs...)         members + 2                   f(shmem*, i, refs...)
  template <class F, class Ref, std::size_t... Is, class... Pre> __device__ __forceinline__ void range_call(F &f, Ref &&ref, std::index_sequence<Is...>, Pre... pre) {
    f(pre..., std::get<Is>(ref)...);
  }
  template <class F, class Ref, class Params, std::size_t... Is, class... Pre>
  __device__ __forceinline__ void range_call_p(F &f, Ref &&ref, const Params &params, std::index_sequence<Is...>, Pre... pre) {
    f(pre..., std::get<Is>(ref)..., params);
  }
  template <class F, class ZipIter> __global__ void range_launch(long long n, F f, ZipIter iter) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    using TR = fn_traits<F>;
    constexpr int members = (int)std::tuple_size_v<typename ZipIter::reference>;
    constexpr auto seq = std::make_index_sequence<members>{};
    static_assert(TR::arity >= members && TR::arity <= members + 2, "range_launch: functor arity does not match the zipped ranges");
    if constexpr (TR::arity == members) range_call(f, iter[id], seq);
    else if constexpr (TR::arity == members + 1) {
      using first = std::remove_reference_t<typename TR::first>;
      static_assert(std::is_integral_v<first> || std::is_pointer_v<first>, "range_launch: the extra leading argument is an index or a shmem pointer");
      if constexpr (std::is_integral_v<first>) range_call(f, iter[id], seq, (first)id);
      else range_call(f, iter[id], seq, (first)zs_rocm_dyn_shmem);
    } else {
      using first = std::remove_reference_t<typename TR::first>;
      static_assert(std::is_pointer_v<first>, "range_launch: with two extra leading arguments the first is a shmem pointer");
      range_call(f, iter[id], seq, (first)zs_rocm_dyn_shmem, id);
    }
  }
  template <class F, class ZipIter, class Params> __global__ void range_launch_with_params(long long n, F f, ZipIter iter, Params params) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    using TR = fn_traits<F>;
    constexpr int members = (int)std::tuple_size_v<typename ZipIter::reference>;
    constexpr auto seq = std::make_index_sequence<members>{};
    static_assert(TR::arity >= members + 1 && TR::arity <= members + 3, "range_launch_with_params: functor arity does not match");
    if constexpr (TR::arity == members + 1) range_call_p(f, iter[id], params, seq);
    else if constexpr (TR::arity == members + 2) {
      using first = std::remove_reference_t<typename TR::first>;
      if constexpr (std::is_integral_v<first>) range_call_p(f, iter[id], params, seq, (first)id);
      else range_call_p(f, iter[id], params, seq, (first)zs_rocm_dyn_shmem);
    } else {
      using first = std::remove_reference_t<typename TR::first>;
      range_call_p(f, iter[id], params, seq, (first)zs_rocm_dyn_shmem, id);
    }
  }
  // dst[i] = src[i] between any two random-access iterators (staging of non-contiguous ranges for the primitives)
  template <class Src, class Dst> __global__ void iter_copy_launch(long long n, Src src, Dst dst) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) dst[id] = src[id];
  }
  // block_thread_launch (:228-243): f(block, thread) or f(shmem*, block, thread)
  template <class F> __global__ void block_thread_launch(F f) {
    if constexpr (fn_traits<F>::arity == 2) f((int)blockIdx.x, (int)threadIdx.x);
    else f((typename fn_traits<F>::first)zs_rocm_dyn_shmem, (int)blockIdx.x, (int)threadIdx.x);
  }
  // block_tile_lane_launch (:324-343): f(block, tileNo, laneInTile); tiles are sub-wavefront groups of tileSize lanes
  template <class F> __global__ void block_tile_lane_launch(int tileSize, F f) {
    const int tile = (int)threadIdx.x / tileSize, lane = (int)threadIdx.x % tileSize;
    if constexpr (fn_traits<F>::arity == 3) f((int)blockIdx.x, tile, lane);
    else f((typename fn_traits<F>::first)zs_rocm_dyn_shmem, (int)blockIdx.x, tile, lane);
  }
}  // namespace detail

// ------------------------------------------------------------------------------------ RocmExecutionPolicy
struct RocmExecutionPolicy {
  using exec_tag = rocm_exec_tag;
  RocmExecutionPolicy() : _h(policy__device()) {}
  RocmExecutionPolicy(const RocmExecutionPolicy &o) : RocmExecutionPolicy() {  // policies are value types
    sync(o._sync).profile(o._profile).device(o._dev).stream(o._stream).shmem(o._shmem).block(o._block);
  }
  ~RocmExecutionPolicy() { del_policy__device(_h); }
  // fluent setters (execution/ExecutionPolicy.hpp:110-126, cuda/execution/ExecutionPolicy.cuh:362-385)
  RocmExecutionPolicy &sync(bool s) { _sync = s; zs_rocm_policy_sync(_h, s); return *this; }
  RocmExecutionPolicy &profile(bool p) { _profile = p; zs_rocm_policy_profile(_h, p); return *this; }
  RocmExecutionPolicy &device(ProcID d) { _dev = d; zs_rocm_policy_device(_h, d); return *this; }
  RocmExecutionPolicy &stream(StreamID s) { _stream = s; zs_rocm_policy_stream(_h, s); return *this; }
  RocmExecutionPolicy &listen(ProcID p, StreamID s) { zs_rocm_policy_listen(_h, p, s); return *this; }
  RocmExecutionPolicy &shmem(std::size_t b) { _shmem = b; zs_rocm_policy_shmem(_h, b); return *this; }
  RocmExecutionPolicy &block(int tpb) { _block = tpb; zs_rocm_policy_block(_h, tpb); return *this; }
  bool shouldSync() const { return _sync; }
  void *getStream() const { return zs_rocm_policy_get_stream(_h); }
  void syncCtx() const { zs_rocm_policy_sync_ctx(_h); }
  ProcID getProcid() const { return _dev; }  // -1 = the current device (cuda/execution/ExecutionPolicy.cuh:905)
  zs_rocm_policy *handle() const { return _h; }

  // pol(range(n), f)
  template <class F> void operator()(index_range r, F &&f) const {
    const long long n = r.e - r.b;
    if (n <= 0) return;
    const int bs = _block > 0 ? _block : 256;  // wave64-aware default (the reference deduces by occupancy, Cuda.cu:376-461)
    hipLaunchKernelGGL((detail::thread_launch<std::decay_t<F>>), dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), _shmem,
                       (hipStream_t)getStream(), r.b, n, f);
    finish();
  }
  // pol(range, f) / pol(range, params, f) for zip / enumerate / container ranges (ExecutionPolicy.cuh:458-535): a range that is not a
  // zip range is wrapped in one, so pol(range(tv, "b"), [](int &v) {...}) and pol(enumerate(vals), [](auto i, int &v) {...}) both work
  template <class It, class F> void operator()(const iterator_range<It> &r, F &&f) const {
    if constexpr (is_zip_iterator<It>::value) launch_zip(r.begin(), (long long)(r.end() - r.begin()), std::forward<F>(f));
    else launch_zip(zip_iterator<It>{{r.begin()}}, (long long)(r.end() - r.begin()), std::forward<F>(f));
  }
  // params: zs::tuple / zs::make_tuple(...) as in the reference (ExecutionPolicy.cuh:281-322, 458-535); std::tuple is accepted too
  template <class It, class... Ps, class F> void operator()(const iterator_range<It> &r, const tuple<Ps...> &params, F &&f) const {
    launch_params(r, params, std::forward<F>(f));
  }
  template <class It, class... Ps, class F> void operator()(const iterator_range<It> &r, const std::tuple<Ps...> &params, F &&f) const {
    launch_params(r, params, std::forward<F>(f));
  }
  template <class T, class F> void operator()(Vector<T> &v, F &&f) const { (*this)(range(v), std::forward<F>(f)); }
  // pol(Collapse{...}, f)
  template <class F> void operator()(Collapse c, F &&f) const {
    using FT = std::decay_t<F>;
    using TR = detail::fn_traits<FT>;
    constexpr bool shm = std::is_pointer_v<typename TR::first>;
    constexpr int nidx = TR::arity - (shm ? 1 : 0);
    if (c.n[0] <= 0) return;
    if (c.dim != nidx) throw std::runtime_error("RocmExecutionPolicy: functor arity does not match the Collapse dimension");
    if constexpr (nidx == 1) {
      (*this)(range(c.n[0]), std::forward<F>(f));
    } else if constexpr (nidx == 2) {
      hipLaunchKernelGGL((detail::block_thread_launch<FT>), dim3((unsigned)c.n[0]), dim3((unsigned)c.n[1]), _shmem,
                         (hipStream_t)getStream(), f);
      finish();
    } else {
      hipLaunchKernelGGL((detail::block_tile_lane_launch<FT>), dim3((unsigned)c.n[0]), dim3((unsigned)(c.n[1] * c.n[2])), _shmem,
                         (hipStream_t)getStream(), (int)c.n[2], f);
      finish();
    }
  }
  // member primitives on contiguous device ranges (ExecutionPolicy.cuh:560-881)
  template <class T, class Op = plus<T>> void reduce(const T *first, const T *last, T *out, T init = T{}, Op = {}) const {
    call_reduce(first, (std::size_t)(last - first), out, init, detail::op_code<Op>::value);
  }
  template <class T, class Op = plus<T>> void exclusive_scan(const T *first, const T *last, T *out, T init = T{}, Op = {}) const {
    call_scan(first, (std::size_t)(last - first), out, init, detail::op_code<Op>::value, 1);
  }
  template <class T, class Op = plus<T>> void inclusive_scan(const T *first, const T *last, T *out, Op = {}) const {
    call_scan(first, (std::size_t)(last - first), out, T{}, detail::op_code<Op>::value, 0);
  }
  template <class K> void radix_sort(const K *first, const K *last, K *out, int sbit = 0, int ebit = sizeof(K) * 8) const {
    call_sort(first, (const int *)nullptr, out, (int *)nullptr, (std::size_t)(last - first), sbit, ebit);
  }
  template <class K> void radix_sort_pair(const K *kin, const int *vin, K *kout, int *vout, std::size_t n, int sbit = 0,
                                          int ebit = sizeof(K) * 8) const {
    call_sort(kin, vin, kout, vout, n, sbit, ebit);
  }

  // the same over ANY random-access device iterators (execution/ExecutionPolicy.hpp:765-781 take iterators: Vector / TileVector-channel
  // begin(), strided views): non-contiguous ranges are staged through this policy's stream-ordered temporaries like the scans below
  template <class InIt, class OutIt, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
  void radix_sort(InIt first, InIt last, OutIt d_first, int sbit = 0,
                  int ebit = (int)sizeof(std::remove_cv_t<std::remove_reference_t<decltype(first[0])>>) * 8) const {
    using K = std::remove_cv_t<std::remove_reference_t<decltype(first[0])>>;
    const std::size_t n = (std::size_t)(last - first);
    if (!n) return;
    K *in = stage_in<K>(first, n), *out = (K *)zs_rocm_policy_temporary(_h, n * sizeof(K));
    radix_sort((const K *)in, (const K *)in + n, out, sbit, ebit);
    stage_out(out, d_first, n);
    zs_rocm_policy_temporary_free(_h, out);
    zs_rocm_policy_temporary_free(_h, in);
    finish();
  }
  template <class KIn, class VIn, class KOut, class VOut, class Tn,
            std::enable_if_t<!(std::is_pointer_v<KIn> && std::is_pointer_v<VIn> && std::is_pointer_v<KOut> && std::is_pointer_v<VOut>) && std::is_integral_v<Tn>, int> = 0>
  void radix_sort_pair(KIn keysIn, VIn valsIn, KOut keysOut, VOut valsOut, Tn count, int sbit = 0,
                       int ebit = (int)sizeof(std::remove_cv_t<std::remove_reference_t<decltype(keysIn[0])>>) * 8) const {
    using K = std::remove_cv_t<std::remove_reference_t<decltype(keysIn[0])>>;
    static_assert(sizeof(std::remove_reference_t<decltype(valsIn[0])>) == sizeof(int), "4-byte values (the C ABI sorts (key, int) pairs)");
    const std::size_t n = (std::size_t)count;
    if (!n) return;
    K *kin = stage_in<K>(keysIn, n), *kout = (K *)zs_rocm_policy_temporary(_h, n * sizeof(K));
    int *vin = stage_in<int>(valsIn, n), *vout = (int *)zs_rocm_policy_temporary(_h, n * sizeof(int));
    radix_sort_pair((const K *)kin, (const int *)vin, kout, vout, n, sbit, ebit);
    stage_out(kout, keysOut, n);
    stage_out(vout, valsOut, n);
    for (void *q : {(void *)vout, (void *)vin, (void *)kout, (void *)kin}) zs_rocm_policy_temporary_free(_h, q);
    finish();
  }

  // merge_sort / merge_sort_pair with a user comparator (cuda/execution/ExecutionPolicy.cuh:698-752): stable, in place;
  // `first`/`keys`/`vals` are random-access device iterators (raw pointers, or views with operator[])
  template <class KeyIter, class Comp = less<void>> void merge_sort(KeyIter first, KeyIter last, Comp comp = {}) const {
    using K = std::remove_cv_t<std::remove_reference_t<decltype(first[0])>>;
    const std::size_t n = (std::size_t)(last - first);
    const std::size_t b = zs_rocm_ms::scratch_bytes<K, zs_rocm_ms::NoVal, false>(n);
    void *scratch = b ? zs_rocm_policy_temporary(_h, b) : nullptr;
    zs_rocm_ms::merge_sort_run<K, zs_rocm_ms::NoVal, false>((hipStream_t)getStream(), first, (zs_rocm_ms::NoVal *)nullptr, n, comp, scratch);
    zs_rocm_policy_temporary_free(_h, scratch);  // stream-ordered: released after the passes above have run
    finish();
  }
  template <class KeyIter, class ValIter, class Comp = less<void>>
  void merge_sort_pair(KeyIter keys, ValIter vals, std::size_t n, Comp comp = {}) const {
    using K = std::remove_cv_t<std::remove_reference_t<decltype(keys[0])>>;
    using V = std::remove_cv_t<std::remove_reference_t<decltype(vals[0])>>;
    const std::size_t b = zs_rocm_ms::scratch_bytes<K, V, true>(n);
    void *scratch = b ? zs_rocm_policy_temporary(_h, b) : nullptr;
    zs_rocm_ms::merge_sort_run<K, V, true>((hipStream_t)getStream(), keys, vals, n, comp, scratch);
    zs_rocm_policy_temporary_free(_h, scratch);
    finish();
  }

  // reduce / scans over ANY random-access device iterators (TileVector channels, strided views, ...): contiguous pointers go
  // straight to the C ABI; anything else is staged through stream-ordered temporaries of this policy
  template <class InIt, class OutIt, class T, class Op, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
  void reduce(InIt first, InIt last, OutIt out, T init, Op op) const {
    const std::size_t n = (std::size_t)(last - first);
    T *in = stage_in<T>(first, n), *res = (T *)zs_rocm_policy_temporary(_h, sizeof(T));
    reduce((const T *)in, (const T *)in + n, res, init, op);
    stage_out(res, out, 1);
    zs_rocm_policy_temporary_free(_h, res);
    if (n) zs_rocm_policy_temporary_free(_h, in);
    finish();
  }
  template <class InIt, class OutIt, class T, class Op, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
  void exclusive_scan(InIt first, InIt last, OutIt out, T init, Op op) const {
    const std::size_t n = (std::size_t)(last - first);
    if (!n) return;
    T *in = stage_in<T>(first, n);
    exclusive_scan((const T *)in, (const T *)in + n, in, init, op);
    stage_out(in, out, n);
    zs_rocm_policy_temporary_free(_h, in);
    finish();
  }
  template <class InIt, class OutIt, class Op, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
  void inclusive_scan(InIt first, InIt last, OutIt out, Op op) const {
    using T = std::remove_cv_t<std::remove_reference_t<decltype(first[0])>>;
    const std::size_t n = (std::size_t)(last - first);
    if (!n) return;
    T *in = stage_in<T>(first, n);
    inclusive_scan((const T *)in, (const T *)in + n, in, op);
    stage_out(in, out, n);
    zs_rocm_policy_temporary_free(_h, in);
    finish();
  }

private:
  template <class It, class Params, class F> void launch_params(const iterator_range<It> &r, const Params &params, F &&f) const {
    const long long n = (long long)(r.end() - r.begin());
    if (n <= 0) return;
    const int bs = _block > 0 ? _block : 256;
    if constexpr (is_zip_iterator<It>::value)
      hipLaunchKernelGGL((detail::range_launch_with_params<std::decay_t<F>, It, Params>), dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), _shmem,
                         (hipStream_t)getStream(), n, f, r.begin(), params);
    else
      hipLaunchKernelGGL((detail::range_launch_with_params<std::decay_t<F>, zip_iterator<It>, Params>), dim3((unsigned)((n + bs - 1) / bs)), dim3(bs),
                         _shmem, (hipStream_t)getStream(), n, f, zip_iterator<It>{{r.begin()}}, params);
    finish();
  }
  template <class ZI, class F> void launch_zip(ZI it, long long n, F &&f) const {
    if (n <= 0) return;
    const int bs = _block > 0 ? _block : 256;
    hipLaunchKernelGGL((detail::range_launch<std::decay_t<F>, ZI>), dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), _shmem,
                       (hipStream_t)getStream(), n, f, it);
    finish();
  }
  template <class T, class It> T *stage_in(It first, std::size_t n) const {
    if (!n) return nullptr;
    T *tmp = (T *)zs_rocm_policy_temporary(_h, n * sizeof(T));
    hipLaunchKernelGGL((detail::iter_copy_launch<It, T *>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)getStream(), (long long)n,
                       first, tmp);
    return tmp;
  }
  template <class T, class It> void stage_out(T *src, It dst, std::size_t n) const {
    if (!n) return;
    hipLaunchKernelGGL((detail::iter_copy_launch<T *, It>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)getStream(), (long long)n,
                       src, dst);
  }
  void finish() const {
    if (_sync) (void)hipStreamSynchronize((hipStream_t)getStream());
  }
  void call_reduce(const int *in, std::size_t n, int *out, int init, int op) const { zs_rocm_reduce_i32(_h, in, n, out, init, op); }
  void call_reduce(const long long *in, std::size_t n, long long *out, long long init, int op) const { zs_rocm_reduce_i64(_h, (const int64_t *)in, n, (int64_t *)out, init, op); }
  void call_reduce(const float *in, std::size_t n, float *out, float init, int op) const { zs_rocm_reduce_f32(_h, in, n, out, init, op); }
  void call_reduce(const double *in, std::size_t n, double *out, double init, int op) const { zs_rocm_reduce_f64(_h, in, n, out, init, op); }
  void call_scan(const int *in, std::size_t n, int *out, int init, int op, int ex) const { zs_rocm_scan_i32(_h, in, n, out, init, op, ex); }
  void call_scan(const float *in, std::size_t n, float *out, float init, int op, int ex) const { zs_rocm_scan_f32(_h, in, n, out, init, op, ex); }
  void call_scan(const double *in, std::size_t n, double *out, double init, int op, int ex) const { zs_rocm_scan_f64(_h, in, n, out, init, op, ex); }
  void call_sort(const int *k, const int *v, int *ko, int *vo, std::size_t n, int s, int e) const { zs_rocm_radix_sort_i32(_h, k, v, ko, vo, n, s, e); }
  void call_sort(const unsigned *k, const int *v, unsigned *ko, int *vo, std::size_t n, int s, int e) const { zs_rocm_radix_sort_u32(_h, k, v, ko, vo, n, s, e); }
  void call_sort(const unsigned long long *k, const int *v, unsigned long long *ko, int *vo, std::size_t n, int s, int e) const {
    zs_rocm_radix_sort_u64(_h, (const uint64_t *)k, v, (uint64_t *)ko, vo, n, s, e);
  }
  zs_rocm_policy *_h;
  bool _sync = true, _profile = false;
  ProcID _dev = -1;
  StreamID _stream = -1;
  std::size_t _shmem = 0;
  int _block = 0;
};
inline RocmExecutionPolicy rocm_exec() { return RocmExecutionPolicy{}; }
template <class Tn, int dim, class Ix, int B> void bht<Tn, dim, Ix, B>::resize(const RocmExecutionPolicy &pol, std::size_t newCapacity) {
  traits::resize(pol.handle(), _h, newCapacity);
}

// free functions (execution/ExecutionPolicy.hpp:684-781)
template <class T, class Op = plus<T>> void reduce(const RocmExecutionPolicy &pol, const T *first, const T *last, T *out, T init = T{}, Op op = {}) {
  pol.reduce(first, last, out, init, op);
}
template <class T, class Op = plus<T>> void exclusive_scan(const RocmExecutionPolicy &pol, const T *first, const T *last, T *out, T init = T{}, Op op = {}) {
  pol.exclusive_scan(first, last, out, init, op);
}
template <class T, class Op = plus<T>> void inclusive_scan(const RocmExecutionPolicy &pol, const T *first, const T *last, T *out, Op op = {}) {
  pol.inclusive_scan(first, last, out, op);
}
template <class InIt, class OutIt, class T, class Op, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
void reduce(const RocmExecutionPolicy &pol, InIt first, InIt last, OutIt out, T init, Op op) {
  pol.reduce(first, last, out, init, op);
}
template <class InIt, class OutIt, class T, class Op, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
void exclusive_scan(const RocmExecutionPolicy &pol, InIt first, InIt last, OutIt out, T init, Op op) {
  pol.exclusive_scan(first, last, out, init, op);
}
template <class InIt, class OutIt, class Op, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
void inclusive_scan(const RocmExecutionPolicy &pol, InIt first, InIt last, OutIt out, Op op) {
  pol.inclusive_scan(first, last, out, op);
}
template <class K> void radix_sort(const RocmExecutionPolicy &pol, const K *first, const K *last, K *out, int sbit = 0, int ebit = sizeof(K) * 8) {
  pol.radix_sort(first, last, out, sbit, ebit);
}
template <class K> void radix_sort_pair(const RocmExecutionPolicy &pol, const K *kin, const int *vin, K *kout, int *vout, std::size_t n, int sbit = 0,
                                        int ebit = sizeof(K) * 8) {
  pol.radix_sort_pair(kin, vin, kout, vout, n, sbit, ebit);
}
template <class InIt, class OutIt, std::enable_if_t<!(std::is_pointer_v<InIt> && std::is_pointer_v<OutIt>), int> = 0>
void radix_sort(const RocmExecutionPolicy &pol, InIt first, InIt last, OutIt d_first, int sbit = 0,
                int ebit = (int)sizeof(std::remove_cv_t<std::remove_reference_t<decltype(first[0])>>) * 8) {
  pol.radix_sort(first, last, d_first, sbit, ebit);
}
template <class KIn, class VIn, class KOut, class VOut, class Tn,
          std::enable_if_t<!(std::is_pointer_v<KIn> && std::is_pointer_v<VIn> && std::is_pointer_v<KOut> && std::is_pointer_v<VOut>) && std::is_integral_v<Tn>, int> = 0>
void radix_sort_pair(const RocmExecutionPolicy &pol, KIn keysIn, VIn valsIn, KOut keysOut, VOut valsOut, Tn count, int sbit = 0,
                     int ebit = (int)sizeof(std::remove_cv_t<std::remove_reference_t<decltype(keysIn[0])>>) * 8) {
  pol.radix_sort_pair(keysIn, valsIn, keysOut, valsOut, count, sbit, ebit);
}

template <class KeyIter, class Comp = less<void>> void merge_sort(const RocmExecutionPolicy &pol, KeyIter first, KeyIter last, Comp comp = {}) {
  pol.merge_sort(first, last, comp);
}
template <class KeyIter, class ValIter, class Comp = less<void>>
void merge_sort_pair(const RocmExecutionPolicy &pol, KeyIter keys, ValIter vals, std::size_t n, Comp comp = {}) {
  pol.merge_sort_pair(keys, vals, n, comp);
}
// zs::sort / sort_pair (execution/ExecutionPolicy.hpp:730-748): "currently [unstable] adopts the [stable] routine" (:341)
template <class KeyIter, class Comp = less<void>> void sort(const RocmExecutionPolicy &pol, KeyIter first, KeyIter last, Comp comp = {}) {
  pol.merge_sort(first, last, comp);
}
template <class KeyIter, class ValIter, class Comp = less<void>>
void sort_pair(const RocmExecutionPolicy &pol, KeyIter keys, ValIter vals, std::size_t n, Comp comp = {}) {
  pol.merge_sort_pair(keys, vals, n, comp);
}


// zs::for_each (execution/ExecutionPolicy.hpp:684-690), zs::par_exec(tag) (:85-97)
template <class Range, class F> void for_each(const RocmExecutionPolicy &pol, Range &&r, F &&f) { pol(std::forward<Range>(r), std::forward<F>(f)); }
inline RocmExecutionPolicy par_exec(rocm_exec_tag) { return rocm_exec(); }
// valid_memspace_for_execution (resource/Resource.h:163-166, rocm branch): device and unified memory
inline bool valid_memspace_for_execution(const RocmExecutionPolicy &, memsrc_e mre) { return mre == memsrc_e::device || mre == memsrc_e::um; }

inline TemporaryMemorySource get_temporary_memory_source(const RocmExecutionPolicy &pol) { return {pol.handle()}; }

// zs::make_monoid(op).identity() (ZpcFunctional.hpp): the identities the primitives use as `init`
template <class Op> struct monoid;
template <class T> struct monoid<plus<T>> : plus<T> { static constexpr T identity() { return T(0); } };
template <class T> struct monoid<multiplies<T>> : multiplies<T> { static constexpr T identity() { return T(1); } };
template <class T> struct monoid<getmin<T>> : getmin<T> { static constexpr T identity() { return std::numeric_limits<T>::max(); } };
template <class T> struct monoid<getmax<T>> : getmax<T> { static constexpr T identity() { return std::numeric_limits<T>::lowest(); } };
template <class Op> constexpr monoid<Op> make_monoid(Op) { return {}; }

// TileVector::append_channels(pol, tags) (TileVector.hpp:583-623): new channels are zero-filled, an existing tag must keep its width
template <class T, int L> template <class Pol> void TileVector<T, L>::append_channels(const Pol &pol, const std::vector<PropertyTag> &tags) {
  std::vector<PropertyTag> all = _tags;
  for (auto &t : tags) {
    bool found = false;
    for (auto &o : _tags)
      if (o.name == t.name) {
        if (o.numChannels != t.numChannels) throw std::runtime_error("append_channels: property \"" + t.name + "\" changes its width");
        found = true;
      }
    if (!found) all.push_back(t);
  }
  if (all.size() == _tags.size()) return;
  TileVector grown(all, _size, _mre);
  grown._buf.reset(0);
  const int Co = _C, Cn = grown._C;
  const T *src = _buf.data();
  T *dst = grown._buf.data();
  const long long rows = (long long)tiles() * Co * L;  // one element of one channel row per thread, whole tiles
  pol(range(rows), [=] ZS_LAMBDA(long long e) {
    const long long tile = e / ((long long)Co * L), r = e % ((long long)Co * L);
    dst[tile * Cn * L + r] = src[e];  // the old channels keep their offsets, so a tile's first Co rows copy straight over
  });
  *this = std::move(grown);
}
namespace detail {
  template <class T> ZS_FUNCTION const T *map_begin(const Vector<T> &v) { return v.data(); }
  template <class T> ZS_FUNCTION T *map_begin(Vector<T> &v) { return v.data(); }
  template <class It> ZS_FUNCTION It map_begin(const iterator_range<It> &r) { return r.begin(); }
}  // namespace detail
template <class T, int L> template <class Pol, class MapRange, bool Scatter>
void TileVector<T, L>::reorderTiles(const Pol &pol, MapRange &&map, wrapv<Scatter>) {
  if (range_size(map) != tiles()) throw std::runtime_error("index mapping range size mismatch");
  TileVector ordered(_tags, _size, _mre);
  const T *src = _buf.data();
  T *dst = ordered._buf.data();
  const int C = _C;
  auto m = detail::map_begin(map);
  static_assert(std::is_integral_v<std::decay_t<decltype(m[0])>>, "the mapping range must yield integers");
  // one thread per (tile, channel row, lane): a wave moves whole 4 L-byte rows
  pol(range((long long)tiles() * C * L), [=] ZS_LAMBDA(long long e) {
    const long long tile = e / ((long long)C * L), r = e % ((long long)C * L);
    const long long other = (long long)m[tile];
    if constexpr (Scatter) dst[other * C * L + r] = src[e];
    else dst[e] = src[other * C * L + r];
  });
  *this = std::move(ordered);
}
// TileVector::reset(pol, val) (TileVector.hpp:624-640)
template <class T, int L> template <class Pol> void TileVector<T, L>::reset(const Pol &pol, T val) {
  T *p = _buf.data();
  pol(range((long long)_buf.size()), [=] ZS_LAMBDA(long long e) { p[e] = val; });
}

}  // namespace zs

// structured bindings for zs::tuple: auto [a, b] = zs::make_tuple(...)
namespace std {
template <class... Ts> struct tuple_size<zs::tuple<Ts...>> : integral_constant<size_t, sizeof...(Ts)> {};
template <size_t I, class... Ts> struct tuple_element<I, zs::tuple<Ts...>> { using type = zs::tuple_element_t<I, zs::tuple<Ts...>>; };
}  // namespace std
