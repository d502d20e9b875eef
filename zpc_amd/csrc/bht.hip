// bht.hip -- zs::bht<int, dim, int, B> (dim 1-4, B 16|32) container handles (py_interop/BhtInstantiations.cpp:6-128,
// py_interop/cuda/BhtUtility.cpp:7-26) and the bulk insert / query / reorder kernels that replace the
// `pol(range(n), [tb = view<space>(tab)](i){ tb.insert(key_i); })` idiom of the C++ face.
#include <random>

#include "bht.hpp"

namespace zsr {

void exclusive_scan_u32(Launch &L, const unsigned *in, size_t n, unsigned *out);
void radix_sort_pair_u32(Launch &L, const unsigned *kin, const int *vin, unsigned *kout, int *vout, size_t n, int sbit, int ebit);
void radix_sort_pair_u64(Launch &L, const unsigned long long *kin, const int *vin, unsigned long long *kout, int *vout, size_t n, int sbit,
                         int ebit);

static size_t next_2pow(size_t n) {  // math/bit/Bits.h:177-184
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
static size_t evaluate_table_size(size_t entryCnt, int B) {  // Bht.hpp:154-158
  if (entryCnt == 0) return 0;
  size_t n = next_2pow(entryCnt) * 2;
  return n + ((size_t)B - n % (size_t)B);
}

static void *bht_alloc(const BhtHost &t, size_t bytes) {
  void *p = nullptr;
  if (bytes == 0) bytes = 16;
  DeviceGuard guard(t.devid);  // the container lives on ITS device, whatever the calling thread is on
  if (t.memsrc == 2) ZSR_CHECK(hipMallocManaged(&p, bytes));
  else ZSR_CHECK(hipMalloc(&p, bytes));
  return p;
}

static void bht_reset_table(BhtHost &t, hipStream_t s) {  // Table::reset, Bht.hpp:107-112
  const int ks = t.dim == 1 ? 1 : (t.dim == 2 ? 2 : 4);
  if (t.tableSize) {
    ZSR_CHECK(hipMemsetAsync(t.keys, 0x3f, t.tableSize * ks * sizeof(int), s));
    ZSR_CHECK(hipMemsetAsync(t.status, 0xff, t.tableSize * sizeof(int), s));
  }
}

static void bht_create(BhtHost &t, int dim, int B, int memsrc, int8_t devid, size_t n) {
  t.dim = dim;
  t.bucket = B;
  t.memsrc = memsrc == 0 ? 1 : memsrc;  // a host-resident bht cannot be used by a device policy
  t.devid = devid;
  t.tableSize = evaluate_table_size(n, B);
  const int ks = dim == 1 ? 1 : (dim == 2 ? 2 : 4);
  t.keys = (int *)bht_alloc(t, t.tableSize * ks * sizeof(int));
  t.indices = (int *)bht_alloc(t, t.tableSize * sizeof(int));
  t.status = (int *)bht_alloc(t, t.tableSize * sizeof(int));
  t.activeKeys = (int *)bht_alloc(t, t.tableSize * dim * sizeof(int));
  t.cnt = (int *)bht_alloc(t, sizeof(int));
  t.success = (int *)bht_alloc(t, sizeof(int));
  std::mt19937 rng(2);  // Bht.hpp:165-169; universal_hash(std::mt19937&) Bcht.hpp:39-43
  for (int f = 0; f < 3; ++f) {
    unsigned hx = (unsigned)(rng() % BHT_PRIME);
    if (hx < 1) hx = 1;
    unsigned hy = (unsigned)(rng() % BHT_PRIME);
    t.hf[2 * f] = hx;
    t.hf[2 * f + 1] = hy;
  }
  ZSR_CHECK(hipMemset(t.cnt, 0, sizeof(int)));
  int one = 1;
  ZSR_CHECK(hipMemcpy(t.success, &one, sizeof(int), hipMemcpyHostToDevice));
  bht_reset_table(t, nullptr);
  ZSR_CHECK(hipDeviceSynchronize());
}
static void bht_destroy(BhtHost &t) {
  (void)hipFree(t.keys); (void)hipFree(t.indices); (void)hipFree(t.status);
  (void)hipFree(t.activeKeys); (void)hipFree(t.cnt); (void)hipFree(t.success);
}
int bht_size(const BhtHost &t, hipStream_t s) {
  int n = 0;
  ZSR_CHECK(hipMemcpyAsync(&n, t.cnt, sizeof(int), hipMemcpyDeviceToHost, s));
  ZSR_CHECK(hipStreamSynchronize(s));
  return n;
}

// ------------------------------------------------------------------------------------ kernels
template <int DIM> __global__ __launch_bounds__(1024) void bht_insert_kernel(BhtDev t, const int *keys, size_t n, int *ret) {
  __shared__ unsigned smem[2 + 16];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int key[DIM] = {};
  if (valid) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) key[d] = keys[i * DIM + d];
  }
  int r = bht_insert_block<DIM>(t, key, valid, smem);
  if (valid && ret) ret[i] = r;
}
// the same with the cooperative probe (lanes of a tile examine one bucket together: bht_tile_find_or_claim)
template <int DIM> __global__ __launch_bounds__(256) void bht_insert_tile_kernel(BhtDev t, const int *keys, size_t n, int *ret) {
  __shared__ unsigned smem[2 + 4];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int key[DIM] = {};
  if (valid) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) key[d] = keys[i * DIM + d];
  }
  const int r = bht_commit_block<DIM>(t, key, bht_find_or_claim_tiled<DIM>(t, key, valid), smem);
  if (valid && ret) ret[i] = r;
}
// assign: table := {keys[i] -> i} (what a partition built elsewhere, e.g. a zs::HashTable's _activeKeys, needs to be used
// by the binned transfers); cnt = n
template <int DIM> __global__ __launch_bounds__(256) void bht_assign_kernel(BhtDev t, const int *keys, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int key[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) key[d] = keys[(size_t)i * DIM + d];
  bht_insert<DIM>(t, key, i, true);
}
template <int DIM> __global__ __launch_bounds__(256) void bht_query_kernel(BhtDev t, const int *keys, size_t n, int *ret) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int key[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) key[d] = keys[i * DIM + d];
  ret[i] = bht_query<DIM>(t, key);
}
// BhtInsertionOp (Bht.hpp:312-318): re-insert activeKeys[i] with fixed index i
template <int DIM> __global__ __launch_bounds__(256) void bht_reinsert_kernel(BhtDev t, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int key[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) key[d] = t.activeKeys[(size_t)i * DIM + d];
  bht_insert<DIM>(t, key, i, false);
}
// ReorderBht (Bht.hpp:342-375)
template <int DIM, bool SCATTER>
__global__ __launch_bounds__(256) void bht_reorder_kernel(BhtDev t, int *orderedKeys, const int *map, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int j = map[i];
  int key[DIM];
  const int src = SCATTER ? i : j, dst = SCATTER ? j : i;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    key[d] = t.activeKeys[(size_t)src * DIM + d];
    orderedKeys[(size_t)dst * DIM + d] = key[d];
  }
  int entry = bht_query<DIM, true>(t, key);
  if (entry != 0x7fffffff) t.indices[entry] = dst;
  else *t.success = 0;
}
__global__ void iota_kernel(int *p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
// gather component d of activeKeys through perm, biased to unsigned order
template <int DIM> __global__ void bht_gather_comp_kernel(const int *activeKeys, const int *perm, int n, int d, unsigned *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (unsigned)activeKeys[(size_t)perm[i] * DIM + d] ^ 0x80000000u;
}

template <int DIM> static void bht_insert_many(zs_rocm_policy *pol, BhtHost &t, const int *keys, size_t n, int *ret) {
  Launch L(pol, "bht_insert");
  if (!n) return;
#ifdef ZS_BHT_AB  // measurement builds: ZS_ROCM_BHT_TILE=1 builds with the cooperative probe (profiles/r06_bht.md)
  static const bool tiled = getenv("ZS_ROCM_BHT_TILE") && atoi(getenv("ZS_ROCM_BHT_TILE")) != 0;
  if (tiled) {
    hipLaunchKernelGGL((bht_insert_tile_kernel<DIM>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.dev(), keys, n, ret);
    return;
  }
#endif
  hipLaunchKernelGGL((bht_insert_kernel<DIM>), dim3(ceil_div(n, 1024)), dim3(1024), 0, L.stream, t.dev(), keys, n, ret);
}
template <int DIM> static void bht_query_many(zs_rocm_policy *pol, const BhtHost &t, const int *keys, size_t n, int *ret) {
  Launch L(pol, "bht_query");
  if (!n) return;
  hipLaunchKernelGGL((bht_query_kernel<DIM>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.dev(), keys, n, ret);
}

template <int DIM> static void bht_assign(zs_rocm_policy *pol, BhtHost &t, const int *keys, size_t n) {
  Launch L(pol, "bht_assign");
  bht_reset_table(t, L.stream);
  int cnt = (int)n, one = 1;
  ZSR_CHECK(hipMemcpyAsync(t.cnt, &cnt, sizeof(int), hipMemcpyHostToDevice, L.stream));
  ZSR_CHECK(hipMemcpyAsync(t.success, &one, sizeof(int), hipMemcpyHostToDevice, L.stream));
  if (n) hipLaunchKernelGGL((bht_assign_kernel<DIM>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.dev(), keys, (int)n);
  ZSR_CHECK(hipStreamSynchronize(L.stream));
}

// bht::resize (Bht.hpp:320-340)
template <int DIM> static void bht_resize(zs_rocm_policy *pol, BhtHost &t, size_t newCapacity) {
  size_t ns = evaluate_table_size(newCapacity, t.bucket);
  if (ns <= t.tableSize) return;
  Launch L(pol, "bht_resize");
  const int n = bht_size(t, L.stream);
  const int ks = DIM == 1 ? 1 : (DIM == 2 ? 2 : 4);
  int *oldKeys = t.keys, *oldIdx = t.indices, *oldSt = t.status, *oldActive = t.activeKeys;
  const size_t oldSize = t.tableSize;
  t.tableSize = ns;
  t.keys = (int *)bht_alloc(t, ns * ks * sizeof(int));
  t.indices = (int *)bht_alloc(t, ns * sizeof(int));
  t.status = (int *)bht_alloc(t, ns * sizeof(int));
  t.activeKeys = (int *)bht_alloc(t, ns * DIM * sizeof(int));
  if (n) ZSR_CHECK(hipMemcpyAsync(t.activeKeys, oldActive, (size_t)n * DIM * sizeof(int), hipMemcpyDeviceToDevice, L.stream));
  bht_reset_table(t, L.stream);
  if (n) hipLaunchKernelGGL((bht_reinsert_kernel<DIM>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.dev(), n);
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  (void)oldSize;
  (void)hipFree(oldKeys); (void)hipFree(oldIdx); (void)hipFree(oldSt); (void)hipFree(oldActive);
}

// bht::reorder (Bht.hpp:377-400)
template <int DIM> static void bht_reorder_impl(Launch &L, BhtHost &t, const int *map, bool scatter, int n) {
  if (!n) return;
  int *ordered = (int *)bht_alloc(t, t.tableSize * DIM * sizeof(int));
  if (scatter)
    hipLaunchKernelGGL((bht_reorder_kernel<DIM, true>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.dev(), ordered, map, n);
  else
    hipLaunchKernelGGL((bht_reorder_kernel<DIM, false>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.dev(), ordered, map, n);
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  (void)hipFree(t.activeKeys);
  t.activeKeys = ordered;
}

// canonical numbering: active keys in lexicographic order (component 0 most significant), LSD over components
// axes: NULL, or a permutation of 0 .. DIM-1 -- axes[0] is the most significant component of the comparison, axes[DIM-1] the one that
// changes fastest along the numbering
// first: entries [0, first) keep their numbers, only [first, size) are sorted (the apron blocks of an MPM partition behind its holders)
template <int DIM> static int bht_canonicalize(zs_rocm_policy *pol, BhtHost &t, const int *axes = nullptr, size_t first = 0) {
  int ax[DIM];
  unsigned seen = 0u;
  for (int d = 0; d < DIM; ++d) {
    ax[d] = axes ? axes[d] : d;
    if (ax[d] < 0 || ax[d] >= DIM || ((seen >> ax[d]) & 1u)) return -1;
    seen |= 1u << ax[d];
  }
  Launch L(pol, "bht_canonicalize");
  const int n = bht_size(t, L.stream);
  if (first >= (size_t)n || n - (int)first <= 1) return 0;
  const int f = (int)first, m = n - f;
  int *perm[2] = {(int *)L.temp(sizeof(int) * n), (int *)L.temp(sizeof(int) * n)};
  unsigned *comp = (unsigned *)L.temp(sizeof(unsigned) * m), *sorted = (unsigned *)L.temp(sizeof(unsigned) * m);
  hipLaunchKernelGGL(iota_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, perm[0], n);
  if (f) hipLaunchKernelGGL(iota_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, perm[1], n);  // (the head stays the identity in both)
  int cur = 0;
  for (int d = DIM - 1; d >= 0; --d) {
    hipLaunchKernelGGL((bht_gather_comp_kernel<DIM>), dim3(ceil_div(m, 256)), dim3(256), 0, L.stream, t.activeKeys, perm[cur] + f, m,
                       ax[d], comp);
    radix_sort_pair_u32(L, comp, perm[cur] + f, sorted, perm[cur ^ 1] + f, (size_t)m, 0, 32);
    cur ^= 1;
  }
  bht_reorder_impl<DIM>(L, t, perm[cur], /*scatter=*/false, n);  // new key i = old key perm[i]
  return 0;
}

// Morton numbering: active keys along the Z-order curve of (key - min key).  Consecutive numbers are spatial neighbours in every
// dimension, which is what the per-block kernels of the MPM path want from their launch order (blocks that share apron nodes run
// close in time and, with the chunked XCD mapping, in the same L2).  64 / DIM bits per component; an ordering only, any key range works.
template <int DIM> __global__ void bht_key_min_kernel(const int *activeKeys, int n, int *mn) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    for (int d = 0; d < DIM; ++d) atomicMin(mn + d, activeKeys[(size_t)i * DIM + d]);
}
template <int DIM> __global__ void bht_morton_kernel(const int *activeKeys, int n, const int *mn, unsigned long long *code, int *perm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int BITS = 64 / DIM;
  unsigned long long c = 0;
  for (int d = 0; d < DIM; ++d) {
    const unsigned long long v = (unsigned long long)(unsigned)(activeKeys[(size_t)i * DIM + d] - mn[d]);
    for (int b = 0; b < BITS && b < 32; ++b) c |= ((v >> b) & 1ull) << (b * DIM + (DIM - 1 - d));
  }
  code[i] = c;
  perm[i] = i;
}
template <int DIM> static void bht_order_morton(zs_rocm_policy *pol, BhtHost &t) {
  Launch L(pol, "bht_order_morton");
  const int n = bht_size(t, L.stream);
  if (n <= 1) return;
  int *perm[2] = {(int *)L.temp(sizeof(int) * n), (int *)L.temp(sizeof(int) * n)};
  unsigned long long *code = (unsigned long long *)L.temp(sizeof(unsigned long long) * n),
                     *sorted = (unsigned long long *)L.temp(sizeof(unsigned long long) * n);
  int *mn = (int *)L.temp(sizeof(int) * 4);
  ZSR_CHECK(hipMemsetAsync(mn, 0x7f, sizeof(int) * 4, L.stream));
  hipLaunchKernelGGL((bht_key_min_kernel<DIM>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.activeKeys, n, mn);
  hipLaunchKernelGGL((bht_morton_kernel<DIM>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.activeKeys, n, mn, code, perm[0]);
  radix_sort_pair_u64(L, code, perm[0], sorted, perm[1], (size_t)n, 0, 64);
  bht_reorder_impl<DIM>(L, t, perm[1], /*scatter=*/false, n);
}

static zs_rocm_bht_view_lite *bht_make_view(const BhtHost &t) {  // py_interop/BhtInstantiations.cpp:62-110
  auto *v = new zs_rocm_bht_view_lite;
  v->keys = t.keys; v->indices = t.indices; v->status = t.status; v->activeKeys = t.activeKeys;
  v->cnt = t.cnt; v->success = t.success; v->tableSize = (uint32_t)t.tableSize;
  v->numBuckets = (uint32_t)(t.tableSize / (size_t)t.bucket);
  v->hf0x = t.hf[0]; v->hf0y = t.hf[1]; v->hf1x = t.hf[2]; v->hf1y = t.hf[3];
  v->hf2x = t.hf[4]; v->hf2y = t.hf[5];
  return v;
}
// clone(mloc) + swap between device and unified memory (a host-resident table cannot serve a device policy)
static void bht_relocate(BhtHost &t, int memsrc, int8_t devid) {
  const int target = memsrc == 0 ? 1 : memsrc;
  t.devid = devid;
  if (target == t.memsrc) return;
  BhtHost n = t;
  n.memsrc = target;
  const int ks = t.dim == 1 ? 1 : (t.dim == 2 ? 2 : 4);
  const size_t ts = t.tableSize;
  n.keys = (int *)bht_alloc(n, ts * ks * sizeof(int));
  n.indices = (int *)bht_alloc(n, ts * sizeof(int));
  n.status = (int *)bht_alloc(n, ts * sizeof(int));
  n.activeKeys = (int *)bht_alloc(n, ts * t.dim * sizeof(int));
  n.cnt = (int *)bht_alloc(n, sizeof(int));
  n.success = (int *)bht_alloc(n, sizeof(int));
  ZSR_CHECK(hipMemcpy(n.keys, t.keys, ts * ks * sizeof(int), hipMemcpyDefault));
  ZSR_CHECK(hipMemcpy(n.indices, t.indices, ts * sizeof(int), hipMemcpyDefault));
  ZSR_CHECK(hipMemcpy(n.status, t.status, ts * sizeof(int), hipMemcpyDefault));
  ZSR_CHECK(hipMemcpy(n.activeKeys, t.activeKeys, ts * t.dim * sizeof(int), hipMemcpyDefault));
  ZSR_CHECK(hipMemcpy(n.cnt, t.cnt, sizeof(int), hipMemcpyDefault));
  ZSR_CHECK(hipMemcpy(n.success, t.success, sizeof(int), hipMemcpyDefault));
  bht_destroy(t);
  t = n;
}

}  // namespace zsr

using namespace zsr;

extern "C" {

#define ZSR_DEFINE_BHT_A(D, B, SFX)                                                                         \
  zs_rocm_bht_##D *container__bht_int_##D##_int_##B##SFX(zs_rocm_allocator *a, size_t n) {                   \
    auto *b = new zs_rocm_bht_##D;                                                                          \
    bht_create(b->t, D, B, a ? a->memsrc : 1, a ? a->devid : (int8_t)current_device(), n);                                         \
    return b;                                                                                               \
  }                                                                                                         \
  void del_container__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *b) {                                       \
    bht_destroy(b->t);                                                                                      \
    delete b;                                                                                               \
  }                                                                                                         \
  void relocate_container__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *b, int memsrc, int8_t devid) {        \
    bht_relocate(b->t, memsrc, devid);                                                                      \
  }                                                                                                         \
  size_t container_size__bht_int_##D##_int_##B##SFX(const zs_rocm_bht_##D *b) { return (size_t)bht_size(b->t, nullptr); } \
  size_t container_capacity__bht_int_##D##_int_##B##SFX(const zs_rocm_bht_##D *b) { return b->t.tableSize; } \
  void reset_container__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *b, int clearCnt) { /* Bht.hpp:306-318 */ \
    bht_reset_table(b->t, nullptr);                                                                         \
    if (clearCnt) ZSR_CHECK(hipMemsetAsync(b->t.cnt, 0, sizeof(int), nullptr));                             \
    int one = 1;                                                                                            \
    ZSR_CHECK(hipMemcpy(b->t.success, &one, sizeof(int), hipMemcpyHostToDevice));                           \
    ZSR_CHECK(hipDeviceSynchronize());                                                                      \
  }                                                                                                         \
  zs_rocm_bht_view_lite *pyview__bht_int_##D##_int_##B##SFX(zs_rocm_bht_##D *b) { return bht_make_view(b->t); } \
  zs_rocm_bht_view_lite *pyview__bht_const_int_##D##_int_##B##SFX(const zs_rocm_bht_##D *b) { return bht_make_view(b->t); } \
  void resize_container__rocm_bht_int_##D##_int_##B##SFX(zs_rocm_policy *pol, zs_rocm_bht_##D *b, size_t cap) { \
    bht_resize<D>(pol, b->t, cap);                                                                          \
  }
#define ZSR_DEFINE_BHT(D, B)                                                                                \
  ZSR_DEFINE_BHT_A(D, B, )                                                                                  \
  ZSR_DEFINE_BHT_A(D, B, _virtual)                                                                          \
  void del_pyview__bht_int_##D##_int_##B(zs_rocm_bht_view_lite *v) { delete v; }                            \
  void zs_rocm_insert__bht_int_##D##_int_##B(zs_rocm_policy *pol, zs_rocm_bht_##D *b, const int *keys,       \
                                            size_t n, int *ret) {                                           \
    bht_insert_many<D>(pol, b->t, keys, n, ret);                                                            \
  }                                                                                                         \
  void zs_rocm_assign__bht_int_##D##_int_##B(zs_rocm_policy *pol, zs_rocm_bht_##D *b, const int *keys, size_t n) { \
    bht_assign<D>(pol, b->t, keys, n);                                                                      \
  }                                                                                                         \
  void zs_rocm_query__bht_int_##D##_int_##B(zs_rocm_policy *pol, const zs_rocm_bht_##D *b, const int *keys,  \
                                           size_t n, int *ret) {                                            \
    bht_query_many<D>(pol, b->t, keys, n, ret);                                                             \
  }                                                                                                         \
  void zs_rocm_reorder__bht_int_##D##_int_##B(zs_rocm_policy *pol, zs_rocm_bht_##D *b, const int *map,       \
                                             int scatter) {                                                 \
    Launch L(pol, "bht_reorder");                                                                           \
    bht_reorder_impl<D>(L, b->t, map, scatter != 0, bht_size(b->t, L.stream));                              \
  }                                                                                                         \
  void zs_rocm_canonicalize__bht_int_##D##_int_##B(zs_rocm_policy *pol, zs_rocm_bht_##D *b) {                \
    bht_canonicalize<D>(pol, b->t);                                                                         \
  }                                                                                                         \
  int zs_rocm_canonicalize_axes__bht_int_##D##_int_##B(zs_rocm_policy *pol, zs_rocm_bht_##D *b, const int *axes) {  \
    return bht_canonicalize<D>(pol, b->t, axes);                                                            \
  }                                                                                                         \
  int zs_rocm_canonicalize_tail__bht_int_##D##_int_##B(zs_rocm_policy *pol, zs_rocm_bht_##D *b, const int *axes, size_t first) {  \
    return bht_canonicalize<D>(pol, b->t, axes, first);                                                     \
  }                                                                                                         \
  void zs_rocm_order_morton__bht_int_##D##_int_##B(zs_rocm_policy *pol, zs_rocm_bht_##D *b) {                \
    bht_order_morton<D>(pol, b->t);                                                                         \
  }
ZSR_DEFINE_BHT(1, 16)
ZSR_DEFINE_BHT(2, 16)
ZSR_DEFINE_BHT(3, 16)
ZSR_DEFINE_BHT(4, 16)
ZSR_DEFINE_BHT(1, 32)
ZSR_DEFINE_BHT(2, 32)
ZSR_DEFINE_BHT(3, 32)
ZSR_DEFINE_BHT(4, 32)

}  // extern "C"
