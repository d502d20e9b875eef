// hashtable.hip -- zs::HashTable<i32, dim, int> (container/HashTable.hpp): the open-addressing table (hash_combine hash,
// linear probing with stride 127) behind `partition_for_particles` (simulation/sparsity/SparsityCompute.tpp:5-24) and the
// Grids-based MPM path (simulation/mpm/Simulator.cpp:122).  The reference exposes it only as a C++ template; the bulk
// entry points below replace the `pol(range(n), [t = proxy<space>(table)](i){ t.insert(key_i); })` idiom.
#include "hashtable.hpp"

namespace zsr {

static size_t ht_next_2pow(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
static size_t ht_table_size(size_t entryCnt) {  // evaluateTableSize, HashTable.hpp:87-90 (reserve_ratio_v = 16)
  return entryCnt == 0 ? 0 : ht_next_2pow(entryCnt) * 16;
}
static void *ht_alloc(const zs_rocm_hashtable &t, size_t bytes) {
  void *p = nullptr;
  if (bytes == 0) bytes = 16;
  DeviceGuard guard(t.devid);  // the container lives on ITS device, whatever the calling thread is on
  if (t.memsrc == 2) ZSR_CHECK(hipMallocManaged(&p, bytes));
  else ZSR_CHECK(hipMalloc(&p, bytes));
  return p;
}
static void ht_alloc_tables(zs_rocm_hashtable &t, size_t ts) {
  t.tableSize = ts;
  t.keys = (int *)ht_alloc(t, ts * t.dim * sizeof(int));
  t.indices = (int *)ht_alloc(t, ts * sizeof(int));
  t.status = (int *)ht_alloc(t, ts * sizeof(int));
}

// ResetHashTable / CleanSparsity (HashTable.hpp:212-228, simulation/sparsity/SparsityOp.hpp:42-57)
__global__ __launch_bounds__(256) void ht_reset_kernel(HtDev t, int dim, int clearCnt) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)t.tableSize) return;
  for (int d = 0; d < dim; ++d) t.keys[e * dim + d] = HT_SENT;
  t.indices[e] = -1;
  t.status[e] = -1;
  if (e == 0 && clearCnt) *t.cnt = 0;
}
template <int DIM> __global__ __launch_bounds__(1024) void ht_insert_kernel(HtDev t, const int *keys, size_t n, int *ret) {
  // dense indices of all slots claimed by the workgroup come from ONE atomic on cnt (see bht_insert_block)
  __shared__ unsigned smem[2 + 16];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int key[DIM] = {};
  if (valid) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) key[d] = keys[i * DIM + d];
  }
  const int e = valid ? ht_find_or_claim<DIM>(t, key) : -1;
  const bool won = e >= 0;
  const unsigned long long m = __ballot(won);
  const int lane = lane_id(), w = wave_id(), nw = (int)((blockDim.x + 63) >> 6);
  if (lane == 0) smem[2 + w] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (int j = 0; j < nw; ++j) {
      const unsigned c = smem[2 + j];
      smem[2 + j] = tot;
      tot += c;
    }
    smem[0] = tot ? atomicAdd((unsigned *)t.cnt, tot) : 0u;
  }
  __syncthreads();
  int r = e;  // -1 present / HT_FAIL
  if (won) {
    r = (int)(smem[0] + smem[2 + w] + (unsigned)__popcll(m & ((1ull << lane) - 1ull)));
    t.indices[e] = r;
#pragma unroll
    for (int d = 0; d < DIM; ++d) t.activeKeys[(size_t)r * DIM + d] = key[d];
  }
  if (valid && ret) ret[i] = r;
}
template <int DIM> __global__ __launch_bounds__(256) void ht_insert_ids_kernel(HtDev t, const int *keys, const int *ids, size_t n, int *ok) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int key[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) key[d] = keys[i * DIM + d];
  const bool r = ht_insert_id<DIM>(t, key, ids ? ids[i] : (int)i);
  if (ok) ok[i] = r ? 1 : 0;
}
template <int DIM, bool ENTRY> __global__ __launch_bounds__(256) void ht_query_kernel(HtDev t, const int *keys, size_t n, int *ret) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int key[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) key[d] = keys[i * DIM + d];
  ret[i] = ht_query<DIM, ENTRY>(t, key);
}
// ReinsertHashTable (HashTable.hpp:229-238): table.insert(activeKeys[entry], entry)
template <int DIM> __global__ __launch_bounds__(256) void ht_reinsert_kernel(HtDev t, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int key[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) key[d] = t.activeKeys[(size_t)i * DIM + d];
  ht_insert_id<DIM>(t, key, i);
}
// RemoveHashTableEntries (HashTable.hpp:239-256)
template <int DIM> __global__ __launch_bounds__(256) void ht_remove_kernel(HtDev t, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int key[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) key[d] = t.activeKeys[(size_t)i * DIM + d];
  const int e = ht_query<DIM, true>(t, key);
  if (e < 0) return;
  // all entries of the table are removed by this launch, so queries of concurrent threads that run into an
  // already-cleared slot of their chain still terminate correctly only if they find their own key first: clear keys last
  t.indices[e] = -2;  // tombstone for the duration of the launch (keeps probe chains alive), reset below
#pragma unroll
  for (int d = 0; d < DIM; ++d) t.keys[(size_t)e * DIM + d] = HT_SENT;
}
__global__ __launch_bounds__(256) void ht_untomb_kernel(HtDev t) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (size_t)t.tableSize && t.indices[e] == -2) t.indices[e] = -1;
}

#define ZSR_HT_DISPATCH(dim, CALL)  \
  switch (dim) {                    \
    case 1: { constexpr int D = 1; CALL; } break; \
    case 2: { constexpr int D = 2; CALL; } break; \
    case 3: { constexpr int D = 3; CALL; } break; \
    default: { constexpr int D = 4; CALL; } break; \
  }

static void ht_reset(Launch &L, zs_rocm_hashtable &t, bool clearCnt) {
  if (t.tableSize)
    hipLaunchKernelGGL(ht_reset_kernel, dim3(ceil_div(t.tableSize, 256)), dim3(256), 0, L.stream, t.dev(), t.dim, clearCnt ? 1 : 0);
  else if (clearCnt)
    ZSR_CHECK(hipMemsetAsync(t.cnt, 0, sizeof(int), L.stream));
}
static int ht_size(const zs_rocm_hashtable &t, hipStream_t s) {
  int n = 0;
  ZSR_CHECK(hipMemcpyAsync(&n, t.cnt, sizeof(int), hipMemcpyDeviceToHost, s));
  ZSR_CHECK(hipStreamSynchronize(s));
  return n;
}
static void ht_reinsert(Launch &L, zs_rocm_hashtable &t, int n) {
  if (n <= 0) return;
  ZSR_HT_DISPATCH(t.dim, hipLaunchKernelGGL((ht_reinsert_kernel<D>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t.dev(), n));
}
static void ht_grow(Launch &L, zs_rocm_hashtable &t, size_t newSize, int keepActive) {
  int *ok = t.keys, *oi = t.indices, *os = t.status, *oa = t.activeKeys;
  ht_alloc_tables(t, newSize);
  t.activeKeys = (int *)ht_alloc(t, newSize * t.dim * sizeof(int));
  if (keepActive > 0)
    ZSR_CHECK(hipMemcpyAsync(t.activeKeys, oa, (size_t)keepActive * t.dim * sizeof(int), hipMemcpyDeviceToDevice, L.stream));
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  (void)hipFree(ok); (void)hipFree(oi); (void)hipFree(os); (void)hipFree(oa);
}

}  // namespace zsr

using namespace zsr;

extern "C" {

zs_rocm_hashtable *zs_rocm_hashtable_create(int dim, size_t numExpectedEntries, int memsrc, int devid) {
  if (dim < 1 || dim > 4) return nullptr;
  auto *t = new zs_rocm_hashtable;
  t->dim = dim;
  t->memsrc = memsrc == 0 ? 1 : memsrc;
  t->devid = (int8_t)devid;
  DeviceGuard guard(devid);
  ht_alloc_tables(*t, ht_table_size(numExpectedEntries));
  t->activeKeys = (int *)ht_alloc(*t, t->tableSize * dim * sizeof(int));
  t->cnt = (int *)ht_alloc(*t, sizeof(int));
  ZSR_CHECK(hipMemset(t->cnt, 0, sizeof(int)));  // _cnt.setVal(0), HashTable.hpp:96
  // the reference leaves the table uninitialised until reset()/CleanSparsity; it is reset here so that a fresh table is usable
  if (t->tableSize) hipLaunchKernelGGL(ht_reset_kernel, dim3(ceil_div(t->tableSize, 256)), dim3(256), 0, nullptr, t->dev(), dim, 1);
  ZSR_CHECK(hipDeviceSynchronize());
  return t;
}
void zs_rocm_hashtable_destroy(zs_rocm_hashtable *t) {
  if (!t) return;
  (void)hipFree(t->keys); (void)hipFree(t->indices); (void)hipFree(t->status); (void)hipFree(t->activeKeys); (void)hipFree(t->cnt);
  delete t;
}
int zs_rocm_hashtable_dim(const zs_rocm_hashtable *t) { return t->dim; }
size_t zs_rocm_hashtable_table_size(const zs_rocm_hashtable *t) { return t->tableSize; }
int zs_rocm_hashtable_size(const zs_rocm_hashtable *t) { return ht_size(*t, nullptr); }
void zs_rocm_hashtable_get_view(const zs_rocm_hashtable *t, zs_rocm_hashtable_view *v) {
  v->keys = t->keys; v->indices = t->indices; v->status = t->status; v->activeKeys = t->activeKeys; v->cnt = t->cnt;
  v->tableSize = (int)t->tableSize;
}
void zs_rocm_hashtable_reset(zs_rocm_policy *pol, zs_rocm_hashtable *t, int clearCnt) {
  Launch L(pol, "hashtable_reset");
  ht_reset(L, *t, clearCnt != 0);
}
void zs_rocm_hashtable_insert(zs_rocm_policy *pol, zs_rocm_hashtable *t, const int *keys, size_t n, int *ret) {
  Launch L(pol, "hashtable_insert");
  if (!n) return;
  ZSR_HT_DISPATCH(t->dim, hipLaunchKernelGGL((ht_insert_kernel<D>), dim3(ceil_div(n, 1024)), dim3(1024), 0, L.stream, t->dev(), keys, n, ret));
}
void zs_rocm_hashtable_insert_ids(zs_rocm_policy *pol, zs_rocm_hashtable *t, const int *keys, const int *ids, size_t n, int *ok) {
  Launch L(pol, "hashtable_insert_ids");
  if (!n) return;
  ZSR_HT_DISPATCH(t->dim, hipLaunchKernelGGL((ht_insert_ids_kernel<D>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t->dev(), keys, ids, n, ok));
}
void zs_rocm_hashtable_query(zs_rocm_policy *pol, const zs_rocm_hashtable *t, const int *keys, size_t n, int *ret) {
  Launch L(pol, "hashtable_query");
  if (!n) return;
  ZSR_HT_DISPATCH(t->dim, hipLaunchKernelGGL((ht_query_kernel<D, false>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t->dev(), keys, n, ret));
}
void zs_rocm_hashtable_entry(zs_rocm_policy *pol, const zs_rocm_hashtable *t, const int *keys, size_t n, int *ret) {
  Launch L(pol, "hashtable_entry");
  if (!n) return;
  ZSR_HT_DISPATCH(t->dim, hipLaunchKernelGGL((ht_query_kernel<D, true>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t->dev(), keys, n, ret));
}
// HashTable::resize (HashTable.hpp:281-292)
void zs_rocm_hashtable_resize(zs_rocm_policy *pol, zs_rocm_hashtable *t, size_t numExpectedEntries) {
  const size_t ns = ht_table_size(numExpectedEntries);
  if (ns <= t->tableSize) return;
  Launch L(pol, "hashtable_resize");
  const int n = ht_size(*t, L.stream);
  ht_grow(L, *t, ns, n);
  ht_reset(L, *t, false);
  ht_reinsert(L, *t, n);
}
// HashTable::preserve (HashTable.hpp:258-279): cnt := numExpectedEntries, the first min(old, new) active keys keep their indices
void zs_rocm_hashtable_preserve(zs_rocm_policy *pol, zs_rocm_hashtable *t, size_t numExpectedEntries) {
  Launch L(pol, "hashtable_preserve");
  const int n = ht_size(*t, L.stream);
  if (numExpectedEntries == (size_t)n) return;
  const int newCnt = (int)numExpectedEntries;
  const size_t ns = ht_table_size(numExpectedEntries);
  const int keep = n < newCnt ? n : newCnt;
  if (ns > t->tableSize) {
    ht_grow(L, *t, ns, n);
    ht_reset(L, *t, false);
  } else if (n > 0) {
    ZSR_HT_DISPATCH(t->dim, hipLaunchKernelGGL((ht_remove_kernel<D>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t->dev(), n));
    hipLaunchKernelGGL(ht_untomb_kernel, dim3(ceil_div(t->tableSize, 256)), dim3(256), 0, L.stream, t->dev());
  }
  ZSR_CHECK(hipMemcpyAsync(t->cnt, &newCnt, sizeof(int), hipMemcpyHostToDevice, L.stream));
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  ht_reinsert(L, *t, keep);
}

}  // extern "C"
