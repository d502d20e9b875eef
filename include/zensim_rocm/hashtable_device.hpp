// hashtable_device.hpp -- device view + insert/query protocol of zs::HashTable<i32, dim, int> (container/HashTable.hpp:16-592),
// the open-addressing table the reference's Grids-based MPM path partitions space with (simulation/sparsity/
// SparsityCompute.tpp:13, simulation/mpm/Simulator.cpp:122).
//
// Layout as in the reference: keys [tableSize] of PACKED vec<int, dim> (dim ints per slot, no padding; empty = INT_MAX in
// every component, HashTable.hpp:65), indices [tableSize] (-1 = empty, terminates queries), status [tableSize] (-1),
// activeKeys [tableSize][dim], cnt; tableSize = next_2pow(n) * 16 (:87-90).  hash = hash_combine chain over the raw
// coordinates with a 64-bit seed truncated to int (:496-500, math/Hash.hpp:19-28), entry = ((h % size) + size) % size,
// linear probing with stride 127 (:362,454).
//
// Insert protocol.  The reference takes the per-slot status spin lock on EVERY probe (atomicKeyCAS, :508-541) inside a
// ballot loop.  Here probes are lock-free: a slot is written exactly once (empty -> key) by a writer that holds
// status[slot] (-1 -> 0, the reference's lock value), stores the components, drains, and unlocks.  A probe that reads a
// mixture of INT_MAX and other components (a half-written slot, or a genuine key containing INT_MAX) consults status
// after its key loads have returned: locked => look again; unlocked => the writer has drained and a second read is complete.
#pragma once
#include <hip/hip_runtime.h>

namespace zsr {

constexpr int HT_SENT = 0x7fffffff;      // key_scalar_sentinel_v = numeric max (HashTable.hpp:65)
constexpr int HT_FAIL = (int)0x80000000; // table full (the reference would probe forever)

struct HtDev {
  int *keys;        // [tableSize][dim]
  int *indices;     // [tableSize]
  int *status;      // [tableSize]
  int *activeKeys;  // [tableSize][dim]
  int *cnt;
  int tableSize;
};

// do_hash (HashTable.hpp:496-500): size_t seed = key[0]; hash_combine(seed, key[d]); truncated to value_t
template <int DIM> __host__ __device__ __forceinline__ int ht_hash(const int *key) {
  unsigned long long ret = (unsigned long long)(long long)key[0];
#pragma unroll
  for (int d = 1; d < DIM; ++d)
    ret ^= ((unsigned long long)(long long)key[d] + 0x9e3779b97f4a7c15ull + (ret << 12) + (ret >> 4));
  return (int)ret;
}
template <int DIM> __host__ __device__ __forceinline__ int ht_home(const int *key, int tableSize) {
  const long long h = (long long)ht_hash<DIM>(key);
  return (int)(((h % tableSize) + tableSize) % tableSize);
}

// 1 found, 0 empty, -1 other key, 2 busy
template <int DIM> __device__ __forceinline__ int ht_probe(const HtDev &t, int e, const int *key) {
  const int *slot = t.keys + (size_t)e * DIM;
  int k[DIM];
  for (int pass = 0; pass < 2; ++pass) {
    int nsent = 0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      k[d] = __hip_atomic_load(slot + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nsent += k[d] == HT_SENT;
    }
    if (nsent == 0 || nsent == DIM || pass == 1) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the key loads have returned before status is read
    if (__hip_atomic_load(t.status + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1) return 2;
  }
  bool eq = true, empty = true;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    eq = eq && k[d] == key[d];
    empty = empty && k[d] == HT_SENT;
  }
  return eq ? 1 : (empty ? 0 : -1);
}

template <int DIM> __device__ __forceinline__ bool ht_claim(const HtDev &t, int e, const int *key) {
  int *slot = t.keys + (size_t)e * DIM;
  if (atomicCAS(t.status + e, -1, 0) != -1) return false;
  bool empty = true;
#pragma unroll
  for (int d = 0; d < DIM; ++d) empty = empty && __hip_atomic_load(slot + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == HT_SENT;
  if (empty) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) __hip_atomic_store(slot + d, key[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the key is at the coherence point before the slot unlocks
  __hip_atomic_store(t.status + e, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return empty;
}

// slot claimed by THIS thread (>= 0), -1 if the key is already present, HT_FAIL if the table is full
template <int DIM> __device__ __forceinline__ int ht_find_or_claim(const HtDev &t, const int *key) {
  if (t.tableSize <= 0) return HT_FAIL;
  int e = ht_home<DIM>(key, t.tableSize);
  for (int visited = 0; visited < t.tableSize;) {
    const int st = ht_probe<DIM>(t, e, key);
    if (st == 2) continue;
    if (st == 1) return -1;
    if (st == 0) {
      if (ht_claim<DIM>(t, e, key)) return e;
      continue;  // lost the slot: look at it again (it may now hold our key)
    }
    e = (e + 127) % t.tableSize;
    ++visited;
  }
  return HT_FAIL;
}
// HashTableView::insert(key) (HashTable.hpp:353-374): dense index for the inserting thread, sentinel_v (-1) otherwise
template <int DIM> __device__ __forceinline__ int ht_insert(const HtDev &t, const int *key) {
  const int e = ht_find_or_claim<DIM>(t, key);
  if (e < 0) return e;
  const int no = (int)atomicAdd((unsigned *)t.cnt, 1u);
  t.indices[e] = no;
#pragma unroll
  for (int d = 0; d < DIM; ++d) t.activeKeys[(size_t)no * DIM + d] = key[d];
  return no;
}
// HashTableView::insert(key, id) (:405-421): true when this call created the entry
template <int DIM> __device__ __forceinline__ bool ht_insert_id(const HtDev &t, const int *key, int id) {
  const int e = ht_find_or_claim<DIM>(t, key);
  if (e < 0) return false;
  t.indices[e] = id;
  return true;
}
// HashTableView::query / entry (:445-470), table quiescent.  The wrap test is `>=` (the reference writes `>`, which lets
// hashedentry == tableSize read one slot past the end; SURVEY.md 8c lists it as a defect not to replicate).
template <int DIM, bool ENTRY = false> __device__ __forceinline__ int ht_query(const HtDev &t, const int *key) {
  if (t.tableSize <= 0) return -1;
  int e = ht_home<DIM>(key, t.tableSize);
  for (int visited = 0; visited < t.tableSize; ++visited) {
    const int *slot = t.keys + (size_t)e * DIM;
    bool eq = true;
#pragma unroll
    for (int d = 0; d < DIM; ++d) eq = eq && slot[d] == key[d];
    if (eq) return ENTRY ? e : t.indices[e];
    if (t.indices[e] == -1) return -1;
    e += 127;
    if (e >= t.tableSize) e %= t.tableSize;
  }
  return -1;
}

}  // namespace zsr
