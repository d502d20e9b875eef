// hashtable.hpp -- host-side object of zs::HashTable<i32, dim, int> (container/HashTable.hpp:16-312); device protocol in
// include/zensim_rocm/hashtable_device.hpp.
#pragma once
#include "common.hpp"
#include "../../include/zensim_rocm/hashtable_device.hpp"

struct zs_rocm_hashtable {
  int dim = 3, memsrc = 1;
  int8_t devid = 0;
  size_t tableSize = 0;
  int *keys = nullptr, *indices = nullptr, *status = nullptr, *activeKeys = nullptr, *cnt = nullptr;
  zsr::HtDev dev() const {
    zsr::HtDev d;
    d.keys = keys; d.indices = indices; d.status = status; d.activeKeys = activeKeys; d.cnt = cnt;
    d.tableSize = (int)tableSize;
    return d;
  }
};

// zs::IndexBuckets<3, i32, i32> (container/IndexBuckets.hpp:9-67)
struct zs_rocm_index_buckets {
  zs_rocm_hashtable *table = nullptr;
  int *indices = nullptr, *offsets = nullptr, *counts = nullptr;
  int numBuckets = 0, numEntries = 0;
  float dx = 1.f;
  float displacement = 0.5f;  // coord_offset the buckets were built with (Query.tpp:11)
  int dense = 0, denseSide = 0;  // zs_rocm_index_buckets_for_partition: bucket number = block * side^3 + cell id, no hash table
  size_t capEntries = 0, capCells = 0, tableFor = 0;  // allocation sizes kept across rebuilds (a time loop rebuilds every step)
};
