/*
 * hashtable.c -- CPU restatement of zs::HashTable<i32, dim, int> (container/HashTable.hpp:16-592).
 * TEST INFRASTRUCTURE ONLY (see zpc_oracle.h).  Sequential insertion in input order (the host insert path,
 * HashTable.hpp:376-397, run by one thread).
 */
#include "zpc_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_hashtable {
  int dim;
  int32_t tableSize;   /* value_t _tableSize */
  int32_t *keys;       /* [tableSize][dim] packed vec<int, dim>; empty = INT_MAX in every component (:65) */
  int32_t *indices;    /* -1 = empty (sentinel_v, :66) */
  int32_t *status;     /* -1 (:67) */
  int32_t *activeKeys; /* [tableSize][dim] */
  int32_t cnt;
};

static size_t ht_next_2pow(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
/* evaluateTableSize, HashTable.hpp:87-90: next_2pow(n) * reserve_ratio_v (16) */
size_t orc_hashtable_table_size(size_t entryCnt) { return entryCnt == 0 ? 0 : ht_next_2pow(entryCnt) * 16; }

/* do_hash, HashTable.hpp:496-500: size_t ret = key[0]; hash_combine(ret, key[d]) -- 64-bit branch of math/Hash.hpp:19-28:
   seed ^= val + 0x9e3779b97f4a7c15 + (seed << 12) + (seed >> 4), val sign-extended; result truncated to value_t (int) */
int32_t orc_hashtable_do_hash(const int32_t *key, int dim) {
  uint64_t ret = (uint64_t)(int64_t)key[0];
  for (int d = 1; d < dim; ++d) ret ^= ((uint64_t)(int64_t)key[d] + 0x9e3779b97f4a7c15ull + (ret << 12) + (ret >> 4));
  return (int32_t)ret;
}
static int32_t home(const orc_hashtable *t, const int32_t *key) { /* (do_hash % size + size) % size, :357 */
  int64_t h = orc_hashtable_do_hash(key, t->dim);
  return (int32_t)(((h % t->tableSize) + t->tableSize) % t->tableSize);
}

orc_hashtable *orc_hashtable_create(int dim, size_t nExpected) {
  orc_hashtable *t = (orc_hashtable *)calloc(1, sizeof(orc_hashtable));
  t->dim = dim;
  t->tableSize = (int32_t)orc_hashtable_table_size(nExpected);
  size_t ts = (size_t)t->tableSize;
  t->keys = (int32_t *)malloc(ts * (size_t)dim * 4 + 16);
  t->indices = (int32_t *)malloc(ts * 4 + 16);
  t->status = (int32_t *)malloc(ts * 4 + 16);
  t->activeKeys = (int32_t *)malloc(ts * (size_t)dim * 4 + 16);
  orc_hashtable_reset(t, 1);
  return t;
}
void orc_hashtable_destroy(orc_hashtable *t) {
  if (!t) return;
  free(t->keys); free(t->indices); free(t->status); free(t->activeKeys); free(t);
}
/* ResetHashTable, HashTable.hpp:212-228 */
void orc_hashtable_reset(orc_hashtable *t, int clearCnt) {
  for (size_t e = 0; e < (size_t)t->tableSize; ++e) {
    for (int d = 0; d < t->dim; ++d) t->keys[e * (size_t)t->dim + d] = INT_MAX;
    t->indices[e] = -1;
    t->status[e] = -1;
  }
  if (clearCnt) t->cnt = 0;
}
static int key_eq(const int32_t *a, const int32_t *b, int dim) {
  for (int d = 0; d < dim; ++d) if (a[d] != b[d]) return 0;
  return 1;
}
static int key_empty(const int32_t *a, int dim) {
  for (int d = 0; d < dim; ++d) if (a[d] != INT_MAX) return 0;
  return 1;
}
/* locate the slot of `key`, claiming the first empty slot of its probe chain: the loop of :357-363 with atomicKeyCAS
   degenerated to "read; write if empty".  *created = 1 when the slot was empty. */
static int32_t find_or_claim(orc_hashtable *t, const int32_t *key, int *created) {
  int32_t e = home(t, key);
  for (int32_t visited = 0; visited < t->tableSize; ++visited) {
    int32_t *slot = t->keys + (size_t)e * (size_t)t->dim;
    if (key_empty(slot, t->dim)) {
      memcpy(slot, key, (size_t)t->dim * 4);
      *created = 1;
      return e;
    }
    if (key_eq(slot, key, t->dim)) { *created = 0; return e; }
    e = (e + 127) % t->tableSize;
  }
  *created = 0;
  return -1;
}
/* insert(key), :376-397 */
int32_t orc_hashtable_insert(orc_hashtable *t, const int32_t *key) {
  if (t->tableSize <= 0) return INT32_MIN;
  int created = 0;
  int32_t e = find_or_claim(t, key, &created);
  if (e < 0) return INT32_MIN;
  if (!created) return -1;
  int32_t localno = t->cnt++;
  t->indices[e] = localno;
  memcpy(t->activeKeys + (size_t)localno * (size_t)t->dim, key, (size_t)t->dim * 4);
  return localno;
}
/* insert(key, id), :423-443 */
int orc_hashtable_insert_id(orc_hashtable *t, const int32_t *key, int32_t id) {
  if (t->tableSize <= 0) return 0;
  int created = 0;
  int32_t e = find_or_claim(t, key, &created);
  if (e < 0 || !created) return 0;
  t->indices[e] = id;
  return 1;
}
/* query / entry, :445-470; the wrap test is >= (the reference's `>` reads one slot past the end when
   hashedentry + 127 == tableSize: a defect SURVEY.md 8c says not to replicate) */
static int32_t lookup(const orc_hashtable *t, const int32_t *key, int wantEntry) {
  if (t->tableSize <= 0) return -1;
  int32_t e = home(t, key);
  for (int32_t visited = 0; visited < t->tableSize; ++visited) {
    if (key_eq(t->keys + (size_t)e * (size_t)t->dim, key, t->dim)) return wantEntry ? e : t->indices[e];
    if (t->indices[e] == -1) return -1;
    e += 127;
    if (e >= t->tableSize) e %= t->tableSize;
  }
  return -1;
}
int32_t orc_hashtable_query(const orc_hashtable *t, const int32_t *key) { return lookup(t, key, 0); }
int32_t orc_hashtable_entry(const orc_hashtable *t, const int32_t *key) { return lookup(t, key, 1); }
void orc_hashtable_insert_many(orc_hashtable *t, const int32_t *keys, size_t n, int32_t *ret) {
  for (size_t i = 0; i < n; ++i) {
    int32_t r = orc_hashtable_insert(t, keys + i * (size_t)t->dim);
    if (ret) ret[i] = r;
  }
}
void orc_hashtable_query_many(const orc_hashtable *t, const int32_t *keys, size_t n, int32_t *ret) {
  for (size_t i = 0; i < n; ++i) ret[i] = orc_hashtable_query(t, keys + i * (size_t)t->dim);
}
int32_t orc_hashtable_size(const orc_hashtable *t) { return t->cnt; }
int32_t orc_hashtable_get_table_size(const orc_hashtable *t) { return t->tableSize; }
const int32_t *orc_hashtable_active_keys(const orc_hashtable *t) { return t->activeKeys; }
const int32_t *orc_hashtable_keys(const orc_hashtable *t) { return t->keys; }
const int32_t *orc_hashtable_indices(const orc_hashtable *t) { return t->indices; }

/* resize, :281-292: grow, reset (cnt kept), re-insert activeKeys[i] with id i */
void orc_hashtable_resize(orc_hashtable *t, size_t nExpected) {
  size_t ns = orc_hashtable_table_size(nExpected);
  if (ns <= (size_t)t->tableSize) return;
  free(t->keys); free(t->indices); free(t->status);
  t->tableSize = (int32_t)ns;
  t->keys = (int32_t *)malloc(ns * (size_t)t->dim * 4 + 16);
  t->indices = (int32_t *)malloc(ns * 4 + 16);
  t->status = (int32_t *)malloc(ns * 4 + 16);
  t->activeKeys = (int32_t *)realloc(t->activeKeys, ns * (size_t)t->dim * 4 + 16);
  orc_hashtable_reset(t, 0);
  for (int32_t i = 0; i < t->cnt; ++i) orc_hashtable_insert_id(t, t->activeKeys + (size_t)i * (size_t)t->dim, i);
}
/* preserve, :258-279 */
void orc_hashtable_preserve(orc_hashtable *t, size_t nExpected) {
  int32_t numEntries = t->cnt;
  if (nExpected == (size_t)numEntries) return;
  t->cnt = (int32_t)nExpected;
  size_t ns = orc_hashtable_table_size(nExpected);
  if (ns > (size_t)t->tableSize) {
    free(t->keys); free(t->indices); free(t->status);
    t->tableSize = (int32_t)ns;
    t->keys = (int32_t *)malloc(ns * (size_t)t->dim * 4 + 16);
    t->indices = (int32_t *)malloc(ns * 4 + 16);
    t->status = (int32_t *)malloc(ns * 4 + 16);
    t->activeKeys = (int32_t *)realloc(t->activeKeys, ns * (size_t)t->dim * 4 + 16);
    orc_hashtable_reset(t, 0);
  } else {
    orc_hashtable_reset(t, 0); /* RemoveHashTableEntries over every entry == an empty table */
  }
  int32_t keep = numEntries < (int32_t)nExpected ? numEntries : (int32_t)nExpected;
  for (int32_t i = 0; i < keep; ++i) orc_hashtable_insert_id(t, t->activeKeys + (size_t)i * (size_t)t->dim, i);
}

/* index_buckets_for_particles, simulation/particle/Query.tpp:9-58 under the sequential policy: cells = floor(x/dx +
   displacement) (ComputeSparsity with blockLen 1, offset 0), counts (SpatiallyCount), offsets = exclusive scan, indices
   filled in particle order (SpatiallyDistribute) => ascending ids inside a bucket.  Arrays are malloc'ed; caller frees
   with orc_index_buckets_free. */
orc_hashtable *orc_index_buckets_for_particles(const float *pos, size_t n, float dx, float displacement, size_t expectedCells,
                                               int32_t **countsOut, int32_t **offsetsOut, int32_t **indicesOut) {
  orc_hashtable *t = orc_hashtable_create(3, expectedCells ? expectedCells : n);
  const float dxinv = 1.0f / dx;
  int32_t *cellOf = (int32_t *)malloc((n + 1) * 4);
  for (size_t i = 0; i < n; ++i) {
    int32_t c[3];
    for (int d = 0; d < 3; ++d) c[d] = (int32_t)floorf(pos[3 * i + d] * dxinv + displacement);
    orc_hashtable_insert(t, c);
    cellOf[i] = 0;
  }
  size_t numCells = (size_t)t->cnt + 1;
  int32_t *counts = (int32_t *)calloc(numCells, 4), *offsets = (int32_t *)calloc(numCells, 4), *fill = (int32_t *)calloc(numCells, 4);
  int32_t *indices = (int32_t *)malloc((n + 1) * 4);
  for (size_t i = 0; i < n; ++i) {
    int32_t c[3];
    for (int d = 0; d < 3; ++d) c[d] = (int32_t)floorf(pos[3 * i + d] * dxinv + displacement);
    cellOf[i] = orc_hashtable_query(t, c);
    counts[cellOf[i]]++;
  }
  int32_t run = 0;
  for (size_t k = 0; k < numCells; ++k) { offsets[k] = run; run += counts[k]; }
  for (size_t i = 0; i < n; ++i) indices[offsets[cellOf[i]] + fill[cellOf[i]]++] = (int32_t)i;
  free(cellOf); free(fill);
  *countsOut = counts; *offsetsOut = offsets; *indicesOut = indices;
  return t;
}
void orc_index_buckets_free(int32_t *counts, int32_t *offsets, int32_t *indices) { free(counts); free(offsets); free(indices); }
