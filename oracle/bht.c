/*
 * bht.c -- CPU restatement of zs::bht<int, dim, int, B> (dim 1-4, B 16|32) (bucketed 3-hash table after BGHT).
 * TEST INFRASTRUCTURE ONLY (see zpc_oracle.h).  Sequential insertion in input order, i.e. the
 * deterministic SequentialExecutionPolicy behaviour SURVEY.md 8(a) names as the canonical form.
 */
#include "zpc_oracle.h"

#include <stdlib.h>
#include <string.h>

/* bucket size B (16 or 32) is a field; threshold = B - 2 (container/Bht.hpp:34) */
#define KEY_SENTINEL 0x3f3f3f3f      /* bytes 0x3f, Bht.hpp:108-112,126-131 */
#define PRIME 4294967291u            /* container/Bcht.hpp:37, py_interop/HashUtils.hpp:12 */

struct orc_bht {
  int dim, kstride, bucket;
  size_t tableSize, numBuckets;
  int32_t *keys;    /* [tableSize][kstride], kstride = next_2pow(dim) (HashUtils.hpp:49-51) */
  int32_t *indices; /* [tableSize] */
  int32_t *status;  /* [tableSize] all -1 */
  int32_t *activeKeys; /* [tableSize][dim] */
  int32_t cnt, success;
  uint32_t hf[6];
};

/* --- std::mt19937 (needed for the three hash-function seeds, Bht.hpp:165-169) */
typedef struct { uint32_t mt[624]; int idx; } mt19937;
static void mt_seed(mt19937 *g, uint32_t s) {
  g->mt[0] = s;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t mt_next(mt19937 *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}

/* universal_hash(std::mt19937&): Bcht.hpp:39-43 */
void orc_bht_hash_params(uint32_t out[6]) {
  mt19937 g;
  mt_seed(&g, 2u);
  for (int f = 0; f < 3; ++f) {
    uint32_t hx = mt_next(&g) % PRIME;
    if (hx < 1) hx = 1;
    uint32_t hy = mt_next(&g) % PRIME;
    out[2 * f] = hx;
    out[2 * f + 1] = hy;
  }
}

/* scalar: ((hx ^ key) + hy) % prime in u32 arithmetic, HashUtils.hpp:23-25 */
uint32_t orc_universal_hash_i32(uint32_t hx, uint32_t hy, int32_t k) {
  return (uint32_t)(((hx ^ (uint32_t)k) + hy) % PRIME);
}
/* math/Hash.hpp:19-28, 32-bit seed */
static void hash_combine32(uint32_t *seed, uint32_t val) {
  *seed ^= (val + 0x9e3779b9u + (*seed << 6) + (*seed >> 2));
}
/* vec: fold of per-component hashes with hash_combine, HashUtils.hpp:26-43 */
uint32_t orc_universal_hash_vec(uint32_t hx, uint32_t hy, const int32_t *k, int dim) {
  uint32_t ret = orc_universal_hash_i32(hx, hy, k[0]);
  for (int d = 1; d < dim; ++d) hash_combine32(&ret, orc_universal_hash_i32(hx, hy, k[d]));
  return ret;
}

static size_t next_2pow(size_t n) { /* math/bit/Bits.h next_2pow: smallest power of two >= n */
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
/* Bht.hpp:154-158 */
size_t orc_bht_table_size_b(size_t entryCnt, int B) {
  if (entryCnt == 0) return 0;
  size_t n = next_2pow(entryCnt) * 2;
  return n + ((size_t)B - n % (size_t)B);
}
size_t orc_bht_table_size(size_t entryCnt) { return orc_bht_table_size_b(entryCnt, 16); }

static void table_alloc(orc_bht *t, size_t tableSize) {
  t->tableSize = tableSize;
  t->numBuckets = tableSize / (size_t)t->bucket;
  t->keys = (int32_t *)malloc(tableSize * (size_t)t->kstride * 4 + 16);
  t->indices = (int32_t *)malloc(tableSize * 4 + 16);
  t->status = (int32_t *)malloc(tableSize * 4 + 16);
}

orc_bht *orc_bht_create(int dim, size_t nExpected) { return orc_bht_create_b(dim, nExpected, 16); }
orc_bht *orc_bht_create_b(int dim, size_t nExpected, int bucket) {
  orc_bht *t = (orc_bht *)calloc(1, sizeof(orc_bht));
  t->dim = dim;
  t->bucket = bucket;
  t->kstride = (int)next_2pow((size_t)dim);
  table_alloc(t, orc_bht_table_size_b(nExpected, bucket));
  t->activeKeys = (int32_t *)malloc(t->tableSize * (size_t)dim * 4 + 16);
  orc_bht_hash_params(t->hf);
  orc_bht_reset(t, 1);
  return t;
}
void orc_bht_destroy(orc_bht *t) {
  if (!t) return;
  free(t->keys); free(t->indices); free(t->status); free(t->activeKeys); free(t);
}
/* Table::reset + bht::reset, Bht.hpp:107-112,306-318: keys bytes 0x3f, status bytes 0xff */
void orc_bht_reset(orc_bht *t, int clearCnt) {
  memset(t->keys, 0x3f, t->tableSize * (size_t)t->kstride * 4);
  memset(t->status, 0xff, t->tableSize * 4);
  if (clearCnt) t->cnt = 0;
  t->success = 1;
}

static uint32_t hf(const orc_bht *t, int which, const int32_t *key) {
  return orc_universal_hash_vec(t->hf[2 * which], t->hf[2 * which + 1], key, t->dim);
}
static int key_eq(const int32_t *a, const int32_t *b, int dim) {
  for (int d = 0; d < dim; ++d) if (a[d] != b[d]) return 0;
  return 1;
}
static int key_is_sentinel(const int32_t *a, int dim) {
  for (int d = 0; d < dim; ++d) if (a[d] != KEY_SENTINEL) return 0;
  return 1;
}

/* insert with optional fixed index (resize path) : Bht.hpp:612-664 (host), 490-542 (device) */
static int32_t insert_impl(orc_bht *t, const int32_t *key, int32_t insertion_index, int enqueue) {
  if (t->numBuckets == 0) return INT32_MIN;
  const int BUCKET = t->bucket, THRESHOLD = t->bucket - 2;
  int iter = 0, load = 0;
  size_t bucketOffset = (size_t)(hf(t, 0, key) % t->numBuckets) * BUCKET;
  while (iter < 3) {
    for (; load != BUCKET; ++load) {
      const int32_t *cur = t->keys + (bucketOffset + (size_t)load) * (size_t)t->kstride;
      if (key_eq(cur, key, t->dim)) { load = -1; break; }
      if (key_is_sentinel(cur, t->dim)) break;
    }
    if (load < 0) return -1; /* sentinel_v: key already present */
    if (load <= THRESHOLD) {
      int32_t *dst = t->keys + (bucketOffset + (size_t)load) * (size_t)t->kstride;
      for (int d = 0; d < t->dim; ++d) dst[d] = key[d]; /* padding keeps 0x3f bytes (:821-834) */
      int32_t localno = insertion_index;
      if (insertion_index == -1) localno = t->cnt++;
      t->indices[bucketOffset + (size_t)load] = localno;
      if (enqueue) memcpy(t->activeKeys + (size_t)localno * (size_t)t->dim, key, (size_t)t->dim * 4);
      if ((uint32_t)localno >= (uint32_t)((uint32_t)t->tableSize - 20u)) { /* proximity guard :522-526, u32 arithmetic (size_type = make_unsigned<Index>, :30) */
        t->success = 0;
        localno = INT32_MIN;
      }
      return localno;
    }
    ++iter;
    load = 0;
    if (iter == 1) bucketOffset = (size_t)(hf(t, 1, key) % t->numBuckets) * BUCKET;
    else if (iter == 2) bucketOffset = (size_t)(hf(t, 2, key) % t->numBuckets) * BUCKET;
    else break;
  }
  t->success = 0;
  return INT32_MIN; /* failure_token_v */
}
int32_t orc_bht_insert(orc_bht *t, const int32_t *key) { return insert_impl(t, key, -1, 1); }

/* Bht.hpp:667-698 */
int32_t orc_bht_query(const orc_bht *t, const int32_t *key) {
  if (t->numBuckets == 0) return -1;
  const int BUCKET = t->bucket;
  size_t bucketOffset = (size_t)(hf(t, 0, key) % t->numBuckets) * BUCKET;
  for (int iter = 0; iter < 3;) {
    int loc = 0;
    for (; loc != BUCKET; ++loc)
      if (key_eq(t->keys + (bucketOffset + (size_t)loc) * (size_t)t->kstride, key, t->dim)) break;
    if (loc != BUCKET) return t->indices[bucketOffset + (size_t)loc];
    ++iter;
    if (iter == 1) bucketOffset = (size_t)(hf(t, 1, key) % t->numBuckets) * BUCKET;
    else if (iter == 2) bucketOffset = (size_t)(hf(t, 2, key) % t->numBuckets) * BUCKET;
  }
  return -1;
}
void orc_bht_insert_many(orc_bht *t, const int32_t *keys, size_t n, int32_t *ret) {
  for (size_t i = 0; i < n; ++i) {
    int32_t r = orc_bht_insert(t, keys + i * (size_t)t->dim);
    if (ret) ret[i] = r;
  }
}
void orc_bht_query_many(const orc_bht *t, const int32_t *keys, size_t n, int32_t *ret) {
  for (size_t i = 0; i < n; ++i) ret[i] = orc_bht_query(t, keys + i * (size_t)t->dim);
}
int32_t orc_bht_size(const orc_bht *t) { return t->cnt; }
size_t orc_bht_get_table_size(const orc_bht *t) { return t->tableSize; }
int32_t orc_bht_build_success(const orc_bht *t) { return t->success; }
const int32_t *orc_bht_active_keys(const orc_bht *t) { return t->activeKeys; }
const int32_t *orc_bht_keys(const orc_bht *t) { return t->keys; }
const int32_t *orc_bht_indices(const orc_bht *t) { return t->indices; }
const int32_t *orc_bht_status(const orc_bht *t) { return t->status; }
int orc_bht_key_stride(const orc_bht *t) { return t->kstride; }

/* bht::resize, Bht.hpp:320-340: grow, reset, re-insert activeKeys[i] with fixed index i */
void orc_bht_resize(orc_bht *t, size_t newCapacity) {
  size_t ns = orc_bht_table_size_b(newCapacity, t->bucket);
  if (ns <= t->tableSize) return;
  free(t->keys); free(t->indices); free(t->status);
  table_alloc(t, ns);
  t->activeKeys = (int32_t *)realloc(t->activeKeys, ns * (size_t)t->dim * 4 + 16);
  orc_bht_reset(t, 0);
  for (int32_t i = 0; i < t->cnt; ++i) insert_impl(t, t->activeKeys + (size_t)i * (size_t)t->dim, i, 0);
}
