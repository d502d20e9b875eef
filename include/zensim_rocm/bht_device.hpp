// bht_device.hpp -- device view + probe/insert protocol of zs::bht<int, dim, int, B> (dim 1-4, B 16|32) for gfx950.
//
// Table layout is byte-identical to the reference (container/Bht.hpp:86-112): `keys` holds
// next_2pow(dim) ints per slot (unused ints stay 0x3f3f3f3f), `indices`, `status` (all -1; this
// implementation never needs the per-slot spin lock of Bht.hpp:821-834,885-894), `activeKeys`
// [tableSize][dim], `cnt`, `success`.  Hashing: universal_hash per component folded with hash_combine
// (py_interop/HashUtils.hpp:23-43, math/Hash.hpp:19-28), three functions seeded from std::mt19937(2).
//
// Insert protocol (replaces atomicSwitchIfEqual / atomicLoad, Bht.hpp:773-895):
//   dim 1: 32-bit CAS sentinel -> key.          dim 2: 64-bit CAS sentinel -> key.
//   dim 3: 16-byte slot {x, y, z, pad}.  The pad word doubles as the claim word:
//          EMPTY (S,S,S,S) -CAS64 on {z,pad}-> (S,S,S,LOCK) -store64 {x,y}, drain, store64 {z,S}-> FULL.
//          Probes take ONE 16-byte agent-scope load per slot (a naturally aligned dwordx4 is served by
//          a single L2 line access; observed untorn on gfx950, see MI355X guide G16/R2) and see exactly
//          one of those states: pad == LOCK means "being written, look again".  No lock is taken on
//          the probe path, whereas the reference locks on every probe of a 3-D key.
//   dim 4: the 16-byte slot has no spare word, so writers serialise on status[slot] (-1 -> -2, the reference's own
//          spin-lock word, Bht.hpp:821-834): lock, re-check emptiness, store {x,y} and {z,w}, drain, unlock.  Probes still
//          take no lock: a 16-byte load that shows one 8-byte half equal to the sentinel and the other not is either a
//          half-written slot or a genuine key with sentinel words; the probe then reads status (issued after the key load
//          has returned): -2 => busy, -1 => the writer has drained, a second key load is complete.
#pragma once
#include <hip/hip_runtime.h>

namespace zsr {

constexpr int BHT_BUCKET = 16;                 // default bucket size; BhtDev::bucket carries the actual one (16 or 32)
constexpr int BHT_SENT = 0x3f3f3f3f;           // key sentinel bytes (Bht.hpp:108-112,126-131)
constexpr int BHT_LOCK = (int)0x80000001;      // transient value of the pad word while a slot is written
constexpr unsigned BHT_PRIME = 4294967291u;    // HashUtils.hpp:12
constexpr int BHT_FAIL = (int)0x80000000;      // failure_token_v = numeric lowest (Bht.hpp:136)

struct BhtDev {
  int *keys;
  int *indices;
  int *status;
  int *activeKeys;
  int *cnt;
  int *success;
  unsigned tableSize, numBuckets;
  unsigned hf[6];
  unsigned bucket;  // B: slots per bucket; threshold = B - 2 (Bht.hpp:34)
};

__host__ __device__ __forceinline__ unsigned bht_hash1(unsigned hx, unsigned hy, int k) {
  return (unsigned)(((hx ^ (unsigned)k) + hy) % BHT_PRIME);
}
template <int DIM> __host__ __device__ __forceinline__ unsigned bht_hash(unsigned hx, unsigned hy, const int *k) {
  unsigned ret = bht_hash1(hx, hy, k[0]);
#pragma unroll
  for (int d = 1; d < DIM; ++d) {
    unsigned v = bht_hash1(hx, hy, k[d]);
    ret ^= (v + 0x9e3779b9u + (ret << 6) + (ret >> 2));
  }
  return ret;
}
template <int DIM> constexpr int bht_kstride() { return DIM == 1 ? 1 : (DIM == 2 ? 2 : 4); }

typedef int bht_int4 __attribute__((ext_vector_type(4)));

// agent-scope 16-byte snapshot of one slot (sc1: served by L2/fabric, never by this CU's L1)
__device__ __forceinline__ bht_int4 bht_load_slot16(const int *p) {
  bht_int4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// returns: 1 key found (slotKey == key), 0 slot empty, -1 other key, 2 busy (retry)
template <int DIM> __device__ __forceinline__ int bht_probe(const BhtDev &t, unsigned si, const int *key) {
  const int *slot = t.keys + (size_t)si * bht_kstride<DIM>();
  if constexpr (DIM == 1) {
    int k = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return k == key[0] ? 1 : (k == BHT_SENT ? 0 : -1);
  } else if constexpr (DIM == 2) {
    unsigned long long v = __hip_atomic_load((const unsigned long long *)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int k0 = (int)(unsigned)v, k1 = (int)(unsigned)(v >> 32);
    if (k0 == key[0] && k1 == key[1]) return 1;
    return (k0 == BHT_SENT && k1 == BHT_SENT) ? 0 : -1;
  } else if constexpr (DIM == 3) {
    bht_int4 v = bht_load_slot16(slot);
    if (v.w == BHT_LOCK) return 2;
    if (v.x == key[0] && v.y == key[1] && v.z == key[2]) return 1;
    return (v.x == BHT_SENT && v.y == BHT_SENT && v.z == BHT_SENT) ? 0 : -1;
  } else {
    bht_int4 v = bht_load_slot16(slot);
    const bool lo = v.x == BHT_SENT && v.y == BHT_SENT, hi = v.z == BHT_SENT && v.w == BHT_SENT;
    if (lo != hi) {  // possibly half written: status decides (the key load above has completed)
      if (__hip_atomic_load(t.status + si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1) return 2;
      v = bht_load_slot16(slot);
    }
    if (v.x == key[0] && v.y == key[1] && v.z == key[2] && v.w == key[3]) return 1;
    return (v.x == BHT_SENT && v.y == BHT_SENT && v.z == BHT_SENT && v.w == BHT_SENT) ? 0 : -1;
  }
}

// try to turn an empty slot into `key`; true on success
template <int DIM> __device__ __forceinline__ bool bht_claim(const BhtDev &t, unsigned si, const int *key) {
  int *slot = t.keys + (size_t)si * bht_kstride<DIM>();
  if constexpr (DIM == 1) {
    return atomicCAS(slot, BHT_SENT, key[0]) == BHT_SENT;
  } else if constexpr (DIM == 2) {
    const unsigned long long sent = ((unsigned long long)(unsigned)BHT_SENT << 32) | (unsigned)BHT_SENT;
    const unsigned long long want = ((unsigned long long)(unsigned)key[1] << 32) | (unsigned)key[0];
    return atomicCAS((unsigned long long *)slot, sent, want) == sent;
  } else if constexpr (DIM == 3) {
    unsigned long long *h0 = (unsigned long long *)slot, *h1 = h0 + 1;
    const unsigned long long sent = ((unsigned long long)(unsigned)BHT_SENT << 32) | (unsigned)BHT_SENT;
    const unsigned long long locked = ((unsigned long long)(unsigned)BHT_LOCK << 32) | (unsigned)BHT_SENT;
    if (atomicCAS(h1, sent, locked) != sent) return false;
    // {z,pad} == sentinel does not prove emptiness when a stored key has z == 0x3f3f3f3f: re-check {x,y}
    if (__hip_atomic_load(h0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sent) {
      __hip_atomic_store(h1, sent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __hip_atomic_store(h0, ((unsigned long long)(unsigned)key[1] << 32) | (unsigned)key[0], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // {x,y} is at the coherence point before {z,pad} unlocks
    __hip_atomic_store(h1, ((unsigned long long)(unsigned)BHT_SENT << 32) | (unsigned)key[2], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    return true;
  } else {
    unsigned long long *h0 = (unsigned long long *)slot, *h1 = h0 + 1;
    const unsigned long long sent = ((unsigned long long)(unsigned)BHT_SENT << 32) | (unsigned)BHT_SENT;
    if (atomicCAS(t.status + si, -1, -2) != -1) return false;  // another writer holds the slot: re-examine
    const bool empty = __hip_atomic_load(h0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sent
                       && __hip_atomic_load(h1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sent;
    if (empty) {
      __hip_atomic_store(h0, ((unsigned long long)(unsigned)key[1] << 32) | (unsigned)key[0], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(h1, ((unsigned long long)(unsigned)key[3] << 32) | (unsigned)key[2], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the key is at the coherence point before the slot unlocks
    __hip_atomic_store(t.status + si, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return empty;
  }
}

// first half of BHTView::insert (Bht.hpp:490-516): find `key` or claim a slot for it.
// returns slot >= 0 when THIS thread claimed the slot, -1 when the key is already present, BHT_FAIL on overflow.
template <int DIM> __device__ __forceinline__ int bht_find_or_claim(const BhtDev &t, const int *key) {
  if (t.numBuckets == 0) return BHT_FAIL;
  const int B = (int)t.bucket;
  int iter = 0, load = 0;
  unsigned bucket = bht_hash<DIM>(t.hf[0], t.hf[1], key) % t.numBuckets * t.bucket;
  while (iter < 3) {
    int st = 0;
    for (; load != B; ++load) {
      st = bht_probe<DIM>(t, bucket + (unsigned)load, key);
      if (st == 2) {  // slot is being written by another wave: look again (no inner spin: lanes of one
        --load;       // wave may depend on each other)
        continue;
      }
      if (st >= 0) break;  // found or empty
    }
    if (load != B && st == 1) return -1;  // sentinel_v: already present
    if (load <= B - 2) {                  // threshold = B - 2 (Bht.hpp:34)
      if (bht_claim<DIM>(t, bucket + (unsigned)load, key)) return (int)(bucket + load);
      // lost the race for this slot: re-examine it (it now holds some key, maybe ours)
    } else {
      ++iter;
      load = 0;
      if (iter == 1) bucket = bht_hash<DIM>(t.hf[2], t.hf[3], key) % t.numBuckets * t.bucket;
      else if (iter == 2) bucket = bht_hash<DIM>(t.hf[4], t.hf[5], key) % t.numBuckets * t.bucket;
      else break;
    }
  }
  *t.success = 0;
  return BHT_FAIL;
}
// second half (Bht.hpp:517-528): record the dense index of a claimed slot
template <int DIM> __device__ __forceinline__ int bht_commit(const BhtDev &t, int slot, const int *key, int no, bool enqueue) {
  t.indices[slot] = no;
  if (enqueue) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) t.activeKeys[(size_t)no * DIM + d] = key[d];
  }
  if ((unsigned)no >= t.tableSize - 20u) {  // proximity guard (Bht.hpp:522-526), u32 wrap-around as in the reference
    *t.success = 0;
    no = BHT_FAIL;
  }
  return no;
}

// BHTView::insert (Bht.hpp:490-542).  insertion_index == -1: take the next dense index from cnt.
template <int DIM>
__device__ __forceinline__ int bht_insert(const BhtDev &t, const int *key, int insertion_index = -1, bool enqueue = true) {
  const int slot = bht_find_or_claim<DIM>(t, key);
  if (slot < 0) return slot;
  int no = insertion_index;
  if (insertion_index == -1) no = (int)atomicAdd((unsigned *)t.cnt, 1u);
  return bht_commit<DIM>(t, slot, key, no, enqueue);
}

// Bulk form for kernels in which EVERY thread of the workgroup calls it (threads without a key pass valid = false):
// the dense indices of all slots claimed by the workgroup are taken with ONE atomic on cnt (a single device-wide
// counter saturates at ~90 atomics/us on MI355X: 10M distinct keys would spend > 2 ms there even wave-aggregated).
// `smem` = 2 + blockDim/64 unsigned of LDS.
// second half of the bulk forms: the dense indices of all slots claimed by the workgroup with ONE atomic on cnt; `slot` = what
// bht_find_or_claim / bht_tile_find_or_claim returned for this thread's key (-1 / BHT_FAIL: nothing to commit)
template <int DIM> __device__ __forceinline__ int bht_commit_block(const BhtDev &t, const int *key, int slot, unsigned *smem) {
  const bool won = slot >= 0;
  const unsigned long long m = __ballot(won);
  const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
  if (lane == 0) smem[2 + w] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (int i = 0; i < nw; ++i) {
      const unsigned c = smem[2 + i];
      smem[2 + i] = tot;
      tot += c;
    }
    smem[0] = tot ? atomicAdd((unsigned *)t.cnt, tot) : 0u;
  }
  __syncthreads();
  int ret = slot;  // -1 (present) or BHT_FAIL
  if (won) ret = bht_commit<DIM>(t, slot, key, (int)(smem[0] + smem[2 + w] + (unsigned)__popcll(m & ((1ull << lane) - 1ull))), true);
  __syncthreads();
  return ret;
}
template <int DIM> __device__ __forceinline__ int bht_insert_block(const BhtDev &t, const int *key, bool valid, unsigned *smem) {
  return bht_commit_block<DIM>(t, key, valid ? bht_find_or_claim<DIM>(t, key) : -1, smem);
}

// ---- cooperative forms (BHTView::tile_insert / tile_query, Bht.hpp:547-608, 703-736): the lanes of a tile carry the SAME key and
// examine one bucket together -- lane r takes slot r (16-byte / 8-byte / 4-byte agent-scope load: a 16-slot bucket of 3-D keys is one
// 256-byte request), ballots find a match and the number of occupied slots, rank 0 claims slot [load] (buckets fill from slot 0, as
// in the reference: `load` = popc of the non-empty slots is the first empty one) and a lost race re-reads the bucket.
// Tile: thread_rank(), size() / num_threads(), ballot(pred) with bit r = rank r, any(pred), shfl(v, srcRank) -- a cooperative-groups
// thread_block_tile<B> or BhtWaveTile below.  A tile narrower than the bucket walks it in pieces of its own width.
struct BhtWaveTile {  // B consecutive lanes of a wavefront, B a power of two <= 64
  int base, width;
  __device__ __forceinline__ BhtWaveTile(int widthPow2) : base((int)(threadIdx.x & 63) & ~(widthPow2 - 1)), width(widthPow2) {}
  __device__ __forceinline__ int thread_rank() const { return (int)(threadIdx.x & 63) - base; }
  __device__ __forceinline__ int size() const { return width; }
  __device__ __forceinline__ unsigned long long ballot(int pred) const {
    return (__ballot(pred) >> base) & (width == 64 ? ~0ull : ((1ull << width) - 1ull));
  }
  __device__ __forceinline__ int any(int pred) const { return ballot(pred) != 0ull; }
  template <class T> __device__ __forceinline__ T shfl(T v, int srcRank) const { return __shfl(v, base + srcRank); }
};
// returns the claimed slot (>= 0, same value on every lane; rank 0 made the claim), -1 when the key is present, BHT_FAIL on overflow
template <int DIM, class Tile> __device__ __forceinline__ int bht_tile_find_or_claim(const BhtDev &t, const int *key, Tile &tile) {
  if (t.numBuckets == 0) return BHT_FAIL;
  const int B = (int)t.bucket, rank = (int)tile.thread_rank(), ts = (int)tile.size();
  unsigned bucket = bht_hash<DIM>(t.hf[0], t.hf[1], key) % t.numBuckets * t.bucket;
  for (int iter = 0; iter < 3;) {
    int load = 0;
    bool found = false, busy = false;
    for (int off = 0; off < B; off += ts) {
      const bool mine = off + rank < B;
      const int st = mine ? bht_probe<DIM>(t, bucket + (unsigned)(off + rank), key) : 0;
      found = found || tile.any(mine && st == 1);
      busy = busy || tile.any(mine && st == 2);
      load += __popcll(tile.ballot(mine && st != 0));
    }
    if (busy) continue;  // a slot is being written (it may become this key): look again
    if (found) return -1;   // sentinel_v: already present
    if (load <= B - 2) {    // threshold = B - 2 (Bht.hpp:34)
      int ok = 0;
      if (rank == 0) ok = bht_claim<DIM>(t, bucket + (unsigned)load, key) ? 1 : 0;
      if (tile.shfl(ok, 0)) return (int)(bucket + (unsigned)load);
      // by the time, position [load] was already filled (Bht.hpp:571): re-read the bucket
    } else {
      ++iter;
      if (iter == 1) bucket = bht_hash<DIM>(t.hf[2], t.hf[3], key) % t.numBuckets * t.bucket;
      else if (iter == 2) bucket = bht_hash<DIM>(t.hf[4], t.hf[5], key) % t.numBuckets * t.bucket;
    }
  }
  if (rank == 0) *t.success = 0;
  return BHT_FAIL;
}
// BHTView::tile_insert (Bht.hpp:547-608): every lane returns the index (only the inserting call returns a fresh one), -1 if present
template <int DIM, class Tile>
__device__ __forceinline__ int bht_tile_insert(const BhtDev &t, const int *key, Tile &tile, int insertion_index = -1, bool enqueue = true) {
  const int slot = bht_tile_find_or_claim<DIM>(t, key, tile);
  if (slot < 0) return slot;
  int no = insertion_index;
  if (tile.thread_rank() == 0) {
    if (insertion_index == -1) no = (int)atomicAdd((unsigned *)t.cnt, 1u);
    no = bht_commit<DIM>(t, slot, key, no, enqueue);
  }
  return tile.shfl(no, 0);
}
// Bulk form of the cooperative insert for kernels in which every thread of the workgroup holds one key (threads without: valid =
// false): the wave's 64 keys are taken in B rounds by its 64 / B tiles -- in round r tile j inserts the key of lane r (64 / B) + j --
// and each lane gets the claim of its own key back.  Follow with bht_commit_block.
template <int DIM> __device__ __forceinline__ int bht_find_or_claim_tiled(const BhtDev &t, const int *key, bool valid) {
  const int B = (int)t.bucket, lane = (int)(threadIdx.x & 63), per = 64 / B;
  BhtWaveTile tile(B);
  int mine = -1;
#pragma unroll 1
  for (int r = 0; r < B; ++r) {
    const int owner = r * per + lane / B;
    int k[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) k[d] = __shfl(key[d], owner);
    const int kv = __shfl((int)valid, owner);
    int s = -1;
    if (kv) s = bht_tile_find_or_claim<DIM>(t, k, tile);
    const int got = __shfl(s, (lane % per) * B);  // (every lane of a tile holds the tile's answer)
    if (lane / per == r) mine = got;
  }
  return mine;
}
// BHTView::tile_query (Bht.hpp:703-736): plain loads, table must be quiescent
template <int DIM, bool RETSLOT = false, class Tile> __device__ __forceinline__ int bht_tile_query(const BhtDev &t, const int *key, Tile &tile) {
  if (t.numBuckets == 0) return RETSLOT ? 0x7fffffff : -1;
  constexpr int KS = bht_kstride<DIM>();
  const int B = (int)t.bucket, rank = (int)tile.thread_rank(), ts = (int)tile.size();
  unsigned bucket = bht_hash<DIM>(t.hf[0], t.hf[1], key) % t.numBuckets * t.bucket;
  for (int iter = 0; iter < 3;) {
    for (int off = 0; off < B; off += ts) {
      bool eq = false;
      if (off + rank < B) {
        const int *s = t.keys + (size_t)(bucket + (unsigned)(off + rank)) * KS;
        if constexpr (DIM >= 3) {
          const bht_int4 v = *reinterpret_cast<const bht_int4 *>(s);
          eq = v.x == key[0] && v.y == key[1] && v.z == key[2] && (DIM == 3 || v.w == key[DIM - 1]);
        } else if constexpr (DIM == 2) {
          eq = s[0] == key[0] && s[1] == key[1];
        } else
          eq = s[0] == key[0];
      }
      const unsigned long long m = tile.ballot(eq);
      if (m) {
        const int loc = off + __ffsll((long long)m) - 1;
        return RETSLOT ? (int)(bucket + (unsigned)loc) : t.indices[bucket + (unsigned)loc];
      }
    }
    ++iter;
    if (iter == 1) bucket = bht_hash<DIM>(t.hf[2], t.hf[3], key) % t.numBuckets * t.bucket;
    else if (iter == 2) bucket = bht_hash<DIM>(t.hf[4], t.hf[5], key) % t.numBuckets * t.bucket;
  }
  return RETSLOT ? 0x7fffffff : -1;
}

// BHTView::query (Bht.hpp:667-698): plain loads, table must be quiescent.  RETSLOT: slot instead of index.
template <int DIM, bool RETSLOT = false> __device__ __forceinline__ int bht_query(const BhtDev &t, const int *key) {
  if (t.numBuckets == 0) return RETSLOT ? 0x7fffffff : -1;
  constexpr int KS = bht_kstride<DIM>();
  const int B = (int)t.bucket;
  unsigned bucket = bht_hash<DIM>(t.hf[0], t.hf[1], key) % t.numBuckets * t.bucket;
  for (int iter = 0; iter < 3;) {
    for (int loc = 0; loc != B; ++loc) {
      const int *s = t.keys + (size_t)(bucket + loc) * KS;
      bool eq;
      if constexpr (DIM == 4) {
        bht_int4 v = *reinterpret_cast<const bht_int4 *>(s);
        eq = v.x == key[0] && v.y == key[1] && v.z == key[2] && v.w == key[3];
      } else if constexpr (DIM == 3) {
        bht_int4 v = *reinterpret_cast<const bht_int4 *>(s);
        eq = v.x == key[0] && v.y == key[1] && v.z == key[2];
      } else if constexpr (DIM == 2) {
        eq = s[0] == key[0] && s[1] == key[1];
      } else
        eq = s[0] == key[0];
      if (eq) return RETSLOT ? (int)(bucket + loc) : t.indices[bucket + loc];
    }
    ++iter;
    if (iter == 1) bucket = bht_hash<DIM>(t.hf[2], t.hf[3], key) % t.numBuckets * t.bucket;
    else if (iter == 2) bucket = bht_hash<DIM>(t.hf[4], t.hf[5], key) % t.numBuckets * t.bucket;
  }
  return RETSLOT ? 0x7fffffff : -1;
}

}  // namespace zsr
