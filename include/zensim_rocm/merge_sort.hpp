// merge_sort.hpp -- stable comparison sort for gfx950 (zs::merge_sort / zs::merge_sort_pair).
//
// Replaces cub::DeviceMergeSort::StableSortKeys/StableSortPairs behind CudaExecutionPolicy::merge_sort(_pair)
// (cuda/execution/ExecutionPolicy.cuh:698-752).  Semantics = the reference's sequential routine
// (execution/ExecutionPolicy.hpp:341-420): stable, in place, ties keep the LEFT element (`!comp(b, a)` takes a).
// A stable sort under a strict weak order has exactly one result, so parity is bit-exact for any comparator.
//
// Shape: (1) tile sort -- a 256-thread workgroup sorts 2048 elements: every lane orders its 8 consecutive elements
// with an odd-even transposition network in registers (adjacent swaps only => stable), then 8 merge-path passes inside
// LDS; (2) log2(#tiles) global passes -- a partition kernel finds the merge-path split of every 2048-element output tile,
// the merge kernel stages the two input pieces in LDS and every lane merges 8 outputs.  All global traffic is coalesced;
// per pass 2*(sizeof(K)+sizeof(V)) bytes per element.  Header-only so that the C++ face can instantiate it with user
// comparators; the C ABI instantiates `less` for int/float/double.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace zs_rocm_ms {

constexpr int MS_BLOCK = 256;
constexpr int MS_ITEMS = 8;
constexpr int MS_TILE = MS_BLOCK * MS_ITEMS;

struct NoVal {};  // value type of keys-only sorts

// Number of elements of run a among the first `diag` outputs of the stable merge of a[0,na) and b[0,nb).
template <class I, class A, class B, class Comp>
__device__ __forceinline__ I merge_path(A a, I na, B b, I nb, I diag, Comp &comp) {
  I lo = diag > nb ? diag - nb : (I)0, hi = diag < na ? diag : na;
  while (lo < hi) {
    const I mid = (lo + hi) >> 1;
    if (!comp(b[diag - 1 - mid], a[mid])) lo = mid + 1;  // a[mid] precedes b[diag-1-mid]
    else hi = mid;
  }
  return lo;
}

// serial stable merge of up to MS_ITEMS outputs from LDS runs s[ai..aend) and s[bi..bend)
template <class K, class V, bool PAIR, class Comp>
__device__ __forceinline__ void serial_merge(const K *s, const V *sv, int ai, int aend, int bi, int bend, int nout,
                                             K (&rk)[MS_ITEMS], V (&rv)[MS_ITEMS], Comp &comp) {
  K ka = ai < aend ? s[ai] : K{}, kb = bi < bend ? s[bi] : K{};
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i) {
    if (i < nout) {
      const bool takeA = bi >= bend || (ai < aend && !comp(kb, ka));
      // (two assignments, not `rk[i] = takeA ? ka : kb`: the select of a struct-typed key kept rk[] in private memory -- 176 B of scratch per
      // lane for a 16-byte key; with plain assignments the array is promoted to registers and no merge kernel uses scratch)
      if (takeA) rk[i] = ka;
      else rk[i] = kb;
      if constexpr (PAIR) rv[i] = sv[takeA ? ai : bi];
      if (takeA) {
        ++ai;
        if (ai < aend) ka = s[ai];
      } else {
        ++bi;
        if (bi < bend) kb = s[bi];
      }
    }
  }
}

template <class K, class V, bool PAIR, class Comp, class KIn, class VIn, class KOut, class VOut>
__global__ __launch_bounds__(MS_BLOCK) void tile_sort_kernel(KIn kin, VIn vin, KOut kout, VOut vout, size_t n, Comp comp) {
  __shared__ K s[MS_TILE];
  __shared__ V sv[PAIR ? MS_TILE : 1];
  const int t = threadIdx.x;
  const size_t tileBase = (size_t)blockIdx.x * MS_TILE;
  const int cnt = (int)(n - tileBase < (size_t)MS_TILE ? n - tileBase : (size_t)MS_TILE);
#pragma unroll
  for (int k = 0; k < MS_ITEMS; ++k) {
    const int i = t + k * MS_BLOCK;
    if (i < cnt) {
      s[i] = kin[tileBase + i];
      if constexpr (PAIR) sv[i] = vin[tileBase + i];
    }
  }
  __syncthreads();
  const int o = t * MS_ITEMS;
  const int m = cnt - o < 0 ? 0 : (cnt - o < MS_ITEMS ? cnt - o : MS_ITEMS);
  K rk[MS_ITEMS];
  V rv[MS_ITEMS];
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i)
    if (i < m) {
      rk[i] = s[o + i];
      if constexpr (PAIR) rv[i] = sv[o + i];
    }
  // odd-even transposition: adjacent exchanges only when strictly out of order => stable
#pragma unroll
  for (int r = 0; r < MS_ITEMS; ++r) {
#pragma unroll
    for (int i = r & 1; i + 1 < MS_ITEMS; i += 2) {
      if (i + 1 < m && comp(rk[i + 1], rk[i])) {
        const K tk = rk[i];
        rk[i] = rk[i + 1];
        rk[i + 1] = tk;
        if constexpr (PAIR) {
          const V tv = rv[i];
          rv[i] = rv[i + 1];
          rv[i + 1] = tv;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i)
    if (i < m) {
      s[o + i] = rk[i];
      if constexpr (PAIR) sv[o + i] = rv[i];
    }
  __syncthreads();
  for (int w = MS_ITEMS; w < MS_TILE; w <<= 1) {
    const int pairStart = o & ~(2 * w - 1);
    const int aBeg = pairStart < cnt ? pairStart : cnt;
    const int aEnd = pairStart + w < cnt ? pairStart + w : cnt;
    const int bEnd = pairStart + 2 * w < cnt ? pairStart + 2 * w : cnt;
    const int na = aEnd - aBeg, nb = bEnd - aEnd;
    int diag = o - pairStart;
    if (diag > na + nb) diag = na + nb;
    const int mp = merge_path<int>(s + aBeg, na, s + aEnd, nb, diag, comp);
    serial_merge<K, V, PAIR>(s, sv, aBeg + mp, aEnd, aEnd + (diag - mp), bEnd, m, rk, rv, comp);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MS_ITEMS; ++i)
      if (i < m) {
        s[o + i] = rk[i];
        if constexpr (PAIR) sv[o + i] = rv[i];
      }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < MS_ITEMS; ++k) {
    const int i = t + k * MS_BLOCK;
    if (i < cnt) {
      kout[tileBase + i] = s[i];
      if constexpr (PAIR) vout[tileBase + i] = sv[i];
    }
  }
}

struct PairGeom {
  size_t aBeg, aEnd, bEnd, diag;
  __device__ __forceinline__ PairGeom(size_t tile, size_t n, size_t w) {
    const size_t o = tile * MS_TILE;
    aBeg = o / (2 * w) * (2 * w);
    aEnd = aBeg + w < n ? aBeg + w : n;
    bEnd = aBeg + 2 * w < n ? aBeg + 2 * w : n;
    diag = o - aBeg;
  }
};

// split[g] = elements of run a consumed before output tile g of its run pair (runs of width w, a multiple of MS_TILE)
template <class K, class Comp>
__global__ void merge_partition_kernel(const K *src, size_t n, size_t w, size_t *split, size_t numTiles, Comp comp) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= numTiles) return;
  const PairGeom G(g, n, w);
  split[g] = merge_path<size_t>(src + G.aBeg, G.aEnd - G.aBeg, src + G.aEnd, G.bEnd - G.aEnd, G.diag, comp);
}

template <class K, class V, bool PAIR, class Comp, class KOut, class VOut>
__global__ __launch_bounds__(MS_BLOCK) void merge_kernel(const K *src, const V *srcv, KOut kout, VOut vout, size_t n, size_t w,
                                                         const size_t *split, Comp comp) {
  __shared__ K s[MS_TILE];
  __shared__ V sv[PAIR ? MS_TILE : 1];
  const int t = threadIdx.x;
  const PairGeom G(blockIdx.x, n, w);
  const size_t na = G.aEnd - G.aBeg, nb = G.bEnd - G.aEnd;
  const size_t diagEnd = G.diag + MS_TILE < na + nb ? G.diag + MS_TILE : na + nb;
  const size_t a0 = split[blockIdx.x], b0 = G.diag - a0;
  const size_t a1 = diagEnd == na + nb ? na : split[blockIdx.x + 1];
  const size_t b1 = diagEnd - a1;
  const int la = (int)(a1 - a0), lb = (int)(b1 - b0), tot = la + lb;
  const K *A = src + G.aBeg + a0, *B = src + G.aEnd + b0;
  const V *Av = srcv + (PAIR ? G.aBeg + a0 : 0), *Bv = srcv + (PAIR ? G.aEnd + b0 : 0);
#pragma unroll
  for (int k = 0; k < MS_ITEMS; ++k) {
    const int i = t + k * MS_BLOCK;
    if (i < la) {
      s[i] = A[i];
      if constexpr (PAIR) sv[i] = Av[i];
    } else if (i < tot) {
      s[i] = B[i - la];
      if constexpr (PAIR) sv[i] = Bv[i - la];
    }
  }
  __syncthreads();
  const int o = t * MS_ITEMS < tot ? t * MS_ITEMS : tot;
  const int m = tot - o < MS_ITEMS ? tot - o : MS_ITEMS;
  K rk[MS_ITEMS];
  V rv[MS_ITEMS];
  const int mp = merge_path<int>(s, la, s + la, lb, o, comp);
  serial_merge<K, V, PAIR>(s, sv, mp, la, la + (o - mp), tot, m, rk, rv, comp);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i)
    if (i < m) {
      s[o + i] = rk[i];
      if constexpr (PAIR) sv[o + i] = rv[i];
    }
  __syncthreads();
  const size_t outBase = G.aBeg + G.diag;
#pragma unroll
  for (int k = 0; k < MS_ITEMS; ++k) {
    const int i = t + k * MS_BLOCK;
    if (i < tot) {
      kout[outBase + i] = s[i];
      if constexpr (PAIR) vout[outBase + i] = sv[i];
    }
  }
}

// bytes of scratch merge_sort_run needs (0 when one tile suffices)
template <class K, class V, bool PAIR> inline size_t scratch_bytes(size_t n) {
  if (n <= (size_t)MS_TILE) return 0;
  const size_t numTiles = (n + MS_TILE - 1) / MS_TILE;
  const size_t al = 256;
  auto up = [&](size_t b) { return (b + al - 1) / al * al; };
  return 2 * up(n * sizeof(K)) + (PAIR ? 2 * up(n * sizeof(V)) : 0) + up((numTiles + 1) * sizeof(size_t));
}

// Sort keys[0,n) (and vals) in place.  `keys`/`vals` are random-access device iterators (`it[i]` readable and
// writable); `scratch` holds scratch_bytes() bytes of device memory.  Everything is enqueued on `stream`.
template <class K, class V, bool PAIR, class KIt, class VIt, class Comp>
inline void merge_sort_run(hipStream_t stream, KIt keys, VIt vals, size_t n, Comp comp, void *scratch) {
  if (n == 0) return;
  const size_t numTiles = (n + MS_TILE - 1) / MS_TILE;
  if (numTiles == 1) {
    hipLaunchKernelGGL((tile_sort_kernel<K, V, PAIR, Comp, KIt, VIt, KIt, VIt>), dim3(1), dim3(MS_BLOCK), 0, stream, keys, vals,
                       keys, vals, n, comp);
    return;
  }
  const size_t al = 256;
  auto up = [&](size_t b) { return (b + al - 1) / al * al; };
  char *p = (char *)scratch;
  K *bk[2];
  V *bv[2] = {nullptr, nullptr};
  bk[0] = (K *)p, p += up(n * sizeof(K));
  bk[1] = (K *)p, p += up(n * sizeof(K));
  if (PAIR) {
    bv[0] = (V *)p, p += up(n * sizeof(V));
    bv[1] = (V *)p, p += up(n * sizeof(V));
  }
  size_t *split = (size_t *)p;
  hipLaunchKernelGGL((tile_sort_kernel<K, V, PAIR, Comp, KIt, VIt, K *, V *>), dim3((unsigned)numTiles), dim3(MS_BLOCK), 0, stream,
                     keys, vals, bk[0], bv[0], n, comp);
  int cur = 0;
  for (size_t w = MS_TILE; w < n; w <<= 1) {
    const bool last = (w << 1) >= n;
    hipLaunchKernelGGL((merge_partition_kernel<K, Comp>), dim3((unsigned)((numTiles + 255) / 256)), dim3(256), 0, stream,
                       (const K *)bk[cur], n, w, split, numTiles, comp);
    if (last)
      hipLaunchKernelGGL((merge_kernel<K, V, PAIR, Comp, KIt, VIt>), dim3((unsigned)numTiles), dim3(MS_BLOCK), 0, stream,
                         (const K *)bk[cur], (const V *)bv[cur], keys, vals, n, w, (const size_t *)split, comp);
    else
      hipLaunchKernelGGL((merge_kernel<K, V, PAIR, Comp, K *, V *>), dim3((unsigned)numTiles), dim3(MS_BLOCK), 0, stream,
                         (const K *)bk[cur], (const V *)bv[cur], bk[cur ^ 1], bv[cur ^ 1], n, w, (const size_t *)split, comp);
    cur ^= 1;
  }
}

}  // namespace zs_rocm_ms
