// lbvh.hip -- zs::LBvh<3, int, f32>::build / refit (container/Bvh.hpp:810-1248) and bulk overlap queries for gfx950.
//
// Pipeline of build (reference functor in brackets):
//   whole box      two-level min/max reduction of the boxes padded by 10 eps (the reference issues 6 float atomics per box,
//                  compute_bounding_box Bvh.hpp:11-23,39-84)
//   morton codes   box centre -> unit cube -> 30-bit code  [_build_init_mc_id :177-188]
//   sort           radix_sort_pair<u32 code, i32 id> (onesweep, primitives.hip)
//   topology       one lane per trunk node: Karras' range search + split over the sorted codes  [_build_build_topo :200-287]
//   layout         exclusive_scan of leaf depths -> pre-order node numbers, levels, escape indices
//                  [_build_supp_topo :288-303, _build_reorder_leaf :304-319, _build_reorder_trunk :320-338]
//   refit          bottom-up with one arrival flag per trunk node  [_refit_bottom_up :469-492]
// Every array of the result is a deterministic function of the input (stable sort, integer topology, min/max boxes), so the
// tests compare bit for bit with the CPU oracle.  Built with -ffp-contract=off: the centre/unit-cube arithmetic must round
// exactly like the reference's scalar code for the morton codes to agree.
#include <cfloat>

#include "common.hpp"
#include "../../include/zensim_rocm/lbvh_device.hpp"

namespace zsr {

void exclusive_scan_u32(Launch &L, const unsigned *in, size_t n, unsigned *out);
void radix_sort_pair_u32(Launch &L, const unsigned *kin, const int *vin, unsigned *kout, int *vout, size_t n, int sbit, int ebit);

}  // namespace zsr

// 32-byte node for the bulk query kernels: box, level and leaf id / escape index in one aligned fetch instead of three arrays
struct alignas(32) LbvhPackedNode {
  float lo[3], hi[3];
  int level, aux;
};
struct zs_rocm_lbvh {
  size_t numLeaves = 0, numNodes = 0, capLeaves = 0;
  zsr::AABB3 *orderedBvs = nullptr;
  int *parents = nullptr, *levels = nullptr, *leafInds = nullptr, *auxIndices = nullptr;
  mutable LbvhPackedNode *packed = nullptr;  // refreshed lazily by the query entry points after a build / refit
  mutable size_t packedCap = 0;
  mutable bool packedValid = false;
  // hits of the last self-query count pass (up to LBVH_HIT_CACHE per leaf, hit-major): the fill pass copies them instead of
  // walking the tree a second time
  mutable int *hitCache = nullptr, *hitCounts = nullptr;
  mutable size_t hitCacheLeaves = 0;
  mutable bool hitCacheValid = false;
  zsr::LBvhDev dev() const {
    zsr::LBvhDev d;
    d.orderedBvs = orderedBvs; d.parents = parents; d.levels = levels; d.leafInds = leafInds; d.auxIndices = auxIndices;
    d.numNodes = (int)numNodes;
    return d;
  }
};

namespace zsr {

__device__ __forceinline__ unsigned expand_bits_32(unsigned v) {  // math/bit/Bits.h:84-90
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
__device__ __forceinline__ unsigned count_lz(unsigned x) { return x ? (unsigned)__clz((int)x) : 32u; }

constexpr int BOX_BLOCK = 256;
// per-block min/max of the padded boxes -> partial[block][6]; `final`: fold the partials into out[6]
__global__ __launch_bounds__(BOX_BLOCK) void lbvh_box_reduce_kernel(const AABB3 *bvs, size_t n, float *partial, int pad) {
  __shared__ float sm[BOX_BLOCK / 64][6];
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  const float eps = pad ? 10 * FLT_EPSILON : 0.f;
  for (size_t i = (size_t)blockIdx.x * BOX_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BOX_BLOCK) {
    const AABB3 b = bvs[i];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = fminf(lo[d], b.lo[d] - eps);
      hi[d] = fmaxf(hi[d], b.hi[d] + eps);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fminf(lo[d], shfl_down(lo[d], o));
      hi[d] = fmaxf(hi[d], shfl_down(hi[d], o));
    }
  if (lane_id() == 0)
    for (int d = 0; d < 3; ++d) { sm[wave_id()][d] = lo[d]; sm[wave_id()][3 + d] = hi[d]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float r = sm[0][threadIdx.x];
    for (int w = 1; w < BOX_BLOCK / 64; ++w) r = threadIdx.x < 3 ? fminf(r, sm[w][threadIdx.x]) : fmaxf(r, sm[w][threadIdx.x]);
    partial[(size_t)blockIdx.x * 6 + threadIdx.x] = r;
  }
}
__global__ void lbvh_box_final_kernel(const float *partial, int nblocks, float *out) {  // one wave
  const int lane = threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int b = lane; b < nblocks; b += 64)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = fminf(lo[d], partial[(size_t)b * 6 + d]);
      hi[d] = fmaxf(hi[d], partial[(size_t)b * 6 + 3 + d]);
    }
#pragma unroll
  for (int d = 0; d < 3; ++d)
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fminf(lo[d], shfl_down(lo[d], o));
      hi[d] = fmaxf(hi[d], shfl_down(hi[d], o));
    }
  if (lane == 0)
    for (int d = 0; d < 3; ++d) { out[d] = lo[d]; out[3 + d] = hi[d]; }
}
// morton code of every box centre in the unit cube of the whole box (what zs::LBvh's build computes, Bvh.hpp:177-188)
__global__ __launch_bounds__(256) void lbvh_morton_kernel(const AABB3 *bvs, int n, const float *whole, unsigned *mcs, int *indices) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AABB3 b = bvs[i];
  unsigned code = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float c = (b.lo[d] + b.hi[d]) / 2;            // getBoxCenter
    const float len = whole[3 + d] - whole[d];           // getUniformCoord (BoundingVolumeInterface.hpp:24-31)
    float o = c - whole[d];
    o = o < 0.f ? 0.f : (o > len ? len : o);
    const float u = __fdiv_rn(o, len);
    code |= expand_bits_32((unsigned)(u * 1024.f)) << (2 - d);  // morton_3d_32, Bits.h:122-125
  }
  mcs[i] = code;
  indices[i] = i;
}
// ---------------------------------------------------------------------------------------------------------------- topology + layout
// Built from Karras' construction ("Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012) over
// ONE ordering function instead of per-case code, and laid out by walking left spines DOWN from their heads; the arrays that come
// out (pre-order nodes, parents, levels, escape indices, leaf positions) are those of zs::LBvh (container/Bvh.hpp:86-175), which the
// tests compare bit for bit.
//
// plen(a, b), a < b: length of the common prefix of the sorted codes a and b -- with equal codes ranked ABOVE every real prefix and
// increasing with a (33 + a).  Adjacent equal codes therefore form a strictly increasing sequence, whose Cartesian tree is the
// right-leaning chain zs::LBvh builds over a run of duplicates (Karras' own tie-break, 32 + clz(a ^ b), would balance the run
// instead).  Outside [0, n) the length is -1.
__device__ __forceinline__ int lbvh_plen(const unsigned *codes, int n, int a, int b) {
  if (a < 0 || b < 0 || a >= n || b >= n) return -1;
  const int lo = a < b ? a : b;
  const unsigned x = codes[a] ^ codes[b];
  return x ? __clz((int)x) : 33 + lo;
}
// One lane per internal ("trunk") node k in Karras' numbering: the node sits at one end of its leaf range, on the side of the
// neighbour it shares the longer prefix with.  Outputs: the range [first, last], the parent link of both children, and whether a
// trunk child is a RIGHT child (= the head of a left spine).
__global__ __launch_bounds__(256) void lbvh_ranges_kernel(const unsigned *codes, int nLeaves, int *rangeFirst, int *rangeLast, int *leftTrunk,
                                                          int *trunkParent, int *leafParent, unsigned char *spineHead) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int nTrunk = nLeaves - 1;
  if (k >= nTrunk) return;
  int first, last;
  if (k == 0) {
    first = 0;
    last = nLeaves - 1;
  } else {
    const int toRight = lbvh_plen(codes, nLeaves, k, k + 1), toLeft = lbvh_plen(codes, nLeaves, k - 1, k);
    const int dir = toLeft > toRight ? -1 : 1;  // (two neighbours never share the same length: sorted codes, ranked ties)
    const int floorLen = toLeft > toRight ? toRight : toLeft;  // everything in the range shares MORE than this with k
    // gallop to an upper bound of the range length, then bisect
    int reach = 2;
    while (lbvh_plen(codes, nLeaves, k, k + reach * dir) > floorLen) reach <<= 1;
    int len = 0;
    for (int t = reach >> 1; t > 0; t >>= 1)
      if (lbvh_plen(codes, nLeaves, k, k + (len + t) * dir) > floorLen) len += t;
    const int other = k + len * dir;
    first = dir > 0 ? k : other;
    last = dir > 0 ? other : k;
  }
  rangeFirst[k] = first;
  rangeLast[k] = last;
  // split: the last leaf that shares more with `first` than the whole range does
  const int nodeLen = lbvh_plen(codes, nLeaves, first, last);
  int cut = 0;
  for (int t = (last - first + 1) >> 1;; t = (t + 1) >> 1) {
    if (first + cut + t < last && lbvh_plen(codes, nLeaves, first, first + cut + t) > nodeLen) cut += t;
    if (t <= 1) break;
  }
  // children: [first, split] and [split + 1, last].  A one-leaf child is the leaf itself; a longer left half is trunk node `split`
  // (it sits at the END of its range), a longer right half trunk node `split + 1` (at the START of its range)
  const int split = first + cut;
  leftTrunk[k] = split == first ? -1 : split;
  if (split == first) leafParent[split] = k;
  else trunkParent[split] = k;
  if (split + 1 == last) leafParent[last] = k;
  else {
    trunkParent[split + 1] = k;
    spineHead[split + 1] = 1;  // a right child starts a new left spine
  }
  if (k == 0) {
    trunkParent[0] = -1;
    spineHead[0] = 1;
  }
}
// A left spine = the trunk nodes that share their first leaf, from the head (the root or a right child) down through left children.
// In pre-order they occupy consecutive positions right in front of that leaf.  Pass 1 (count): the head walks its spine and writes its
// length to spineLen[first leaf] (one writer per leaf: no atomics).  Pass 2 (place): with the exclusive scan `slot` of (spineLen + 1)
// the head walks again: the r-th node of the spine gets position slot[first] + r and level (length - r).
template <bool PLACE>
__global__ __launch_bounds__(256) void lbvh_spine_kernel(int nLeaves, const int *rangeFirst, const int *leftTrunk, const unsigned char *spineHead,
                                                         unsigned *spineLen, const unsigned *slot, int *trunkPos, int *levels) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int nTrunk = nLeaves - 1;
  if (k >= nTrunk || !spineHead[k]) return;
  const int first = rangeFirst[k];
  if (!PLACE) {
    unsigned len = 0;
    for (int node = k;;) {
      ++len;
      const int cand = leftTrunk[node];
      if (cand < 0) break;
      node = cand;
    }
    spineLen[first] = len;
  } else {
    const unsigned len = spineLen[first];
    const unsigned base = slot[first];
    unsigned r = 0;
    for (int node = k;; ++r) {
      trunkPos[node] = (int)(base + r);
      levels[base + r] = (int)(len - r);
      const int cand = leftTrunk[node];
      if (cand < 0) break;
      node = cand;
    }
  }
}
// pre-order records: position of every node, its parent's position, escape index (the node that follows the subtree in pre-order =
// the first node of the spine in front of leaf last + 1, i.e. slot[last + 1]) or the primitive index of a leaf
__global__ __launch_bounds__(256) void lbvh_emit_kernel(int nLeaves, const int *rangeLast, const int *trunkParent, const int *leafParent,
                                                        const int *trunkPos, const unsigned *slot, const unsigned *spineLen, const int *sortedPrim,
                                                        int *auxIndices, int *parents, int *levels, int *leafInds) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int nTrunk = nLeaves - 1;
  if (g < nTrunk) {
    const int pos = trunkPos[g];
    const int last = rangeLast[g];
    auxIndices[pos] = last == nTrunk ? -1 : (int)slot[last + 1];
    const int par = trunkParent[g];
    parents[pos] = par < 0 ? -1 : trunkPos[par];
  } else if (g < nTrunk + nLeaves) {
    const int leaf = g - nTrunk;
    const int pos = (int)(slot[leaf] + spineLen[leaf]);
    auxIndices[pos] = sortedPrim[leaf];
    parents[pos] = trunkPos[leafParent[leaf]];
    levels[pos] = 0;
    leafInds[leaf] = pos;
  }
}
__global__ __launch_bounds__(256) void lbvh_spine_depths_kernel(int nLeaves, const unsigned *spineLen, unsigned *depth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nLeaves) depth[i] = spineLen[i] + 1u;
  else if (i == nLeaves) depth[i] = 0u;
}
// agent-scope box accesses (sc1: served at the device coherence point, never from a CU- or XCD-local cache line), so the
// bottom-up walk needs no __threadfence (on gfx950 a device fence writes back / invalidates L2: measured 94 ms per refit
// of 10 M leaves with fences vs the figure in DESIGN.md without)
// 16 + 8 bytes per box (two memory transactions instead of three 8-byte ones; the refit is bound by the number of L2
// transactions).  sc1 = agent scope: served by / written through to L2.  The accesses need not be single-copy atomic: a box is
// read only after its writer has drained its stores and signed in at the parent's flag.
typedef float lbvh_f4 __attribute__((ext_vector_type(4)));
typedef float lbvh_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_box_agent(AABB3 *p, const AABB3 &b) {
  const lbvh_f4 a = {b.lo[0], b.lo[1], b.lo[2], b.hi[0]};
  const lbvh_f2 c = {b.hi[1], b.hi[2]};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx2 %0, %2, off offset:16 sc1" ::"v"(p), "v"(a), "v"(c) : "memory");
}
__device__ __forceinline__ AABB3 load_box_agent(const AABB3 *p) {
  lbvh_f4 a;
  lbvh_f2 c;
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(c)
               : "v"(p)
               : "memory");
  AABB3 b;
  b.lo[0] = a.x; b.lo[1] = a.y; b.lo[2] = a.z; b.hi[0] = a.w; b.hi[1] = c.x; b.hi[2] = c.y;
  return b;
}
// both children in flight together: four loads, one wait
__device__ __forceinline__ void load_box2_agent(const AABB3 *pl, const AABB3 *pr, AABB3 &L, AABB3 &R) {
  lbvh_f4 a, d;
  lbvh_f2 c, e;
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %4, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %5, off sc1\n\tglobal_load_dwordx2 %3, %5, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(c), "=&v"(d), "=&v"(e)
      : "v"(pl), "v"(pr)
      : "memory");
  L.lo[0] = a.x; L.lo[1] = a.y; L.lo[2] = a.z; L.hi[0] = a.w; L.hi[1] = c.x; L.hi[2] = c.y;
  R.lo[0] = d.x; R.lo[1] = d.y; R.lo[2] = d.z; R.hi[0] = d.w; R.hi[1] = e.x; R.hi[2] = e.y;
}
// _refit_bottom_up (Bvh.hpp:469-492): the second lane to arrive at a trunk node merges its children and climbs
__global__ __launch_bounds__(256) void lbvh_refit_kernel(int numLeaves, const AABB3 *primBvs, AABB3 *orderedBvs, const int *auxIndices,
                                                         const int *leafInds, const int *parents, const int *levels, int *flags) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= numLeaves) return;
  int node = leafInds[idx];
  store_box_agent(orderedBvs + node, primBvs[auxIndices[node]]);
  node = parents[node];
  while (node != -1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's box is at the coherence point before it signs in
    if (atomicCAS(&flags[node], 0, 1) == 0) break;     // first to arrive: the sibling will do the merge
    const int lc = node + 1;
    const int rc = levels[lc] ? auxIndices[lc] : lc + 1;
    AABB3 L, R;
    load_box2_agent(orderedBvs + lc, orderedBvs + rc, L, R);
    AABB3 bv;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      bv.lo[d] = fminf(L.lo[d], R.lo[d]);
      bv.hi[d] = fmaxf(L.hi[d], R.hi[d]);
    }
    store_box_agent(orderedBvs + node, bv);
    node = parents[node];
  }
}
// The same refit, windowed (r04).  In the pre-order layout the subtree of node p occupies the positions [p, esc(p)), so a workgroup that
// owns a window [P0, P1) of positions can finish every subtree that lies inside it on its own: boxes, arrival flags and the topology of
// the window live in LDS, and only the nodes whose subtree sticks out of the window (p < P0 or esc(p) > P1: ~1 % of the nodes at 2048
// positions per window) take the global arrival flags and agent-scope box accesses of lbvh_refit_kernel.  A lane that leaves the local
// domain first publishes the box it carries (agent-scope store), then signs in at the parent's global flag like any other lane.
// Same min / max merges in the same pairs: the boxes are bit-identical to the unwindowed kernel's (and the reference's, Bvh.hpp:469-492).
// A lane that leaves the local domain does not walk on inside this kernel (ten or twenty dependent agent-scope round trips would keep its
// whole 1024-thread workgroup resident: measured, half of the kernel's time): it publishes its box and leaves the node in the window's
// hand-over list (LBVH_RL entries per window; a full list falls back to walking here); lbvh_refit_upper_kernel continues from the lists.
constexpr int LBVH_RL = 128;
__device__ __forceinline__ void lbvh_refit_global_walk(int node, AABB3 *orderedBvs, const int *auxIndices, const int *parents, const int *levels,
                                                       int *gflags) {
  while (node != -1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's box is at the coherence point before it signs in
    if (atomicCAS(&gflags[node], 0, 1) == 0) break;     // first to arrive: the sibling will do the merge
    const int lc = node + 1;
    const int rc = levels[lc] ? auxIndices[lc] : lc + 1;
    AABB3 L, R;
    load_box2_agent(orderedBvs + lc, orderedBvs + rc, L, R);
    AABB3 bv;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      bv.lo[d] = fminf(L.lo[d], R.lo[d]);
      bv.hi[d] = fmaxf(L.hi[d], R.hi[d]);
    }
    store_box_agent(orderedBvs + node, bv);
    node = parents[node];
  }
}
__global__ __launch_bounds__(256) void lbvh_refit_upper_kernel(int nWindows, const int *listCount, const int *list, AABB3 *orderedBvs,
                                                               const int *auxIndices, const int *parents, const int *levels, int *gflags) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = idx / LBVH_RL, k = idx - w * LBVH_RL;
  if (w >= nWindows || k >= listCount[w]) return;
  lbvh_refit_global_walk(list[idx], orderedBvs, auxIndices, parents, levels, gflags);  // (the entry is the first node above the window's domain)
}
template <int LBVH_RW, int LBVH_RT>
__global__ __launch_bounds__(LBVH_RT) void lbvh_refit_window_kernel(int numNodes, const AABB3 *primBvs, AABB3 *orderedBvs, const int *auxIndices,
                                                                    const int *parents, const int *levels, int *gflags, int *listCount,
                                                                    int *list) {
  __shared__ float sb[6][LBVH_RW];
  __shared__ int sf[LBVH_RW];       // low two bits: arrivals (2 = merged here); bit 2: box already published by an agent-scope store
  __shared__ int sPar[LBVH_RW], sAux[LBVH_RW];
  __shared__ unsigned char sTrunk[LBVH_RW];
  __shared__ int sList;
  const int t = threadIdx.x;
  if (t == 0) sList = 0;
  const int P0 = blockIdx.x * LBVH_RW, P1 = P0 + LBVH_RW < numNodes ? P0 + LBVH_RW : numNodes, cntW = P1 - P0;
  for (int i = t; i < cntW; i += LBVH_RT) {
    sf[i] = 0;
    sPar[i] = parents[P0 + i];
    sAux[i] = auxIndices[P0 + i];
    sTrunk[i] = levels[P0 + i] != 0;
  }
  __syncthreads();
  // the leaves' boxes first (a random gather from the primitive array: all of a lane's requests in flight together), then the climbs
  for (int i = t; i < cntW; i += LBVH_RT) {
    if (sTrunk[i]) continue;
    const AABB3 b = primBvs[sAux[i]];
#pragma unroll
    for (int d = 0; d < 3; ++d) { sb[d][i] = b.lo[d]; sb[3 + d][i] = b.hi[d]; }
  }
  __syncthreads();
  for (int i = t; i < cntW; i += LBVH_RT) {
    if (sTrunk[i]) continue;  // trunk nodes are reached by climbing
    AABB3 cur;
#pragma unroll
    for (int d = 0; d < 3; ++d) { cur.lo[d] = sb[d][i]; cur.hi[d] = sb[3 + d][i]; }
    int child = P0 + i;
    int node = sPar[i];
    bool waiting = false;  // first to arrive at a local node: the sibling's lane carries on
    while (node >= P0) {   // (node < P0, incl. -1 above the root: not local)
      const int a = sAux[node - P0];
      if ((a < 0 ? numNodes : a) > P1) break;  // its subtree sticks out of the window
      __threadfence_block();                     // the box of `child` is in LDS before this lane signs in
      if ((atomicAdd(&sf[node - P0], 1) & 3) == 0) { waiting = true; break; }
      __threadfence_block();                     // ... and the sibling's box is read after its lane signed in
      const int lc = node + 1 - P0;
      const int rc = (sTrunk[lc] ? sAux[lc] : lc + 1 + P0) - P0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        cur.lo[d] = fminf(sb[d][lc], sb[d][rc]);
        cur.hi[d] = fmaxf(sb[3 + d][lc], sb[3 + d][rc]);
        sb[d][node - P0] = cur.lo[d];
        sb[3 + d][node - P0] = cur.hi[d];
      }
      child = node;
      node = sPar[node - P0];
    }
    if (waiting || node == -1) continue;  // (node == -1: the root was merged here -- the whole tree is one window)

    // leaving the local domain: publish the carried box and hand the walk over
    store_box_agent(orderedBvs + child, cur);
    atomicOr(&sf[child - P0], 4);
    const int slot = atomicAdd(&sList, 1);
    if (slot < LBVH_RL) list[(size_t)blockIdx.x * LBVH_RL + slot] = node;
    else lbvh_refit_global_walk(node, orderedBvs, auxIndices, parents, levels, gflags);
  }
  __syncthreads();
  // the window's finished boxes that nobody published yet: leaves and the trunk nodes merged here
  float *out = reinterpret_cast<float *>(orderedBvs + P0);
  for (int e = t; e < cntW * 6; e += LBVH_RT) {
    const int i = e / 6, c = e - 6 * i;
    const int f = sf[i];
    if ((!sTrunk[i] || (f & 3) == 2) && !(f & 4)) out[e] = sb[c][i];
  }
  if (t == 0) listCount[blockIdx.x] = sList < LBVH_RL ? sList : LBVH_RL;
}
__global__ __launch_bounds__(256) void lbvh_small_kernel(int n, const AABB3 *primBvs, AABB3 *orderedBvs, int *leafInds, int *auxIndices,
                                                         int *parents, int *levels) {
  const int i = threadIdx.x;
  if (i >= n) return;
  orderedBvs[i] = primBvs[i];
  if (leafInds) { leafInds[i] = i; auxIndices[i] = i; parents[i] = 0; levels[i] = 0; }
}
// bulk iter_neighbors: count pass / fill pass
template <bool FILL>
__global__ __launch_bounds__(256) void lbvh_query_kernel(LBvhDev bvh, const AABB3 *queries, size_t nq, int *counts, const int *offsets,
                                                         int *out) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const AABB3 bv = queries[q];
  int c = 0;
  int *dst = FILL ? out + offsets[q] : nullptr;
  bvh.iter_neighbors(bv, [&](int id) {
    if constexpr (FILL) dst[c] = id;
    ++c;
  });
  if (!FILL) counts[q] = c;
}

// self-collision broadphase: thread k takes the k-th leaf in node (Morton) order, so neighbouring lanes walk neighbouring
// subtrees; LBvhView::self_iter_neighbors reports every overlapping unordered pair once (and the leaf itself, skipped here)
template <bool FILL>
__global__ __launch_bounds__(256) void lbvh_self_query_kernel(LBvhDev bvh, int *counts, const int *offsets, int *pairs) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= bvh.numLeaves()) return;
  const int self = bvh.numNodes <= 2 ? k : bvh.auxIndices[bvh.leafInds[k]];
  int c = 0;
  int *dst = FILL ? pairs + 2 * (size_t)offsets[k] : nullptr;
  bvh.self_iter_neighbors(k, [&](int id) {
    if (id == self) return;
    if constexpr (FILL) {
      dst[2 * c] = self;
      dst[2 * c + 1] = id;
    }
    ++c;
  });
  if (!FILL) counts[k] = c;
}

__global__ __launch_bounds__(256) void lbvh_pack_kernel(LBvhDev bvh, LbvhPackedNode *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bvh.numNodes) return;
  const AABB3 b = bvh.orderedBvs[i];
  LbvhPackedNode n;
#pragma unroll
  for (int d = 0; d < 3; ++d) { n.lo[d] = b.lo[d]; n.hi[d] = b.hi[d]; }
  n.level = bvh.numNodes > 2 ? bvh.levels[i] : 0;
  n.aux = bvh.numNodes > 2 ? bvh.auxIndices[i] : i;
  if (bvh.numNodes > 2 && n.level != 0 && n.aux < 0) n.aux = bvh.numNodes;  // escape index behind the last subtree: one past the end
  out[i] = n;
}
// the reference's stack-less walk over packed nodes: at a trunk node an overlap descends to the left child (node + 1), a miss
// follows the escape index; this is the `for (; level; --level, ++node)` spine loop of Bvh.hpp:661-678 unrolled per node
template <class F>
__device__ __forceinline__ void lbvh_walk_packed(const LbvhPackedNode *nodes, int numNodes, int node, const AABB3 &bv, F &&f) {
  if (numNodes <= 2) {
    for (int i = node; i != numNodes; ++i) {
      const LbvhPackedNode n = nodes[i];
      bool ov = true;
#pragma unroll
      for (int d = 0; d < 3; ++d) ov = ov && !(bv.lo[d] > n.hi[d] || bv.hi[d] < n.lo[d]);
      if (ov) f(i);
    }
    return;
  }
  while (node != -1 && node != numNodes) {
    const LbvhPackedNode n = nodes[node];
    bool ov = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) ov = ov && !(bv.lo[d] > n.hi[d] || bv.hi[d] < n.lo[d]);
    if (n.level == 0) {
      if (ov) f(n.aux);
      node++;
    } else
      node = ov ? node + 1 : n.aux;
  }
}
// perm (optional): thread k takes query perm[k] -- the queries in Morton order of their centres, so that the lanes of a wave walk
// neighbouring parts of the tree and their node fetches fall into the same cache lines; results go to the query's own slot
template <bool FILL>
__global__ __launch_bounds__(256) void lbvh_query_packed_kernel(const LbvhPackedNode *nodes, int numNodes, const AABB3 *queries, size_t nq,
                                                                int *counts, const int *offsets, int *out, const int *perm) {
  size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  if (perm) q = (size_t)perm[q];
  const AABB3 bv = queries[q];
  int c = 0;
  int *dst = FILL ? out + offsets[q] : nullptr;
  lbvh_walk_packed(nodes, numNodes, 0, bv, [&](int id) {
    if constexpr (FILL) dst[c] = id;
    ++c;
  });
  if (!FILL) counts[q] = c;
}
constexpr int LBVH_HIT_CACHE = 32;
// FILL = false: count pass; also remembers the first LBVH_HIT_CACHE hits of every leaf (cache[j * numLeaves + k]) and its count.
// FILL = true: leaves whose hits all fit into the cache are copied from it, the others walk again.
template <bool FILL>
__global__ __launch_bounds__(256) void lbvh_self_query_packed_kernel(const LbvhPackedNode *nodes, int numNodes, int numLeaves,
                                                                     const int *leafInds, int *counts, const int *offsets, int *pairs,
                                                                     int *cache, int *cacheCounts, int useCache) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= numLeaves) return;
  const int start = numNodes <= 2 ? k : leafInds[k];
  const LbvhPackedNode me = nodes[start];
  const int self = me.aux;
  int *dst = FILL ? pairs + 2 * (size_t)offsets[k] : nullptr;
  if constexpr (FILL) {
    if (useCache) {
      const int cc = cacheCounts[k];
      if (cc <= LBVH_HIT_CACHE) {
        for (int j = 0; j < cc; ++j) {
          dst[2 * j] = self;
          dst[2 * j + 1] = cache[(size_t)j * numLeaves + k];
        }
        return;
      }
    }
  }
  AABB3 bv;
#pragma unroll
  for (int d = 0; d < 3; ++d) { bv.lo[d] = me.lo[d]; bv.hi[d] = me.hi[d]; }
  int c = 0;
  // numNodes <= 2: the reference starts at leafId + 1 (Bvh.hpp:697-707); otherwise AT the leaf, which reports itself first
  lbvh_walk_packed(nodes, numNodes, numNodes <= 2 ? start + 1 : start, bv, [&](int id) {
    if (id == self) return;
    if constexpr (FILL) {
      dst[2 * c] = self;
      dst[2 * c + 1] = id;
    } else {
      if (c < LBVH_HIT_CACHE) cache[(size_t)c * numLeaves + k] = id;
    }
    ++c;
  });
  if (!FILL) {
    counts[k] = c;
    cacheCounts[k] = c;
  }
}
// The same pass with ONE walk per wave (r04).  The 64 leaves of a wave are neighbours in node (Morton) order, so their walks visit
// largely the same nodes; done lane by lane every visit is a divergent 32-byte fetch (10 M leaves x ~80 nodes: the count pass moved
// ~50 GB through the L2 request path and ran at its pace).  Here the wave walks the UNION of its lanes' walks in pre-order: `cur` is
// wave-uniform (the node is fetched once, by scalar loads), lane j takes part in a visit iff cur == next_j, its own next node.  At a
// trunk node a lane descends (cur + 1) on overlap and escapes (aux) otherwise -- exactly LBvhView::self_iter_neighbors'
// (Bvh.hpp:697-750) per-leaf walk, so every leaf reports the same ids in the same order.  The next node of the wave is the minimum
// of the lanes' next nodes, and needs no reduction: if a lane descends it is cur + 1; otherwise every lane at cur escapes to the same
// E = aux(cur), lanes that escaped earlier wait at some X >= E (cur lies inside the subtree they skipped), and lanes that have not
// started yet wait at their leaves, which come in lane order -- so it is min(E, start of the first lane not started yet).
// The node is fetched through the scalar cache (one s_load_dwordx8 per step).  Measured alternatives (profiles/r04_lbvh.md): node pairs
// per s_load_dwordx16 (3.30 vs 2.95 ms), the vector path with a wave-uniform address (3.84 ms), 2 / 4 / 8 independent walks per wave
// with their fetches issued together (3.37 / 3.51 / 4.47 ms: the pass is bound by the instruction stream of the steps -- ~40 mostly
// scalar, mostly dependent instructions each -- not by the latency of one chain), and the same kernel over external queries sorted by
// the Morton code of their centres (3.0 vs 1.7 ms for 1 M queries with 15.6 hits each: their unions are too long).
// FILL: leaves whose hits all fit the count pass's cache (LBVH_HIT_CACHE = 32: 99.998 % of the leaves of BASELINE config 5) are copied from
// it by the whole wave in output order; the others walk again.
template <bool FILL>
__global__ __launch_bounds__(256) void lbvh_self_query_wave_kernel(const LbvhPackedNode *__restrict__ nodes, int numNodes, int numLeaves,
                                                                   const int *__restrict__ leafInds, int *counts, const int *offsets, int *pairs,
                                                                   int *cache, int *cacheCounts, int useCache) {
  typedef int v8i __attribute__((ext_vector_type(8)));
  constexpr int NONE = 0x7fffffff;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  const bool valid = k < numLeaves;
  const int start = valid ? leafInds[k] : NONE;
  LbvhPackedNode me{};
  if (valid) me = nodes[start];
  const int self = me.aux;
  int *dst = nullptr;
  bool need = valid;
  if constexpr (FILL) {
    const int off = valid ? offsets[k] : 0;
    if (valid) dst = pairs + 2 * (size_t)off;
    if (useCache) {
      // position p of the wave's contiguous output range belongs to the last leaf whose offset is <= p (leaves without hits share their
      // successor's offset); 8-byte stores to consecutive pairs instead of every lane writing its own short run
      const int cc = valid ? cacheCounts[k] : 0;
      need = valid && cc > LBVH_HIT_CACHE;
      const int nv = __popcll(__ballot(valid));  // valid lanes are a prefix of the wave
      // the cooperative copy below needs the wave's ranges back to back in leaf order (offsets = exclusive scan of the counts, which is
      // what every caller of the count pass builds); any other `offsets` (padded, permuted): every leaf copies its own run
      const int prevEnd = __shfl_up(off + cc, 1, 64);
      const bool contiguous = __ballot(valid && lane > 0 && off != prevEnd) == 0ull;
      if (!contiguous) {
        if (valid && cc <= LBVH_HIT_CACHE)
          for (int c2 = 0; c2 < cc; ++c2) {
            dst[2 * c2] = self;
            dst[2 * c2 + 1] = cache[(size_t)c2 * numLeaves + (size_t)k];
          }
      } else if (nv) {
        const int base = __builtin_amdgcn_readlane(off, 0), end = __builtin_amdgcn_readlane(off + cc, nv - 1);
        const int k0 = k - lane;
        for (int p0 = base; p0 < end; p0 += 64) {  // (wave-uniform trip count: the shuffles below need every lane)
          const int pp = p0 + lane;
          int loI = 0, hiI = nv - 1;
#pragma unroll
          for (int it = 0; it < 6; ++it) {
            const int mid = (loI + hiI + 1) >> 1;
            const int om = __shfl(off, mid, 64);
            if (om <= pp) loI = mid;
            else hiI = mid - 1;
          }
          const int oOff = __shfl(off, loI, 64), oCc = __shfl(cc, loI, 64), oSelf = __shfl(self, loI, 64);
          if (pp < end && oCc <= LBVH_HIT_CACHE && pp - oOff < oCc) {
            typedef int i2 __attribute__((ext_vector_type(2)));
            i2 pr;
            pr.x = oSelf;
            pr.y = cache[(size_t)(pp - oOff) * numLeaves + (size_t)(k0 + loI)];
            __builtin_nontemporal_store(pr, reinterpret_cast<i2 *>(pairs + 2 * (size_t)pp));  // the pair list is written once, streamed
          }
        }
      }
    }
  }
  int next = need ? start : NONE;  // the walk starts AT the leaf, which reports itself first (skipped below)
  int c = 0;
  unsigned long long pending = __ballot(need);  // lanes that have not started yet, in lane (= leaf = node) order
  int nextStart = NONE;
  if (pending) nextStart = __builtin_amdgcn_readlane(start, __ffsll((long long)pending) - 1);
  int cur = nextStart;
  while (cur < numNodes) {  // (cur is wave-uniform by construction and the compiler sees it: it lives in an SGPR -- a readfirstlane here forces a VGPR copy)
    if (cur == nextStart) {  // the first pending lane starts here: the next one in line
      pending &= pending - 1;
      nextStart = NONE;
      if (pending) nextStart = __builtin_amdgcn_readlane(start, __ffsll((long long)pending) - 1);
    }
    // the whole node in ONE scalar load (eight dwords); no short-circuit in the overlap test: a second, dependent load group under
    // a branch would double the latency of a step
    // (the scalar unit sets the pace of this loop -- eight waves per SIMD, ~25 scalar instructions a step: a 32-bit byte offset instead
    // of a sign-extended 64-bit index, the escape index of the last subtree stored as numNodes by the pack kernel, and the leaf's hit
    // test under a scalar branch each take instructions off it)
    const v8i raw = *reinterpret_cast<const v8i *>(reinterpret_cast<const char *>(nodes) + ((unsigned)cur << 5));
    const float nlo0 = __int_as_float(raw[0]), nlo1 = __int_as_float(raw[1]), nlo2 = __int_as_float(raw[2]);
    const float nhi0 = __int_as_float(raw[3]), nhi1 = __int_as_float(raw[4]), nhi2 = __int_as_float(raw[5]);
    const int level = raw[6], aux = raw[7];
    const bool active = next == cur;
    // the six interval tests as ONE compare: !(a > b) for every pair  <=>  !(max over the pairs of (a - b) > 0).  Exact: a difference of two
    // distinct floats is never zero (f32 denormals are kept), inf - inf and NaN operands give NaN, which v_max3 drops as the chained
    // comparisons ignore them, and an all-NaN maximum fails `> 0` like every `a > b` did.  Six subtractions and two max3 on the vector
    // unit replace five 64-bit mask ANDs on the scalar unit, which is the unit this loop saturates (one scalar instruction per cycle and CU).
    const float sep = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(me.lo[0] - nhi0, nlo0 - me.hi[0]), __builtin_fmaxf(me.lo[1] - nhi1, nlo1 - me.hi[1])),
                                      __builtin_fmaxf(me.lo[2] - nhi2, nlo2 - me.hi[2]));
    const bool ov = !(sep > 0.f);
    const unsigned long long downMask = __ballot(active) & __ballot(ov);   // (ballots of two compares: their masks, no materialised booleans)
    if (__builtin_amdgcn_readfirstlane(level) == 0) {  // a leaf (wave-uniform branch): report it, continue at cur + 1
      if (active && ov && aux != self) {
        if constexpr (FILL) {
          dst[2 * c] = self;
          dst[2 * c + 1] = aux;
        } else {
          if (c < LBVH_HIT_CACHE) cache[(size_t)c * numLeaves + k] = aux;
        }
        ++c;
      }
      if (active) next = cur + 1;
      cur = cur + 1;  // (some lane is always at a leaf the wave visits)
    } else {             // a trunk node: descend on overlap, escape otherwise (aux = the escape index; numNodes behind the last subtree)
      const bool down = active && ov;
      if (active) next = down ? cur + 1 : aux;
      if (downMask) cur = cur + 1;
      else cur = aux < nextStart ? aux : nextStart;
    }
  }
  if (!FILL && valid) {
    counts[k] = c;
    cacheCounts[k] = c;
  }
}
// 30-bit Morton code of a query box's centre inside the root box (an ordering only)
__global__ __launch_bounds__(256) void lbvh_query_code_kernel(const LbvhPackedNode *nodes, const AABB3 *queries, int nq, unsigned *codes, int *ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const LbvhPackedNode root = nodes[0];
  const AABB3 q = queries[i];
  unsigned code = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float len = root.hi[d] - root.lo[d];
    float u = len > 0.f ? (0.5f * (q.lo[d] + q.hi[d]) - root.lo[d]) / len : 0.f;
    u = u < 0.f ? 0.f : (u > 0.999999f ? 0.999999f : u);
    code |= expand_bits_32((unsigned)(u * 1024.f)) << (2 - d);
  }
  codes[i] = code;
  ids[i] = i;
}
static const LbvhPackedNode *lbvh_packed(Launch &L, const zs_rocm_lbvh &b) {
  if (b.packedCap < b.numNodes) {
    (void)hipFree(b.packed);
    ZSR_CHECK(hipMalloc((void **)&b.packed, b.numNodes * sizeof(LbvhPackedNode)));
    b.packedCap = b.numNodes;
    b.packedValid = false;
  }
  if (!b.packedValid) {
    hipLaunchKernelGGL(lbvh_pack_kernel, dim3(ceil_div(b.numNodes, 256)), dim3(256), 0, L.stream, b.dev(), b.packed);
    b.packedValid = true;
  }
  return b.packed;
}

static void lbvh_reserve(zs_rocm_lbvh &b, size_t n) {
  if (n <= b.capLeaves) return;
  (void)hipFree(b.orderedBvs); (void)hipFree(b.parents); (void)hipFree(b.levels); (void)hipFree(b.leafInds); (void)hipFree(b.auxIndices);
  const size_t nodes = n > 2 ? 2 * n - 1 : n;
  ZSR_CHECK(hipMalloc((void **)&b.orderedBvs, nodes * sizeof(AABB3)));
  ZSR_CHECK(hipMalloc((void **)&b.parents, nodes * sizeof(int)));
  ZSR_CHECK(hipMalloc((void **)&b.levels, nodes * sizeof(int)));
  ZSR_CHECK(hipMalloc((void **)&b.auxIndices, nodes * sizeof(int)));
  ZSR_CHECK(hipMalloc((void **)&b.leafInds, n * sizeof(int)));
  b.capLeaves = n;
}
static void lbvh_refit_impl(Launch &L, zs_rocm_lbvh &b, const AABB3 *primBvs) {
  const int n = (int)b.numLeaves;
  if (n == 0) return;
  if (n <= 2) {
    hipLaunchKernelGGL(lbvh_small_kernel, dim3(1), dim3(64), 0, L.stream, n, primBvs, b.orderedBvs, (int *)nullptr, (int *)nullptr,
                       (int *)nullptr, (int *)nullptr);
    return;
  }
  int *flags = (int *)L.temp(sizeof(int) * b.numNodes);
  ZSR_CHECK(hipMemsetAsync(flags, 0, sizeof(int) * b.numNodes, L.stream));
  static const bool plain = [] { const char *e = getenv("ZS_ROCM_LBVH_REFIT"); return e && e[0] == 'p'; }();  // A/B runs: the unwindowed kernel
  if (plain)
    hipLaunchKernelGGL(lbvh_refit_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, n, primBvs, b.orderedBvs, b.auxIndices, b.leafInds,
                       b.parents, b.levels, flags);
  else {
    constexpr int RW = 2048, RT = 1024;  // (1024 / 512: the same time; 2048 / 512 and 1024 / 1024: slower -- profiles/r04_lbvh.md)
    const int nWindows = (int)ceil_div(b.numNodes, RW);
    int *listCount = (int *)L.temp(sizeof(int) * nWindows), *list = (int *)L.temp(sizeof(int) * (size_t)nWindows * LBVH_RL);
    hipLaunchKernelGGL((lbvh_refit_window_kernel<RW, RT>), dim3(nWindows), dim3(RT), 0, L.stream, (int)b.numNodes, primBvs, b.orderedBvs,
                       b.auxIndices, b.parents, b.levels, flags, listCount, list);
    hipLaunchKernelGGL(lbvh_refit_upper_kernel, dim3(ceil_div((size_t)nWindows * LBVH_RL, 256)), dim3(256), 0, L.stream, nWindows,
                       (const int *)listCount, (const int *)list, b.orderedBvs, b.auxIndices, b.parents, b.levels, flags);
  }
}

}  // namespace zsr

using namespace zsr;

extern "C" {

zs_rocm_lbvh *zs_rocm_lbvh_create(void) { return new zs_rocm_lbvh; }
void zs_rocm_lbvh_destroy(zs_rocm_lbvh *b) {
  if (!b) return;
  (void)hipFree(b->orderedBvs); (void)hipFree(b->parents); (void)hipFree(b->levels); (void)hipFree(b->leafInds); (void)hipFree(b->auxIndices);
  (void)hipFree(b->packed);
  (void)hipFree(b->hitCache); (void)hipFree(b->hitCounts);
  delete b;
}
size_t zs_rocm_lbvh_num_leaves(const zs_rocm_lbvh *b) { return b->numLeaves; }
size_t zs_rocm_lbvh_num_nodes(const zs_rocm_lbvh *b) { return b->numNodes; }
void zs_rocm_lbvh_get_view(const zs_rocm_lbvh *b, zs_rocm_lbvh_view *v) {
  v->orderedBvs = (float *)b->orderedBvs; v->parents = b->parents; v->levels = b->levels; v->leafInds = b->leafInds;
  v->auxIndices = b->auxIndices; v->numNodes = (int)b->numNodes; v->numLeaves = (int)b->numLeaves;
}

void zs_rocm_lbvh_build(zs_rocm_policy *pol, zs_rocm_lbvh *b, const float *primBvsF, size_t n, int refit) {
  if (n == 0) return;  // Bvh.hpp:821
  Launch L(pol, "lbvh_build");
  const AABB3 *primBvs = (const AABB3 *)primBvsF;
  lbvh_reserve(*b, n);
  b->numLeaves = n;
  b->numNodes = n > 2 ? 2 * n - 1 : n;
  b->packedValid = false;
  b->hitCacheValid = false;
  if (n <= 2) {  // :823-831
    hipLaunchKernelGGL(lbvh_small_kernel, dim3(1), dim3(64), 0, L.stream, (int)n, primBvs, b->orderedBvs, b->leafInds, b->auxIndices,
                       b->parents, b->levels);
    return;
  }
  const int numLeaves = (int)n, numTrunk = numLeaves - 1;
  const int rb = (int)std::min<size_t>(ceil_div(n, BOX_BLOCK * 4), 1024);
  float *partial = (float *)L.temp(sizeof(float) * 6 * rb), *whole = (float *)L.temp(sizeof(float) * 8);
  unsigned *mcs = (unsigned *)L.temp(sizeof(unsigned) * n), *sortedMcs = (unsigned *)L.temp(sizeof(unsigned) * n);
  int *indices = (int *)L.temp(sizeof(int) * n), *pInds = (int *)L.temp(sizeof(int) * n);
  unsigned *spineLen = (unsigned *)L.temp(sizeof(unsigned) * (n + 1)), *depth = (unsigned *)L.temp(sizeof(unsigned) * (n + 1));
  unsigned *slot = (unsigned *)L.temp(sizeof(unsigned) * (n + 1));
  int *rangeFirst = (int *)L.temp(sizeof(int) * n), *rangeLast = (int *)L.temp(sizeof(int) * n), *leftTrunk = (int *)L.temp(sizeof(int) * n);
  int *trunkParent = (int *)L.temp(sizeof(int) * n), *leafParent = (int *)L.temp(sizeof(int) * n), *trunkPos = (int *)L.temp(sizeof(int) * n);
  unsigned char *spineHead = (unsigned char *)L.temp(n);
  hipLaunchKernelGGL(lbvh_box_reduce_kernel, dim3(rb), dim3(BOX_BLOCK), 0, L.stream, primBvs, n, partial, 1);
  hipLaunchKernelGGL(lbvh_box_final_kernel, dim3(1), dim3(64), 0, L.stream, partial, rb, whole);
  hipLaunchKernelGGL(lbvh_morton_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, primBvs, numLeaves, whole, mcs, indices);
  radix_sort_pair_u32(L, mcs, indices, sortedMcs, pInds, n, 0, 32);
  ZSR_CHECK(hipMemsetAsync(spineHead, 0, n, L.stream));
  ZSR_CHECK(hipMemsetAsync(spineLen, 0, sizeof(unsigned) * (n + 1), L.stream));
  hipLaunchKernelGGL(lbvh_ranges_kernel, dim3(ceil_div(numTrunk, 256)), dim3(256), 0, L.stream, sortedMcs, numLeaves, rangeFirst, rangeLast, leftTrunk,
                     trunkParent, leafParent, spineHead);
  hipLaunchKernelGGL((lbvh_spine_kernel<false>), dim3(ceil_div(numTrunk, 256)), dim3(256), 0, L.stream, numLeaves, rangeFirst, leftTrunk, spineHead,
                     spineLen, (const unsigned *)nullptr, (int *)nullptr, (int *)nullptr);
  hipLaunchKernelGGL(lbvh_spine_depths_kernel, dim3(ceil_div(n + 1, 256)), dim3(256), 0, L.stream, numLeaves, spineLen, depth);
  exclusive_scan_u32(L, depth, n + 1, slot);
  hipLaunchKernelGGL((lbvh_spine_kernel<true>), dim3(ceil_div(numTrunk, 256)), dim3(256), 0, L.stream, numLeaves, rangeFirst, leftTrunk, spineHead,
                     spineLen, (const unsigned *)slot, trunkPos, b->levels);
  hipLaunchKernelGGL(lbvh_emit_kernel, dim3(ceil_div(2 * n, 256)), dim3(256), 0, L.stream, numLeaves, rangeLast, trunkParent, leafParent, trunkPos,
                     (const unsigned *)slot, (const unsigned *)spineLen, (const int *)pInds, b->auxIndices, b->parents, b->levels, b->leafInds);
  if (refit) lbvh_refit_impl(L, *b, primBvs);
}
int zs_rocm_lbvh_refit(zs_rocm_policy *pol, zs_rocm_lbvh *b, const float *primBvs, size_t n) {
  if (n != b->numLeaves) return -1;  // "bvh topology changes, require rebuild!" (Bvh.hpp:1230-1231)
  Launch L(pol, "lbvh_refit");
  b->packedValid = false;
  b->hitCacheValid = false;
  lbvh_refit_impl(L, *b, (const AABB3 *)primBvs);
  return 0;
}
void zs_rocm_lbvh_total_box(zs_rocm_policy *pol, const zs_rocm_lbvh *b, float *box6Dev) {
  Launch L(pol, "lbvh_total_box");
  if (b->numLeaves == 0) return;
  if (b->numLeaves > 2) {  // root box (getTotalBox, Bvh.hpp:152-171)
    ZSR_CHECK(hipMemcpyAsync(box6Dev, b->orderedBvs, sizeof(AABB3), hipMemcpyDeviceToDevice, L.stream));
    return;
  }
  float *partial = (float *)L.temp(sizeof(float) * 6);
  hipLaunchKernelGGL(lbvh_box_reduce_kernel, dim3(1), dim3(BOX_BLOCK), 0, L.stream, b->orderedBvs, b->numLeaves, partial, 0);
  hipLaunchKernelGGL(lbvh_box_final_kernel, dim3(1), dim3(64), 0, L.stream, partial, 1, box6Dev);
}
#ifndef LBVH_QUERY_BLOCK
#define LBVH_QUERY_BLOCK 256
#endif
// bulk iter_neighbors: >= 16384 queries are walked in Morton order of their centres (codes + one pair sort: ~0.1 ms per million);
// ZS_ROCM_LBVH_QUERY=u keeps the caller's order (A/B runs).  nullptr: caller's order.
static const int *lbvh_query_order(Launch &L, const zs_rocm_lbvh *b, const float *queryBvs, size_t nq) {
  static const bool unsorted = [] { const char *e = getenv("ZS_ROCM_LBVH_QUERY"); return e && e[0] == 'u'; }();
  if (unsorted || nq < 16384 || nq > 0x7fffffffu || b->numNodes <= 2) return nullptr;
  unsigned *codes = (unsigned *)L.temp(sizeof(unsigned) * nq), *sorted = (unsigned *)L.temp(sizeof(unsigned) * nq);
  int *ids = (int *)L.temp(sizeof(int) * nq), *perm = (int *)L.temp(sizeof(int) * nq);
  hipLaunchKernelGGL(lbvh_query_code_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, L.stream, lbvh_packed(L, *b), (const AABB3 *)queryBvs, (int)nq,
                     codes, ids);
  radix_sort_pair_u32(L, codes, ids, sorted, perm, nq, 0, 30);
  return perm;
}
void zs_rocm_lbvh_query_count(zs_rocm_policy *pol, const zs_rocm_lbvh *b, const float *queryBvs, size_t nq, int *counts) {
  Launch L(pol, "lbvh_query_count");
  if (!nq) return;
  if (!b->numLeaves) { ZSR_CHECK(hipMemsetAsync(counts, 0, nq * sizeof(int), L.stream)); return; }
  const int *perm = lbvh_query_order(L, b, queryBvs, nq);
  hipLaunchKernelGGL((lbvh_query_packed_kernel<false>), dim3(ceil_div(nq, LBVH_QUERY_BLOCK)), dim3(LBVH_QUERY_BLOCK), 0, L.stream, lbvh_packed(L, *b), (int)b->numNodes,
                     (const AABB3 *)queryBvs, nq, counts, (const int *)nullptr, (int *)nullptr, perm);
}
void zs_rocm_lbvh_query_fill(zs_rocm_policy *pol, const zs_rocm_lbvh *b, const float *queryBvs, size_t nq, const int *offsets, int *out) {
  Launch L(pol, "lbvh_query_fill");
  if (!nq) return;
  if (!b->numLeaves) return;
  const int *perm = lbvh_query_order(L, b, queryBvs, nq);
  hipLaunchKernelGGL((lbvh_query_packed_kernel<true>), dim3(ceil_div(nq, LBVH_QUERY_BLOCK)), dim3(LBVH_QUERY_BLOCK), 0, L.stream, lbvh_packed(L, *b), (int)b->numNodes,
                     (const AABB3 *)queryBvs, nq, (int *)nullptr, offsets, out, perm);
}
// one wave per workgroup: the walks of neighbouring waves differ in length, and a long one would keep the other wave slots of its
// workgroup idle (count pass 2.97 -> 2.85 ms; 128 threads: 2.95)
#ifndef LBVH_SELF_BLOCK
#define LBVH_SELF_BLOCK 64
#endif
void zs_rocm_lbvh_self_query_count(zs_rocm_policy *pol, const zs_rocm_lbvh *b, int *counts) {
  Launch L(pol, "lbvh_self_query_count");
  if (!b->numLeaves) return;
  if (b->hitCacheLeaves < b->numLeaves) {
    (void)hipFree(b->hitCache); (void)hipFree(b->hitCounts);
    ZSR_CHECK(hipMalloc((void **)&b->hitCache, b->numLeaves * LBVH_HIT_CACHE * sizeof(int)));
    ZSR_CHECK(hipMalloc((void **)&b->hitCounts, b->numLeaves * sizeof(int)));
    b->hitCacheLeaves = b->numLeaves;
  }
  // A/B runs: ZS_ROCM_LBVH_SELF=l selects the one-walk-per-leaf kernel of r03
  static const bool perLeaf = [] { const char *e = getenv("ZS_ROCM_LBVH_SELF"); return e && e[0] == 'l'; }();
  // (the wave walk addresses nodes by a 32-bit byte offset, node << 5: trees of 2^27 nodes and more take the per-leaf kernel)
  if (b->numNodes > 2 && b->numNodes < (1u << 27) && !perLeaf)
    hipLaunchKernelGGL((lbvh_self_query_wave_kernel<false>), dim3(ceil_div(b->numLeaves, LBVH_SELF_BLOCK)), dim3(LBVH_SELF_BLOCK), 0, L.stream, lbvh_packed(L, *b),
                       (int)b->numNodes, (int)b->numLeaves, (const int *)b->leafInds, counts, (const int *)nullptr, (int *)nullptr,
                       b->hitCache, b->hitCounts, 0);
  else
    hipLaunchKernelGGL((lbvh_self_query_packed_kernel<false>), dim3(ceil_div(b->numLeaves, 256)), dim3(256), 0, L.stream, lbvh_packed(L, *b),
                       (int)b->numNodes, (int)b->numLeaves, (const int *)b->leafInds, counts, (const int *)nullptr, (int *)nullptr,
                       b->hitCache, b->hitCounts, 0);
  b->hitCacheValid = true;  // until the next build / refit
}
void zs_rocm_lbvh_self_query_fill(zs_rocm_policy *pol, const zs_rocm_lbvh *b, const int *offsets, int *pairs) {
  Launch L(pol, "lbvh_self_query_fill");
  if (!b->numLeaves) return;
  static const bool perLeaf = [] { const char *e = getenv("ZS_ROCM_LBVH_SELF"); return e && e[0] == 'l'; }();
  if (b->numNodes > 2 && b->numNodes < (1u << 27) && !perLeaf)
    hipLaunchKernelGGL((lbvh_self_query_wave_kernel<true>), dim3(ceil_div(b->numLeaves, LBVH_SELF_BLOCK)), dim3(LBVH_SELF_BLOCK), 0, L.stream, lbvh_packed(L, *b),
                       (int)b->numNodes, (int)b->numLeaves, (const int *)b->leafInds, (int *)nullptr, offsets, pairs, b->hitCache, b->hitCounts,
                       b->hitCacheValid ? 1 : 0);
  else
    hipLaunchKernelGGL((lbvh_self_query_packed_kernel<true>), dim3(ceil_div(b->numLeaves, 256)), dim3(256), 0, L.stream, lbvh_packed(L, *b),
                       (int)b->numNodes, (int)b->numLeaves, (const int *)b->leafInds, (int *)nullptr, offsets, pairs, b->hitCache, b->hitCounts,
                       b->hitCacheValid ? 1 : 0);
}

}  // extern "C"
