// lbvh_device.hpp -- device view of zs::LBvh<3, int, f32> (container/Bvh.hpp:495-800 LBvhView) for gfx950.
//
// Node layout (Bvh.hpp:288-338): nodes in depth-first pre-order; the left child of trunk node k is k + 1; `levels[k]` =
// number of trunk nodes on the left spine below and including k (0 for a leaf); `auxIndices[k]` = primitive id for a leaf,
// escape index (next node in pre-order that is not in k's subtree, -1 at the end) for a trunk node; `parents`;
// `leafInds[sorted leaf] -> node`.  Boxes are AABBBox<3, f32> = {min xyz, max xyz}, 24 bytes.
// Traversal is the reference's stack-less walk: run down the left spine while boxes overlap, then follow escape indices.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace zsr {

struct AABB3 {
  float lo[3], hi[3];
};
__host__ __device__ __forceinline__ bool aabb_overlaps(const AABB3 &a, const AABB3 &b) {  // geometry/AnalyticLevelSet.h:262-266
#pragma unroll
  for (int d = 0; d < 3; ++d)
    if (b.lo[d] > a.hi[d] || b.hi[d] < a.lo[d]) return false;
  return true;
}

struct LBvhDev {
  const AABB3 *orderedBvs;
  const int *parents, *levels, *leafInds, *auxIndices;
  int numNodes;
  __device__ __forceinline__ int numLeaves() const { return numNodes > 2 ? (numNodes + 1) / 2 : numNodes; }
  __device__ __forceinline__ AABB3 getNodeBV(int node) const { return orderedBvs[node]; }

  // iter_neighbors (Bvh.hpp:644-680): f(primitive id) for every leaf whose box overlaps bv; a bool-returning f stops the
  // walk when it returns true
  template <class F> __device__ __forceinline__ void iter_neighbors(const AABB3 &bv, F &&f) const {
    auto call = [&](int id) -> bool {
      if constexpr (std::is_same_v<decltype(f(id)), void>) {
        f(id);
        return false;
      } else
        return f(id);
    };
    if (numNodes <= 2) {
      for (int i = 0; i != numNodes; ++i)
        if (aabb_overlaps(orderedBvs[i], bv))
          if (call(i)) return;
      return;
    }
    int node = 0;
    while (node != -1 && node != numNodes) {
      int level = levels[node];
      for (; level; --level, ++node)
        if (!aabb_overlaps(orderedBvs[node], bv)) break;
      if (level == 0) {
        if (aabb_overlaps(orderedBvs[node], bv))
          if (call(auxIndices[node])) return;
        node++;
      } else
        node = auxIndices[node];
    }
  }
  // self_iter_neighbors (Bvh.hpp:695-728): the walk starts AT the leaf of sorted index `leafId`, so the leaf itself and
  // every overlapping leaf after it in node order are reported -- each unordered pair once over all leafIds
  template <class F> __device__ __forceinline__ void self_iter_neighbors(int leafId, F &&f) const {
    auto call = [&](int id) -> bool {
      if constexpr (std::is_same_v<decltype(f(id)), void>) {
        f(id);
        return false;
      } else
        return f(id);
    };
    if (numNodes <= 2) {
      const AABB3 bv = orderedBvs[leafId];
      for (int i = leafId + 1; i != numNodes; ++i)
        if (aabb_overlaps(orderedBvs[i], bv))
          if (call(i)) return;
      return;
    }
    int node = leafInds[leafId];
    const AABB3 bv = orderedBvs[node];
    while (node != -1 && node != numNodes) {
      int level = levels[node];
      for (; level; --level, ++node)
        if (!aabb_overlaps(orderedBvs[node], bv)) break;
      if (level == 0) {
        if (aabb_overlaps(orderedBvs[node], bv))
          if (call(auxIndices[node])) return;
        node++;
      } else
        node = auxIndices[node];
    }
  }
  // iter_neighbors over the subtree of `node` (Bvh.hpp:730-750)
  template <class F> __device__ __forceinline__ void iter_neighbors(const AABB3 &bv, int node, F &&f) const {
    if (numNodes <= 2) {
      if (aabb_overlaps(orderedBvs[node], bv)) f(node);
      return;
    }
    const int ed = levels[node] != 0 ? auxIndices[node] : node + 1;
    while (node != ed && node != numNodes) {
      int level = levels[node];
      for (; level; --level, ++node)
        if (!aabb_overlaps(orderedBvs[node], bv)) break;
      if (level == 0) {
        if (aabb_overlaps(orderedBvs[node], bv)) f(auxIndices[node]);
        node++;
      } else
        node = auxIndices[node];
    }
  }
  // find_nearest_point-style walk with a user distance functor (Bvh.hpp:553-590): f(primId, dist&, idx&) updates the best
  template <class F> __device__ __forceinline__ float find_nearest(const float p[3], F &&f, float cap, int *bestIdx = nullptr) const {
    auto boxdist = [&](const AABB3 &b) {  // distance(point, AABB): 0 inside
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float e = fmaxf(fmaxf(b.lo[d] - p[d], p[d] - b.hi[d]), 0.f);
        s += e * e;
      }
      return sqrtf(s);
    };
    int idx = -1;
    float dist = cap;
    if (numNodes <= 2) {
      for (int i = 0; i != numNodes; ++i)
        if (boxdist(orderedBvs[i]) < dist) f(i, dist, idx);
    } else {
      int node = 0;
      while (node != -1 && node != numNodes) {
        int level = levels[node];
        for (; level; --level, ++node)
          if (boxdist(orderedBvs[node]) > dist) break;
        if (level == 0) {
          if (boxdist(orderedBvs[node]) < dist) f(auxIndices[node], dist, idx);
          node++;
        } else
          node = auxIndices[node];
      }
    }
    if (bestIdx) *bestIdx = idx;
    return dist;
  }
};

}  // namespace zsr
