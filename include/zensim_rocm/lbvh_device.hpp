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
  // signed distance point <-> box (geometry/AnalyticLevelSet.h:292-305): negative inside (largest face distance)
  __host__ __device__ __forceinline__ static float box_distance(const AABB3 &b, const float (&p)[3]) {
    float mx = -3.402823466e+38f, s = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float c = (b.lo[d] + b.hi[d]) / 2;
      float q = fabsf(p[d] - c) - (b.hi[d] - b.lo[d]) / 2;
      mx = q > mx ? q : mx;
      q = q < 0.f ? 0.f : q;
      s += q * q;
    }
    return (mx < 0.f ? mx : 0.f) + sqrtf(s);
  }
  // ray / box slab test (geometry/AnalyticLevelSet.h:389-412; division by zero follows IEEE as in the reference)
  __host__ __device__ __forceinline__ static bool ray_box_intersect(const float (&ro)[3], const float (&rd)[3], const AABB3 &b) {
    const float ix = 1.f / rd[0], iy = 1.f / rd[1], iz = 1.f / rd[2];
    float tmin = ((ix < 0 ? b.hi[0] : b.lo[0]) - ro[0]) * ix, tmax = ((ix < 0 ? b.lo[0] : b.hi[0]) - ro[0]) * ix;
    const float tymin = ((iy < 0 ? b.hi[1] : b.lo[1]) - ro[1]) * iy, tymax = ((iy < 0 ? b.lo[1] : b.hi[1]) - ro[1]) * iy;
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    const float tzmin = ((iz < 0 ? b.hi[2] : b.lo[2]) - ro[2]) * iz, tzmax = ((iz < 0 ? b.lo[2] : b.hi[2]) - ro[2]) * iz;
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmax < tmax) tmax = tzmax;
    return tmax >= 0.f;
  }
  // ray_intersect (Bvh.hpp:521-545): f(primitive id) for every leaf box the ray hits
  template <class F> __device__ __forceinline__ void ray_intersect(const float (&ro)[3], const float (&rd)[3], F &&f) const {
    if (numNodes <= 2) {
      for (int i = 0; i != numNodes; ++i)
        if (ray_box_intersect(ro, rd, orderedBvs[i])) f(i);
      return;
    }
    int node = 0;
    while (node != -1 && node != numNodes) {
      int level = levels[node];
      for (; level; --level, ++node)
        if (!ray_box_intersect(ro, rd, orderedBvs[node])) break;
      if (level == 0) {
        if (ray_box_intersect(ro, rd, orderedBvs[node])) f(auxIndices[node]);
        node++;
      } else
        node = auxIndices[node];
    }
  }
  // find_nearest (Bvh.hpp:547-590): walk pruned by the (signed) box distance; f(primId, dist&, idx&) refines the current best
  template <class F> __device__ __forceinline__ float find_nearest(const float (&p)[3], F &&f, float cap, int *bestIdx = nullptr) const {
    int idx = -1;
    float dist = cap;
    if (numNodes <= 2) {
      for (int i = 0; i != numNodes; ++i)
        if (box_distance(orderedBvs[i], p) < dist) f(i, dist, idx);
    } else {
      int node = 0;
      while (node != -1 && node != numNodes) {
        int level = levels[node];
        for (; level; --level, ++node)
          if (box_distance(orderedBvs[node], p) > dist) break;
        if (level == 0) {
          if (box_distance(orderedBvs[node], p) < dist) f(auxIndices[node], dist, idx);
          node++;
        } else
          node = auxIndices[node];
      }
    }
    if (bestIdx) *bestIdx = idx;
    return dist;
  }
  // find_nearest_point (Bvh.hpp:622-661): leaves are points (_min of their box); returns the distance, *bestIdx the primitive
  __device__ __forceinline__ float find_nearest_point(const float (&p)[3], float dist2 = 3.402823466e+38f, int *bestIdx = nullptr) const {
    int idx = -1;
    auto d2pt = [&](const AABB3 &b) {
      const float x = p[0] - b.lo[0], y = p[1] - b.lo[1], z = p[2] - b.lo[2];
      return x * x + y * y + z * z;
    };
    if (numNodes <= 2) {
      for (int i = 0; i != numNodes; ++i) {
        const float d2 = d2pt(orderedBvs[i]);
        if (d2 < dist2) { dist2 = d2; idx = i; }
      }
    } else {
      int node = 0;
      while (node != -1 && node != numNodes) {
        int level = levels[node];
        for (; level; --level, ++node) {
          const float d = fmaxf(0.f, box_distance(orderedBvs[node], p));
          if (d * d > dist2) break;
        }
        if (level == 0) {
          const float d2 = d2pt(orderedBvs[node]);
          if (d2 < dist2) { dist2 = d2; idx = auxIndices[node]; }
          node++;
        } else
          node = auxIndices[node];
      }
    }
    if (bestIdx) *bestIdx = idx;
    return sqrtf(dist2);
  }
};

}  // namespace zsr
