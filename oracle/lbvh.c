/*
 * lbvh.c -- CPU restatement of zs::LBvh<3, int, f32> (container/Bvh.hpp): build (Karras 2012 topology over sorted
 * 30-bit morton codes, pre-order node layout with escape indices), refit, and the stack-less traversal of
 * LBvhView::iter_neighbors.  TEST INFRASTRUCTURE ONLY (see zpc_oracle.h).  Sequential; every integer array of the
 * result is a deterministic function of the input boxes, so the GPU build is compared bit for bit.
 */
#include "zpc_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

struct orc_lbvh {
  size_t numLeaves, numNodes;
  float *orderedBvs; /* [numNodes][6] = {min xyz, max xyz} (AABBBox<3,f32>) */
  int32_t *parents, *levels, *leafInds, *auxIndices;
};

/* math/bit/Bits.h:84-90,122-125 */
static uint32_t expand_bits_32(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
uint32_t orc_morton_3d_32(float x, float y, float z) {
  return (expand_bits_32((uint32_t)(x * 1024.f)) << 2) | (expand_bits_32((uint32_t)(y * 1024.f)) << 1)
         | expand_bits_32((uint32_t)(z * 1024.f));
}
static uint32_t count_lz(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 32u; }

/* compute_bounding_box, Bvh.hpp:11-23,39-84: min/max of the boxes padded by 10 epsilon */
void orc_lbvh_whole_box(const float *bvs, size_t n, float box[6]) {
  for (int d = 0; d < 3; ++d) { box[d] = FLT_MAX; box[3 + d] = -FLT_MAX; }
  for (size_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      float lo = bvs[6 * i + d] - 10 * FLT_EPSILON, hi = bvs[6 * i + 3 + d] + 10 * FLT_EPSILON;
      if (lo < box[d]) box[d] = lo;
      if (hi > box[3 + d]) box[3 + d] = hi;
    }
}
/* _build_init_mc_id, Bvh.hpp:177-188: code of the box centre in the unit cube of the whole box
   (BoundingVolumeInterface.hpp:12-31: centre = (lo + hi) / 2, coord = clamp(c - wlo, 0, len) / len) */
uint32_t orc_lbvh_morton(const float whole[6], const float bv[6]) {
  float u[3];
  for (int d = 0; d < 3; ++d) {
    float c = (bv[d] + bv[3 + d]) / 2;
    float len = whole[3 + d] - whole[d];
    float o = c - whole[d];
    o = o < 0.f ? 0.f : (o > len ? len : o);
    u[d] = o / len;
  }
  return orc_morton_3d_32(u[0], u[1], u[2]);
}

orc_lbvh *orc_lbvh_create(void) { return (orc_lbvh *)calloc(1, sizeof(orc_lbvh)); }
static void lbvh_free_arrays(orc_lbvh *b) {
  free(b->orderedBvs); free(b->parents); free(b->levels); free(b->leafInds); free(b->auxIndices);
  b->orderedBvs = NULL; b->parents = b->levels = b->leafInds = b->auxIndices = NULL;
}
void orc_lbvh_destroy(orc_lbvh *b) {
  if (!b) return;
  lbvh_free_arrays(b);
  free(b);
}
size_t orc_lbvh_num_leaves(const orc_lbvh *b) { return b->numLeaves; }
size_t orc_lbvh_num_nodes(const orc_lbvh *b) { return b->numNodes; }
const float *orc_lbvh_bvs(const orc_lbvh *b) { return b->orderedBvs; }
const int32_t *orc_lbvh_parents(const orc_lbvh *b) { return b->parents; }
const int32_t *orc_lbvh_levels(const orc_lbvh *b) { return b->levels; }
const int32_t *orc_lbvh_leaf_inds(const orc_lbvh *b) { return b->leafInds; }
const int32_t *orc_lbvh_aux_indices(const orc_lbvh *b) { return b->auxIndices; }

/* LBvh::refit, Bvh.hpp:1219-1248 with _refit_bottom_up (:469-492).  Children follow their parent in the node order, so
   one backward sweep is the sequential form of the flag-synchronised bottom-up walk. */
void orc_lbvh_refit(orc_lbvh *b, const float *primBvs) {
  if (b->numLeaves <= 2) {
    memcpy(b->orderedBvs, primBvs, b->numLeaves * 6 * sizeof(float));
    return;
  }
  for (size_t k = b->numNodes; k-- > 0;) {
    float *bv = b->orderedBvs + 6 * k;
    if (b->levels[k] == 0) {
      memcpy(bv, primBvs + 6 * (size_t)b->auxIndices[k], 6 * sizeof(float));
    } else {
      size_t lc = k + 1;
      size_t rc = b->levels[lc] ? (size_t)b->auxIndices[lc] : lc + 1;
      const float *l = b->orderedBvs + 6 * lc, *r = b->orderedBvs + 6 * rc;
      for (int d = 0; d < 3; ++d) {
        bv[d] = l[d] < r[d] ? l[d] : r[d];
        bv[3 + d] = l[3 + d] > r[3 + d] ? l[3 + d] : r[3 + d];
      }
    }
  }
}

/* LBvh::build, Bvh.hpp:810-1082 */
void orc_lbvh_build(orc_lbvh *b, const float *primBvs, size_t n, int refit) {
  if (n == 0) return;
  lbvh_free_arrays(b);
  b->numLeaves = n;
  if (n <= 2) { /* :823-831 */
    b->numNodes = n;
    b->orderedBvs = (float *)malloc(n * 6 * sizeof(float));
    memcpy(b->orderedBvs, primBvs, n * 6 * sizeof(float));
    b->leafInds = (int32_t *)malloc(n * 4);
    b->auxIndices = (int32_t *)malloc(n * 4);
    b->parents = (int32_t *)calloc(n, 4);
    b->levels = (int32_t *)calloc(n, 4);
    for (size_t i = 0; i < n; ++i) b->leafInds[i] = b->auxIndices[i] = (int32_t)i;
    return;
  }
  const int32_t numLeaves = (int32_t)n, numTrunk = numLeaves - 1;
  const size_t numNodes = 2 * n - 1;
  b->numNodes = numNodes;
  b->orderedBvs = (float *)calloc(numNodes * 6, sizeof(float));
  b->auxIndices = (int32_t *)malloc(numNodes * 4);
  b->parents = (int32_t *)malloc(numNodes * 4);
  b->levels = (int32_t *)malloc(numNodes * 4);
  b->leafInds = (int32_t *)malloc(n * 4);
  int32_t *tPars = (int32_t *)malloc(n * 4), *tLcs = (int32_t *)malloc(n * 4), *tRcs = (int32_t *)malloc(n * 4);
  int32_t *tLs = (int32_t *)malloc(n * 4), *tRs = (int32_t *)malloc(n * 4), *tDst = (int32_t *)malloc(n * 4);
  int32_t *lPars = (int32_t *)malloc(n * 4), *lLcas = (int32_t *)malloc(n * 4), *pInds = (int32_t *)malloc(n * 4);
  int32_t *lDepths = (int32_t *)malloc((n + 1) * 4), *lOffsets = (int32_t *)malloc((n + 1) * 4);
  uint32_t *mc0 = (uint32_t *)malloc(n * 4), *mcs = (uint32_t *)malloc(n * 4);
  int32_t *ind0 = (int32_t *)malloc(n * 4);

  float whole[6];
  orc_lbvh_whole_box(primBvs, n, whole);
  for (size_t i = 0; i < n; ++i) {
    mc0[i] = orc_lbvh_morton(whole, primBvs + 6 * i);
    ind0[i] = (int32_t)i;
  }
  orc_radix_sort_pair_u32(mc0, ind0, mcs, pInds, n, 0, 32); /* :898-902; pInds = sortedIndices (:190-199) */
  for (int32_t i = 0; i < numLeaves; ++i) lDepths[i] = 1;
  lDepths[numLeaves] = 0;

  /* _build_build_topo, :200-287 */
  for (int32_t idx = 0; idx < numTrunk; ++idx) {
    int32_t i = 0, j = 0;
    if (idx == 0) {
      i = 0;
      j = numLeaves - 1;
    } else {
      int32_t left = idx, right = idx, dir = 0;
      uint32_t minLZ = 0;
      uint32_t pre = mcs[idx - 1], cur = mcs[idx], nxt = mcs[idx + 1];
      if (pre == cur && cur == nxt) {
        for (++right; right < numLeaves - 1; ++right)
          if (mcs[right] != mcs[right + 1]) break;
        j = right;
        i = left;
      } else {
        uint32_t lLZ = count_lz(pre ^ cur), rLZ = count_lz(nxt ^ cur);
        if (lLZ > rLZ) { dir = -1; minLZ = rLZ; } else { dir = 1; minLZ = lLZ; }
        int32_t step;
        for (step = 2;; step <<= 1) {
          right = left + step * dir;
          if (!(right < numLeaves && right >= 0 && count_lz(mcs[right] ^ cur) > minLZ)) break;
        }
        int32_t len = 0;
        for (step >>= 1; step >= 1; step >>= 1) {
          right = left + (len + step) * dir;
          if (right < numLeaves && right >= 0)
            if (count_lz(mcs[right] ^ cur) > minLZ) len += step;
        }
        if (dir == 1) { i = left; j = left + len; } else { i = left - len; j = left; }
      }
    }
    lDepths[i] += 1;
    tLs[idx] = i;
    tRs[idx] = j;
    int32_t gamma;
    uint32_t lCode = mcs[i], rCode = mcs[j];
    if (lCode == rCode)
      gamma = i;
    else {
      uint32_t LZ = count_lz(lCode ^ rCode);
      int32_t step, len = 0;
      for (step = (j - i + 1) >> 1;; step = (step + 1) >> 1) {
        if (i + len + step <= numTrunk)
          if (count_lz(mcs[i + len + step] ^ lCode) > LZ) len += step;
        if (step <= 1) break;
      }
      gamma = i + len;
    }
    tLcs[idx] = gamma;
    tRcs[idx] = gamma + 1;
    int32_t mi = i < j ? i : j, ma = i > j ? i : j;
    if (mi == gamma) { lPars[gamma] = idx; tLcs[idx] += numTrunk; } else tPars[gamma] = idx;
    if (ma == gamma + 1) { lPars[gamma + 1] = idx; tRcs[idx] += numTrunk; } else tPars[gamma + 1] = idx;
    if (idx == 0) tPars[0] = -1;
  }
  /* exclusive_scan(leafDepths) -> leafOffsets, :914 */
  {
    int32_t run = 0;
    for (int32_t i = 0; i <= numLeaves; ++i) { lOffsets[i] = run; run += lDepths[i]; }
  }
  /* _build_supp_topo, :288-303 */
  for (int32_t idx = 0; idx < numLeaves; ++idx) {
    int32_t depth = lOffsets[idx + 1] - lOffsets[idx];
    int32_t dst = lOffsets[idx + 1] - 2;
    int32_t node = lPars[idx], ch = idx + numTrunk, level = 0;
    for (; --depth; node = tPars[node], --dst) {
      tDst[node] = dst;
      b->levels[dst] = ++level;
      ch = node;
    }
    lLcas[idx] = ch;
  }
  /* _build_reorder_leaf, :304-319 */
  for (int32_t idx = 0; idx < numLeaves; ++idx) {
    int32_t dst = lOffsets[idx + 1] - 1;
    b->auxIndices[dst] = pInds[idx];
    b->parents[dst] = tDst[lPars[idx]];
    b->levels[dst] = 0;
    b->leafInds[idx] = dst;
  }
  /* _build_reorder_trunk, :320-338 */
  for (int32_t idx = 0; idx < numTrunk; ++idx) {
    int32_t dst = tDst[idx], r = tRs[idx];
    if (r != numTrunk) {
      int32_t lca = lLcas[r + 1];
      b->auxIndices[dst] = lca < numTrunk ? tDst[lca] : lOffsets[r + 1];
    } else
      b->auxIndices[dst] = -1;
    b->parents[dst] = idx != 0 ? tDst[tPars[idx]] : -1;
  }
  free(tPars); free(tLcs); free(tRcs); free(tLs); free(tRs); free(tDst); free(lPars); free(lLcas); free(pInds);
  free(lDepths); free(lOffsets); free(mc0); free(mcs); free(ind0);
  if (refit) orc_lbvh_refit(b, primBvs);
}

static int bv_overlaps(const float *a, const float *q) { /* geometry/AnalyticLevelSet.h:262-266 */
  for (int d = 0; d < 3; ++d)
    if (q[d] > a[3 + d] || q[3 + d] < a[d]) return 0;
  return 1;
}
/* LBvhView::iter_neighbors, Bvh.hpp:644-680: primitive ids whose box overlaps `bv`, in traversal order; returns the count
   (out may be NULL; at most cap ids are written) */
size_t orc_lbvh_iter_neighbors(const orc_lbvh *b, const float bv[6], int32_t *out, size_t cap) {
  size_t cnt = 0;
  const int32_t numNodes = (int32_t)b->numNodes;
  if (numNodes <= 2) {
    for (int32_t i = 0; i < numNodes; ++i)
      if (bv_overlaps(b->orderedBvs + 6 * (size_t)i, bv)) { if (out && cnt < cap) out[cnt] = i; ++cnt; }
    return cnt;
  }
  int32_t node = 0;
  while (node != -1 && node != numNodes) {
    int32_t level = b->levels[node];
    for (; level; --level, ++node)
      if (!bv_overlaps(b->orderedBvs + 6 * (size_t)node, bv)) break;
    if (level == 0) {
      if (bv_overlaps(b->orderedBvs + 6 * (size_t)node, bv)) { if (out && cnt < cap) out[cnt] = b->auxIndices[node]; ++cnt; }
      node++;
    } else
      node = b->auxIndices[node];
  }
  return cnt;
}

/* LBvhView::self_iter_neighbors (container/Bvh.hpp:695-728): the walk starts AT the leaf with sorted index `leafId`; reports the
   leaf itself and every overlapping leaf after it in node order (ids = primitive indices) */
size_t orc_lbvh_self_iter_neighbors(const orc_lbvh *b, int32_t leafId, int32_t *out, size_t cap) {
  size_t cnt = 0;
  const int32_t numNodes = (int32_t)b->numNodes;
  if (numNodes <= 2) {
    const float *bv = b->orderedBvs + 6 * (size_t)leafId;
    for (int32_t i = leafId + 1; i != numNodes; ++i)
      if (bv_overlaps(b->orderedBvs + 6 * (size_t)i, bv)) { if (out && cnt < cap) out[cnt] = i; ++cnt; }
    return cnt;
  }
  int32_t node = b->leafInds[leafId];
  const float *bv = b->orderedBvs + 6 * (size_t)node;
  while (node != -1 && node != numNodes) {
    int32_t level = b->levels[node];
    for (; level; --level, ++node)
      if (!bv_overlaps(b->orderedBvs + 6 * (size_t)node, bv)) break;
    if (level == 0) {
      if (bv_overlaps(b->orderedBvs + 6 * (size_t)node, bv)) { if (out && cnt < cap) out[cnt] = b->auxIndices[node]; ++cnt; }
      node++;
    } else
      node = b->auxIndices[node];
  }
  return cnt;
}
