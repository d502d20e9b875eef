/*
 * collider.c -- CPU restatement of zs::Collider<AnalyticLevelSet<Plane|Cuboid|Sphere|Cylinder, f32, 3>>::resolveCollision
 * (geometry/Collider.h:82-112) with the level sets of geometry/AnalyticLevelSet.h:11-250, and of the grid pass
 * ApplyBoundaryConditionOnGridBlocks (simulation/grid/GridOp.hpp:111-164).
 * TEST INFRASTRUCTURE ONLY (see zpc_oracle.h).  Pinned against the reference's own AnalyticLevelSet / Rotation / vec code
 * through oracle/_ref (ref_collider_resolve) and the golden vectors in tests/golden/collider.npz.
 * Built with -ffp-contract=off: the finite-difference normals depend on the rounding of every operation.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct {
  int geometry, type; /* 0 plane, 1 cuboid, 2 sphere, 3 cylinder; 0 sticky, 1 slip, 2 separate (collider_e) */
  float param[8];
  float s, dsdt;
  float R[9]; /* row-major */
  float omega[3];
  float b[3], dbdt[3];
} orc_collider;

/* AnalyticLevelSet::do_getSignedDistance */
static float orc_sd(const orc_collider *c, const float X[3]) {
  const float *p = c->param;
  if (c->geometry == 0) /* :30-32 */
    return p[3] * (X[0] - p[0]) + p[4] * (X[1] - p[1]) + p[5] * (X[2] - p[2]);
  if (c->geometry == 1) { /* :90-97 */
    float pt[3], mx, s = 0.f;
    for (int d = 0; d < 3; ++d) {
      float center = (p[d] + p[3 + d]) / 2;
      pt[d] = fabsf(X[d] - center) - (p[3 + d] - p[d]) / 2;
    }
    mx = pt[0];
    for (int d = 1; d < 3; ++d)
      if (pt[d] > mx) mx = pt[d];
    for (int d = 0; d < 3; ++d) {
      if (pt[d] < 0) pt[d] = 0;
      s += pt[d] * pt[d];
    }
    return (mx < 0 ? mx : 0) + sqrtf(s);
  }
  if (c->geometry == 2) { /* :144-146 */
    float a = X[0] - p[0], b = X[1] - p[1], cc = X[2] - p[2];
    return sqrtf(a * a + b * b + cc * cc) - p[3];
  }
  { /* cylinder :188-215 */
    int ax = (int)p[5];
    float diffR[2], radius = p[3], length = p[4];
    for (int k = 0, i = 0; k != 3; ++k)
      if (k != ax) diffR[i++] = X[k] - p[k];
    float disR = sqrtf(diffR[0] * diffR[0] + diffR[1] * diffR[1]);
    int outsideCircle = disR > radius;
    if (X[ax] < p[ax]) {
      float disL = p[ax] - X[ax];
      return outsideCircle ? sqrtf((disR - radius) * (disR - radius) + disL * disL) : disL;
    } else if (X[ax] > p[ax] + length) {
      float disL = X[ax] - (p[ax] + length);
      return outsideCircle ? sqrtf((disR - radius) * (disR - radius) + disL * disL) : disL;
    } else {
      if (outsideCircle) return disR - radius;
      float e0 = p[ax] + length - X[ax], e1 = X[ax] - p[ax];
      float disL = e0 < e1 ? e0 : e1;
      float e2 = radius - disR;
      return -(disL < e2 ? disL : e2);
    }
  }
}
static void orc_normal(const orc_collider *c, const float X[3], float n[3]) {
  const float *p = c->param;
  if (c->geometry == 0) {
    n[0] = p[3]; n[1] = p[4]; n[2] = p[5];
    return;
  }
  if (c->geometry == 2) { /* :148-152 */
    float a = X[0] - p[0], b = X[1] - p[1], cc = X[2] - p[2];
    float l2 = a * a + b * b + cc * cc;
    if (l2 < 1e-7f) { n[0] = n[1] = n[2] = 0.f; return; }
    float l = sqrtf(l2);
    n[0] = a / l; n[1] = b / l; n[2] = cc / l;
    return;
  }
  { /* :99-111 / :217-229 */
    float eps = 1e-6f, diff[3];
    for (int i = 0; i < 3; ++i) {
      float v1[3] = {X[0], X[1], X[2]}, v2[3] = {X[0], X[1], X[2]};
      v1[i] = X[i] + eps;
      v2[i] = X[i] - eps;
      diff[i] = (orc_sd(c, v1) - orc_sd(c, v2)) / (eps + eps);
    }
    float l = sqrtf(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
    n[0] = diff[0] / l; n[1] = diff[1] / l; n[2] = diff[2] / l;
  }
}

/* Collider::resolveCollision(x, v, erosion = 0), geometry/Collider.h:82-112 */
int orc_collider_resolve(const orc_collider *c, const float x[3], float v[3]) {
  float xmb[3], X[3];
  float one_over_s = 1 / c->s;
  for (int d = 0; d < 3; ++d) xmb[d] = x[d] - c->b[d];
  for (int d = 0; d < 3; ++d) X[d] = (c->R[d] * xmb[0] + c->R[3 + d] * xmb[1] + c->R[6 + d] * xmb[2]) * one_over_s;
  if (!(orc_sd(c, X) < -0.f)) return 0;
  float k = c->dsdt * one_over_s;
  float vo[3] = {c->omega[1] * xmb[2] - c->omega[2] * xmb[1], c->omega[2] * xmb[0] - c->omega[0] * xmb[2],
                 c->omega[0] * xmb[1] - c->omega[1] * xmb[0]};
  for (int d = 0; d < 3; ++d) vo[d] = (vo[d] + k * xmb[d]) + c->dbdt[d];
  if (c->type == 0) {
    for (int d = 0; d < 3; ++d) v[d] = vo[d];
  } else {
    float nm[3], n[3];
    orc_normal(c, X, nm);
    for (int d = 0; d < 3; ++d) {
      v[d] -= vo[d];
      n[d] = c->R[3 * d] * nm[0] + c->R[3 * d + 1] * nm[1] + c->R[3 * d + 2] * nm[2];
    }
    float proj = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
    if ((c->type == 2 && proj < 0) || c->type == 1)
      for (int d = 0; d < 3; ++d) v[d] -= proj * n[d];
    for (int d = 0; d < 3; ++d) v[d] += vo[d];
  }
  return 1;
}
void orc_collider_resolve_many(const orc_collider *c, const float *x, float *v, size_t n, int32_t *inside) {
  for (size_t i = 0; i < n; ++i) {
    int in = orc_collider_resolve(c, x + 3 * i, v + 3 * i);
    if (inside) inside[i] = in;
  }
}
/* ApplyBoundaryConditionOnGridBlocks::operator() (GridOp.hpp:137-158) over grid[(b*7 + ch)*side^3 + cell] */
void orc_mpm_apply_boundary(const orc_collider *c, const int32_t *keys, float *grid, size_t nblocks, int side, int kscale, float dx) {
  const int nc = side * side * side;
  for (size_t b = 0; b < nblocks; ++b)
    for (int cell = 0; cell < nc; ++cell) {
      float *g = grid + b * 7 * nc + cell;
      if (!(g[0] > 0)) continue;
      int cc[3] = {cell / (side * side), (cell / side) % side, cell % side};
      float pos[3], vel[3];
      for (int d = 0; d < 3; ++d) {
        pos[d] = (float)(keys[3 * b + d] / kscale * side + cc[d]) * dx;
        vel[d] = g[(1 + d) * nc];
      }
      orc_collider_resolve(c, pos, vel);
      for (int d = 0; d < 3; ++d) g[(1 + d) * nc] = vel[d];
    }
}
