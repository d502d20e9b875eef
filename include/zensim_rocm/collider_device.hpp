// collider_device.hpp -- device (and host) side of zs::Collider<AnalyticLevelSet<Plane|Cuboid|Sphere|Cylinder, f32, 3>>
// (geometry/Collider.h:10-206, geometry/AnalyticLevelSet.h:11-250): signed distance, normal, and the grid-velocity
// projection `resolveCollision(x, v)` the boundary pass of the MPM sub-step applies to every grid node
// (simulation/grid/GridOp.hpp:111-164).  Plain struct = zs_rocm_collider of the C ABI; usable inside user lambdas.
// The arithmetic follows the reference operation by operation (the cuboid / cylinder normals are float finite differences
// with eps = 1e-6, so evaluation order matters): compile translation units that use it with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include "../zs_rocm.h"

namespace zsr {

struct ColliderDev : zs_rocm_collider {
  __host__ __device__ ColliderDev() = default;
  __host__ __device__ ColliderDev(const zs_rocm_collider &c) : zs_rocm_collider(c) {}

  // AnalyticLevelSet<...>::do_getSignedDistance, material space
  __host__ __device__ __forceinline__ float signed_distance(const float (&X)[3]) const {
    switch (geometry) {
      case ZS_ROCM_GEOM_PLANE:  // :30-32  normal . (x - origin)
        return param[3] * (X[0] - param[0]) + param[4] * (X[1] - param[1]) + param[5] * (X[2] - param[2]);
      case ZS_ROCM_GEOM_CUBOID: {  // :90-97
        float mx = 0.f, s = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float c = (param[d] + param[3 + d]) / 2;
          float q = fabsf(X[d] - c) - (param[3 + d] - param[d]) / 2;
          mx = d == 0 ? q : (q > mx ? q : mx);
          q = q < 0.f ? 0.f : q;
          s += q * q;
        }
        return (mx < 0.f ? mx : 0.f) + sqrtf(s);
      }
      case ZS_ROCM_GEOM_SPHERE: {  // :144-146
        const float a = X[0] - param[0], b = X[1] - param[1], c = X[2] - param[2];
        return sqrtf(a * a + b * b + c * c) - param[3];
      }
      default: {  // cylinder :188-215: bottom centre param[0..2], radius param[3], length param[4], axis param[5]
        const int ax = (int)param[5];
        float r2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k != ax) {
            const float q = X[k] - param[k];
            r2 += q * q;
          }
        const float disR = sqrtf(r2), radius = param[3], length = param[4];
        const float xa = X[ax], ba = param[ax];
        const bool outside = disR > radius;
        if (xa < ba) {
          const float disL = ba - xa;
          return outside ? sqrtf((disR - radius) * (disR - radius) + disL * disL) : disL;
        } else if (xa > ba + length) {
          const float disL = xa - (ba + length);
          return outside ? sqrtf((disR - radius) * (disR - radius) + disL * disL) : disL;
        } else {
          if (outside) return disR - radius;
          const float e0 = ba + length - xa, e1 = xa - ba;
          const float disL = e0 < e1 ? e0 : e1;
          const float e2 = radius - disR;
          return -(disL < e2 ? disL : e2);
        }
      }
    }
  }
  // do_getNormal, material space
  __host__ __device__ __forceinline__ void normal(const float (&X)[3], float (&n)[3]) const {
    if (geometry == ZS_ROCM_GEOM_PLANE) {
      n[0] = param[3]; n[1] = param[4]; n[2] = param[5];
      return;
    }
    if (geometry == ZS_ROCM_GEOM_SPHERE) {  // :148-152
      const float a = X[0] - param[0], b = X[1] - param[1], c = X[2] - param[2];
      const float l2 = a * a + b * b + c * c;
      if (l2 < 1e-7f) { n[0] = n[1] = n[2] = 0.f; return; }
      const float l = sqrtf(l2);
      n[0] = a / l; n[1] = b / l; n[2] = c / l;
      return;
    }
    // cuboid :99-111, cylinder :217-229: central differences of the signed distance, eps = 1e-6, then normalised
    const float eps = 1e-6f;
    float diff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float v1[3] = {X[0], X[1], X[2]}, v2[3] = {X[0], X[1], X[2]};
      v1[i] = X[i] + eps;
      v2[i] = X[i] - eps;
      diff[i] = (signed_distance(v1) - signed_distance(v2)) / (eps + eps);
    }
    const float l = sqrtf(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
    n[0] = diff[0] / l; n[1] = diff[1] / l; n[2] = diff[2] / l;
  }
  // material-space point X = (1/s) R^T (x - b)
  __host__ __device__ __forceinline__ void to_material(const float (&x)[3], float (&xmb)[3], float (&X)[3]) const {
    const float one_over_s = 1 / s;
#pragma unroll
    for (int d = 0; d < 3; ++d) xmb[d] = x[d] - b[d];
#pragma unroll
    for (int d = 0; d < 3; ++d) X[d] = (R[d] * xmb[0] + R[3 + d] * xmb[1] + R[6 + d] * xmb[2]) * one_over_s;  // R^T: R is row-major
  }
  __host__ __device__ __forceinline__ bool queryInside(const float (&x)[3]) const {  // Collider.h:25-30
    float xmb[3], X[3];
    to_material(x, xmb, X);
    return signed_distance(X) < 0.f;
  }
  // Collider::resolveCollision(x, v, erosion) (Collider.h:82-112); returns true when x is inside
  __host__ __device__ __forceinline__ bool resolveCollision(const float (&x)[3], float (&v)[3], float erosion = 0.f) const {
    float xmb[3], X[3];
    to_material(x, xmb, X);
    if (!(signed_distance(X) < -erosion)) return false;
    const float one_over_s = 1 / s, k = dsdt * one_over_s;
    // v_object = omega x (x-b) + (s'/s)(x-b) + R s X' + b'   (X' = material velocity = 0 for the analytic shapes)
    float vo[3] = {omega[1] * xmb[2] - omega[2] * xmb[1], omega[2] * xmb[0] - omega[0] * xmb[2], omega[0] * xmb[1] - omega[1] * xmb[0]};
#pragma unroll
    for (int d = 0; d < 3; ++d) vo[d] = (vo[d] + k * xmb[d]) + dbdt[d];
    if (type == ZS_ROCM_COLLIDER_STICKY) {
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] = vo[d];
    } else {
      float nm[3], n[3];
      normal(X, nm);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        v[d] -= vo[d];
        n[d] = R[3 * d] * nm[0] + R[3 * d + 1] * nm[1] + R[3 * d + 2] * nm[2];
      }
      const float proj = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
      if ((type == ZS_ROCM_COLLIDER_SEPARATE && proj < 0.f) || type == ZS_ROCM_COLLIDER_SLIP) {
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] -= proj * n[d];
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] += vo[d];
    }
    return true;
  }
};

}  // namespace zsr
