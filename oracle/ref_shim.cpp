/*
 * ref_shim.cpp -- thin extern "C" wrapper that compiles the REFERENCE's own header-only numerics in
 * place (from /root/reference/include, nothing copied) into oracle/_ref/libzpcref.so.
 * TEST INFRASTRUCTURE ONLY: it is used to pin the C restatement in this directory and to generate
 * the golden vectors under tests/golden (tools/gen_golden.py).  It exists only in the build
 * container; the GPU box never sees /root/reference and uses the committed fixtures instead.
 *
 * Only headers that build with the toolchain and libraries already in the image are used (the
 * bundled fmt of PyTorch's include dir serves zensim/zpc_tpls/fmt).  The containers, execution
 * policies and transfer functors of the reference need the un-vendored magic_enum / plog headers
 * and are NOT built ("Oracle").
 */
#include "zensim/zpc_tpls/fmt/format.h"
#include "zensim/math/matrix/SVD.hpp"
#include "zensim/physics/ConstitutiveModel_Vol_dP.hpp"
#include "zensim/math/curve/InterpolationKernel.hpp"
#include "zensim/math/Hash.hpp" /* must precede HashUtils.hpp, which calls hash_combine unqualified */
#include "zensim/py_interop/HashUtils.hpp"
#include "zensim/math/bit/Bits.h"
#include "zensim/geometry/AnalyticLevelSet.h" /* AABBBox, overlaps, BoundingVolumeInterface */
#include "zensim/math/Rotation.hpp"            /* Rotation, AngularVelocity (members of Collider) */
#include <random>

using namespace zs;

/* Collider<LS>::resolveCollision (geometry/Collider.h:82-112).  Collider.h itself includes GenericLevelSet.h, which pulls in
   the execution-policy headers (un-vendored magic_enum: unbuildable here), so the member function is spelled out over the
   reference's own AnalyticLevelSet (getSignedDistance / getNormal / getMaterialVelocity through LevelSetInterface), Rotation,
   AngularVelocity and vec types -- those are what this pins. */
template <class LS>
static int ref_resolve_with(const LS &levelset, int type, float s, float dsdt, const float *Rm, const float *om, const float *b_,
                            const float *dbdt_, const float *x_, float *v_) {
  using T = float;
  using TV = vec<float, 3>;
  Rotation<float, 3> R{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = Rm[3 * i + j];
  AngularVelocity<float, 3> omega{TV{om[0], om[1], om[2]}};
  TV b{b_[0], b_[1], b_[2]}, dbdt{dbdt_[0], dbdt_[1], dbdt_[2]}, x{x_[0], x_[1], x_[2]}, v{v_[0], v_[1], v_[2]};
  T erosion = 0;
  int hit = 0;
  TV x_minus_b = x - b;
  T one_over_s = 1 / s;
  TV X = R.transpose() * x_minus_b * one_over_s;
  if (levelset.getSignedDistance(X) < -erosion) {
    TV v_object = omega.cross(x_minus_b) + (dsdt * one_over_s) * x_minus_b + R * s * levelset.getMaterialVelocity(X) + dbdt;
    if (type == 0)
      v = v_object;
    else {
      v -= v_object;
      TV n = R * levelset.getNormal(X);
      T proj = n.dot(v);
      if ((type == 2 && proj < 0) || type == 1) v -= proj * n;
      v += v_object;
    }
    hit = 1;
  }
  for (int d = 0; d < 3; ++d) v_[d] = v[d];
  return hit;
}

extern "C" {

/* column-major 9-vectors, argument order as at the reference call sites
 * (physics/ConstitutiveModel_Vol_dP.hpp:14-16) */
void ref_svd3(const float *F, float *U, float *S, float *V) {
  math::svd_3d(F[0], F[3], F[6], F[1], F[4], F[7], F[2], F[5], F[8], U[0], U[3], U[6], U[1], U[4],
               U[7], U[2], U[5], U[8], S[0], S[1], S[2], V[0], V[3], V[6], V[1], V[4], V[7], V[2],
               V[5], V[8]);
}
void ref_lame(float E, float nu, float *mu, float *lam) {
  auto [m, l] = lame_parameters(E, nu);
  *mu = m;
  *lam = l;
}
void ref_stress_fixedcorotated(float volume, float mu, float lam, const float *F, float *PF) {
  vec<float, 9> f{}, pf{};
  for (int i = 0; i < 9; ++i) f[i] = F[i];
  compute_stress_fixedcorotated(volume, mu, lam, f, pf);
  for (int i = 0; i < 9; ++i) PF[i] = pf[i];
}
void ref_stress_sand(float volume, float mu, float lam, float cohesion, float beta,
                     float yieldSurface, int volCorrection, float *logJp, float *F, float *PF) {
  vec<float, 9> f{}, pf{};
  for (int i = 0; i < 9; ++i) f[i] = F[i];
  compute_stress_sand(volume, mu, lam, cohesion, beta, yieldSurface, (bool)volCorrection, *logJp, f, pf);
  for (int i = 0; i < 9; ++i) {
    PF[i] = pf[i];
    F[i] = f[i];
  }
}
void ref_stress_vonmises(float volume, float mu, float lam, float yieldStress, float *F, float *PF) {
  vec<float, 9> f{}, pf{};
  for (int i = 0; i < 9; ++i) f[i] = F[i];
  compute_stress_vonmisesfixedcorotated(volume, mu, lam, yieldStress, f, pf);
  for (int i = 0; i < 9; ++i) {
    PF[i] = pf[i];
    F[i] = f[i];
  }
}
void ref_stress_nacc(float volume, float mu, float lam, float bm, float xi, float beta, float Msqr, int hardeningOn, float *logJp,
                     float *F, float *PF) {
  vec<float, 9> f{}, pf{};
  for (int i = 0; i < 9; ++i) f[i] = F[i];
  compute_stress_nacc(volume, mu, lam, bm, xi, beta, Msqr, (bool)hardeningOn, *logJp, f, pf);
  for (int i = 0; i < 9; ++i) {
    PF[i] = pf[i];
    F[i] = f[i];
  }
}
/* NACCConfig (physics/ConstitutiveModel.hpp:759-785) */
void ref_nacc_config(float E, float nu, float fa, float *bulk, float *Msqr) {
  NACCConfig c{};
  c.E = E;
  c.nu = nu;
  c.fa = fa;
  *bulk = c.bulk();
  *Msqr = c.Msqr();
}
/* base_node<1> and quadratic_bspline_weights<0>: InterpolationKernel.hpp:47-55,93-130 */
int ref_base_node_quadratic(float x) { return base_node<1>(x); }
void ref_quadratic_weights(const float *x, float *w /*[3][3]*/) {
  vec<float, 3> p{x[0], x[1], x[2]};
  auto ws = quadratic_bspline_weights<0>(p);
  auto &m = get<0>(ws);
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < 3; ++k) w[3 * d + k] = m(d, k);
}
/* universal_hash on vec<int, dim>: py_interop/HashUtils.hpp:23-43 + math/Hash.hpp:19-28 */
unsigned ref_universal_hash3(unsigned hx, unsigned hy, const int *k) {
  universal_hash_base<vec<int, 3>> h{hx, hy};
  return h(vec<int, 3>{k[0], k[1], k[2]});
}
unsigned ref_universal_hash2(unsigned hx, unsigned hy, const int *k) {
  universal_hash_base<vec<int, 2>> h{hx, hy};
  return h(vec<int, 2>{k[0], k[1]});
}
unsigned ref_universal_hash1(unsigned hx, unsigned hy, int k) {
  universal_hash_base<int> h{hx, hy};
  return h(k);
}
/* seeds exactly as universal_hash(std::mt19937&) draws them (container/Bcht.hpp:39-43) from
 * std::mt19937 rng(2) (container/Bht.hpp:165-169); Bcht.hpp itself needs the execution policy
 * headers, so the three draws are restated here around the real std::mt19937 */
void ref_bht_hash_params(unsigned *out) {
  std::mt19937 rng(2);
  constexpr unsigned prime = universal_hash_base<int>::prime_divisor;
  for (int f = 0; f < 3; ++f) {
    unsigned hx = rng() % prime;
    if (hx < 1) hx = 1;
    unsigned hy = rng() % prime;
    out[2 * f] = hx;
    out[2 * f + 1] = hy;
  }
}
unsigned long long ref_next_2pow(unsigned long long n) { return next_2pow(n); }
unsigned ref_hash_combine32(unsigned seed, unsigned val) {
  hash_combine(seed, val);
  return seed;
}
/* HashTableView::do_hash (container/HashTable.hpp:496-500) is three lines over hash_combine with a size_t seed; HashTable.hpp
   itself pulls in the execution-policy headers (unbuildable here), so the fold is spelled out over the reference's own
   64-bit hash_combine (math/Hash.hpp:19-28), which is what this pins. */
/* _build_init_mc_id of LBvh (container/Bvh.hpp:177-188; Bvh.hpp itself needs the policy headers): the box-centre /
   unit-cube / morton-code chain over the reference's own AABBBox (geometry/AnalyticLevelSet.h), getBoxCenter /
   getUniformCoord (geometry/BoundingVolumeInterface.hpp:12-31) and morton_code<3> (math/bit/Bits.h:122-140) */
unsigned ref_lbvh_morton(const float *whole, const float *bv) {
  using Box = AABBBox<3, float>;
  using TV = vec<float, 3>;
  Box w{TV{whole[0], whole[1], whole[2]}, TV{whole[3], whole[4], whole[5]}};
  Box b{TV{bv[0], bv[1], bv[2]}, TV{bv[3], bv[4], bv[5]}};
  auto c = b.getBoxCenter();
  auto coord = w.getUniformCoord(c).template cast<f32>();
  return morton_code<3>(coord);
}
int ref_aabb_overlaps(const float *a, const float *q) {
  using Box = AABBBox<3, float>;
  using TV = vec<float, 3>;
  return overlaps(Box{TV{a[0], a[1], a[2]}, TV{a[3], a[4], a[5]}}, Box{TV{q[0], q[1], q[2]}, TV{q[3], q[4], q[5]}}) ? 1 : 0;
}
int ref_collider_resolve(int geometry, int type, const float *p, float s, float dsdt, const float *R, const float *omega, const float *b,
                         const float *dbdt, const float *x, float *v) {
  using TV = vec<float, 3>;
  if (geometry == 0)
    return ref_resolve_with(AnalyticLevelSet<analytic_geometry_e::Plane, float, 3>{TV{p[0], p[1], p[2]}, TV{p[3], p[4], p[5]}}, type, s, dsdt,
                            R, omega, b, dbdt, x, v);
  if (geometry == 1)
    return ref_resolve_with(AnalyticLevelSet<analytic_geometry_e::Cuboid, float, 3>{TV{p[0], p[1], p[2]}, TV{p[3], p[4], p[5]}}, type, s,
                            dsdt, R, omega, b, dbdt, x, v);
  if (geometry == 2)
    return ref_resolve_with(AnalyticLevelSet<analytic_geometry_e::Sphere, float, 3>{TV{p[0], p[1], p[2]}, p[3]}, type, s, dsdt, R, omega, b,
                            dbdt, x, v);
  return ref_resolve_with(AnalyticLevelSet<analytic_geometry_e::Cylinder, float, 3>{TV{p[0], p[1], p[2]}, p[3], p[4], (int)p[5]}, type, s,
                          dsdt, R, omega, b, dbdt, x, v);
}
int ref_hashtable_do_hash(const int *key, int dim) {
  size_t ret = key[0];
  for (int d = 1; d < dim; ++d) hash_combine(ret, key[d]);
  return static_cast<int>(ret);
}
}
