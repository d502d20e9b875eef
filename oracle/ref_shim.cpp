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
#include <random>

using namespace zs;

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
int ref_hashtable_do_hash(const int *key, int dim) {
  size_t ret = key[0];
  for (int d = 1; d < dim; ++d) hash_combine(ret, key[d]);
  return static_cast<int>(ret);
}
}
