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
#include "zensim/simulation/Utils.hpp"         /* LocalArena / make_local_arena, unpack_coord_in_grid(coord, side) */
#include "zensim/math/matrix/MatrixUtils.h"    /* matrixMatrixMultiplication3d */
#include <array>
#include <map>
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

/* ---- P2GTransfer / G2PTransfer as whole functions (simulation/transfer/P2G.hpp:51-125, G2P.hpp:44-83) ---------------------
   The functor headers include ExecutionPolicy.hpp / Structure.hpp / HashTable.hpp (un-vendored magic_enum: unbuildable here), so --
   exactly like Collider::resolveCollision above -- the per-particle bodies are spelled out over the reference's OWN pieces, which do
   build: make_local_arena / LocalArena::{range, coord, diff, weight} and the 2-argument unpack_coord_in_grid
   (simulation/Utils.hpp), compute_stress_* and the *Config structs, lame_parameters, vec, matrixMatrixMultiplication3d.  Only the
   containers are replaced: the partition (HashTable query of a block coordinate) by a std::map over the caller's block keys, the
   Grids block (grid_block(chn, cellid), Structure.hpp:323-333 coord_to_cellid for a power-of-two side: (x << b | y) << b | z) by a
   [nblocks][7][side^3] float array, atomic_add by += under the SequentialExecutionPolicy's particle order 0..n-1. */
struct RefGrid {
  std::map<std::array<int, 3>, int> blocks;
  float *data;
  int side, ncell, bits;
  float *chan(const vec<int, 3> &blk, int chn) const {
    auto it = blocks.find({blk[0], blk[1], blk[2]});
    return it == blocks.end() ? nullptr : data + ((size_t)it->second * 7 + chn) * ncell;
  }
  int cellid(const vec<int, 3> &loc) const { return (((loc[0] << bits) | loc[1]) << bits) | loc[2]; }
};
static RefGrid make_ref_grid(int side, int nblocks, const int *keys, float *data) {
  RefGrid g;
  g.data = data;
  g.side = side;
  g.ncell = side * side * side;
  g.bits = side == 4 ? 2 : 3;
  for (int b = 0; b < nblocks; ++b) g.blocks[{keys[3 * b], keys[3 * b + 1], keys[3 * b + 2]}] = b;
  return g;
}
template <class model_t>
static int ref_p2g_with(const model_t &model, float dx, float dt, const RefGrid &grids, size_t n, const float *massA, const float *posA,
                        const float *velA, const float *CA, const float *FA, float *logJpA) {
  using vec3 = vec<float, 3>;
  using vec9 = vec<float, 9>;
  int missed = 0;
  float const dx_inv = 1.0f / dx; /* dxinv(): static_cast<decltype(grids._dx)>(1.0) / grids._dx */
  float const D_inv = 4.f * dx_inv * dx_inv;
  for (size_t parid = 0; parid < n; ++parid) {
    vec3 local_pos{posA[3 * parid], posA[3 * parid + 1], posA[3 * parid + 2]};
    vec3 vel{velA[3 * parid], velA[3 * parid + 1], velA[3 * parid + 2]};
    float mass = massA[parid];
    vec9 contrib{vec9::zeros()}, C{};
    for (int d = 0; d < 9; ++d) C[d] = CA[9 * parid + d];
    if constexpr (is_same_v<model_t, EquationOfStateConfig>) {
      float J = FA[9 * parid]; /* the J attribute is handed over in component 0 of the F slot */
      float vol = model.volume * J;
      float pressure = model.bulk;
      {
        float J2 = J * J;
        float J4 = J2 * J2;
        pressure = pressure * (1 / (J * J2 * J4) - 1);
      }
      contrib[0] = ((C[0] + C[0]) * model.viscosity - pressure) * vol;
      contrib[1] = (C[1] + C[3]) * model.viscosity * vol;
      contrib[2] = (C[2] + C[6]) * model.viscosity * vol;
      contrib[3] = (C[3] + C[1]) * model.viscosity * vol;
      contrib[4] = ((C[4] + C[4]) * model.viscosity - pressure) * vol;
      contrib[5] = (C[5] + C[7]) * model.viscosity * vol;
      contrib[6] = (C[6] + C[2]) * model.viscosity * vol;
      contrib[7] = (C[7] + C[5]) * model.viscosity * vol;
      contrib[8] = ((C[8] + C[8]) * model.viscosity - pressure) * vol;
    } else {
      const auto [mu, lambda] = lame_parameters(model.E, model.nu);
      vec9 F{};
      for (int d = 0; d < 9; ++d) F[d] = FA[9 * parid + d];
      if constexpr (is_same_v<model_t, FixedCorotatedConfig>) {
        compute_stress_fixedcorotated(model.volume, mu, lambda, F, contrib);
      } else if constexpr (is_same_v<model_t, VonMisesFixedCorotatedConfig>) {
        compute_stress_vonmisesfixedcorotated(model.volume, mu, lambda, model.yieldStress, F, contrib);
      } else {
        float logJp = logJpA[parid];
        if constexpr (is_same_v<model_t, DruckerPragerConfig>) {
          compute_stress_sand(model.volume, mu, lambda, model.cohesion, model.beta, model.yieldSurface, model.volumeCorrection, logJp, F,
                              contrib);
        } else if constexpr (is_same_v<model_t, NACCConfig>) {
          compute_stress_nacc(model.volume, mu, lambda, model.bulk(), model.xi, model.beta, model.Msqr(), model.hardeningOn, logJp, F,
                              contrib);
        }
        logJpA[parid] = logJp;
      }
    }
    contrib = contrib * -dt * D_inv;
    auto arena = make_local_arena((float)dx, local_pos);
    for (auto loc : arena.range()) {
      auto [blockCoord, local_index] = unpack_coord_in_grid(arena.coord(loc), grids.side);
      auto xixp = arena.diff(loc);
      float W = arena.weight(loc);
      const auto cellid = grids.cellid(local_index);
      float *m = grids.chan(blockCoord, 0);
      if (!m) {
        ++missed;
        continue;
      }
      m[cellid] += mass * W;
      for (int d = 0; d != 3; ++d) {
        grids.chan(blockCoord, 1 + d)[cellid] += W * mass * (vel[d] + (C[d] * xixp[0] + C[3 + d] * xixp[1] + C[6 + d] * xixp[2]));
        grids.chan(blockCoord, 3 + 1 + d)[cellid] += (contrib[d] * xixp[0] + contrib[3 + d] * xixp[1] + contrib[6 + d] * xixp[2]) * W;
      }
    }
  }
  return missed;
}
template <class model_t>
static int ref_g2p_with(const model_t &model, float dx, float dt, const RefGrid &grids, size_t n, float *posA, float *velA, float *CA,
                        float *FA) {
  using value_type = float;
  using vec3 = vec<value_type, 3>;
  using vec9 = vec<value_type, 9>;
  int missed = 0;
  value_type const dx_inv = (value_type)1 / dx;
  value_type const D_inv = 4.f * dx_inv * dx_inv;
  for (size_t parid = 0; parid < n; ++parid) {
    vec3 pos{posA[3 * parid], posA[3 * parid + 1], posA[3 * parid + 2]};
    vec3 vel{vec3::zeros()};
    vec9 C{vec9::zeros()};
    auto arena = make_local_arena(dx, pos);
    for (auto loc : arena.range()) {
      auto [blockCoord, local_index] = unpack_coord_in_grid(arena.coord(loc), grids.side);
      auto xixp = arena.diff(loc);
      float W = arena.weight(loc);
      const auto cellid = grids.cellid(local_index);
      if (!grids.chan(blockCoord, 1)) {
        ++missed;
        continue;
      }
      vec3 vi{grids.chan(blockCoord, 1)[cellid], grids.chan(blockCoord, 2)[cellid], grids.chan(blockCoord, 3)[cellid]}; /* pack<3>(1, cellid) */
      vel += vi * W;
      for (int d = 0; d < 9; ++d) C[d] += W * vi(d % 3) * xixp(d / 3) * D_inv;
    }
    pos += vel * dt;
    if constexpr (is_same_v<model_t, EquationOfStateConfig>) {
      float J = FA[9 * parid];
      J = (1 + (C[0] + C[4] + C[8]) * dt) * J;
      FA[9 * parid] = J;
    } else {
      vec9 oldF{}, tmp{}, F{};
      for (int d = 0; d < 9; ++d) oldF[d] = FA[9 * parid + d];
      for (int d = 0; d < 9; ++d) tmp(d) = C[d] * dt + ((d & 0x3) ? 0.f : 1.f);
      matrixMatrixMultiplication3d(tmp.data(), oldF.data(), F.data());
      for (int d = 0; d < 9; ++d) FA[9 * parid + d] = F[d];
    }
    for (int d = 0; d < 3; ++d) posA[3 * parid + d] = pos[d];
    for (int d = 0; d < 3; ++d) velA[3 * parid + d] = vel[d];
    for (int d = 0; d < 9; ++d) CA[9 * parid + d] = C[d];
  }
  return missed;
}
/* prm: {volume, E, nu, cohesion, beta, yieldSurface, volumeCorrection, yieldStress, xi, fa, hardeningOn, bulk, viscosity} */
template <class Fn> static int ref_with_model(int model, const float *prm, Fn &&fn) {
  if (model == 0) {
    FixedCorotatedConfig c{};
    c.volume = prm[0], c.E = prm[1], c.nu = prm[2];
    return fn(c);
  } else if (model == 1) {
    DruckerPragerConfig c{};
    c.volume = prm[0], c.E = prm[1], c.nu = prm[2], c.cohesion = prm[3], c.beta = prm[4], c.yieldSurface = prm[5], c.volumeCorrection = prm[6] != 0;
    return fn(c);
  } else if (model == 2) {
    VonMisesFixedCorotatedConfig c{};
    c.volume = prm[0], c.E = prm[1], c.nu = prm[2], c.yieldStress = prm[7];
    return fn(c);
  } else if (model == 3) {
    NACCConfig c{};
    c.volume = prm[0], c.E = prm[1], c.nu = prm[2], c.beta = prm[4], c.xi = prm[8], c.fa = prm[9], c.hardeningOn = prm[10] != 0;
    return fn(c);
  }
  EquationOfStateConfig c{};
  c.volume = prm[0], c.bulk = prm[11], c.viscosity = prm[12];
  return fn(c);
}

extern "C" {

/* grid: [nblocks][7][side^3] accumulated in place; returns the number of stencil nodes whose block is not in blockKeys (must be 0) */
int ref_mpm_p2g(int model, const float *prm, float dx, float dt, int side, int nblocks, const int *blockKeys, float *grid, size_t n,
                const float *mass, const float *pos, const float *vel, const float *C, const float *F, float *logJp) {
  RefGrid g = make_ref_grid(side, nblocks, blockKeys, grid);
  return ref_with_model(model, prm, [&](const auto &m) { return ref_p2g_with(m, dx, dt, g, n, mass, pos, vel, C, F, logJp); });
}
int ref_mpm_g2p(int model, const float *prm, float dx, float dt, int side, int nblocks, const int *blockKeys, const float *grid, size_t n,
                float *pos, float *vel, float *C, float *F) {
  RefGrid g = make_ref_grid(side, nblocks, blockKeys, const_cast<float *>(grid));
  return ref_with_model(model, prm, [&](const auto &m) { return ref_g2p_with(m, dx, dt, g, n, pos, vel, C, F); });
}

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
