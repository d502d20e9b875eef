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
#include <cstring>
#include <limits>
#include <map>
#include <random>
#include <vector>

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
/* ---- P2C2GTransfer / G2C2PTransfer as whole functions (simulation/transfer/P2C2G.hpp:53-189, G2C2P.hpp:59-135) -----------------
   Same treatment as P2G / G2P above: the per-(block, cell) functor bodies spelled over the reference's own vec, lower_trunc, zs::abs,
   lame_parameters, compute_stress_*, the *Config structs and the 2-argument unpack_coord_in_grid.  Replaced: the partition and the grid
   by RefGrid; IndexBuckets (bucketCoord = lower_trunc(pos / dx + 0), IndexBuckets.hpp:101-107; offsets / indices as the sequential
   policy builds them: ascending particle ids per bucket) by a std::map from cell coordinate to id list; the Collapse{nblocks, side^3}
   launch by the loop over (block, cell) in that order; atomic_add by +=. */
struct RefBuckets {
  std::map<std::array<int, 3>, std::vector<int>> cells;
  const std::vector<int> *find(const vec<int, 3> &c) const {
    auto it = cells.find({c[0], c[1], c[2]});
    return it == cells.end() ? nullptr : &it->second;
  }
};
static RefBuckets make_ref_buckets(float dx, size_t n, const float *posA) {
  RefBuckets b;
  const float dxinv = 1.0f / dx;
  for (size_t i = 0; i < n; ++i) {
    std::array<int, 3> c;
    for (int d = 0; d < 3; ++d) c[d] = lower_trunc(posA[3 * i + d] * dxinv + 0.f, number_c<int>);
    b.cells[c].push_back((int)i);
  }
  return b;
}
template <class model_t>
static int ref_p2c2g_with(const model_t &model, float dx, float dt, const RefGrid &grids, int nblocks, const int *blockKeys, size_t n,
                          const float *massA, const float *posA, const float *velA, const float *BA, const float *FA, float *logJpA) {
  using value_type = float;
  constexpr int dim = 3;
  using TV = vec<value_type, dim>;
  using TM = vec<value_type, dim * dim>;
  using IV = vec<int, dim>;
  const RefBuckets buckets = make_ref_buckets(dx, n, posA);
  value_type const dx_inv = static_cast<float>(1.0) / dx;
  int missed = 0;
  for (int blockid = 0; blockid < nblocks; ++blockid)
    for (int cellid = 0; cellid < grids.ncell; ++cellid) {
      value_type m_c{(value_type)0};
      TV mv_c{TV::zeros()};
      TM QDinv_c{TM::zeros()};
      TV QDinvXp_c{TV::zeros()};
      /* cellid_to_coord, Structure.hpp:334-343 (power-of-two side): x = cellid >> 2b, y = (cellid >> b) & mask, z = cellid & mask */
      IV coord{blockKeys[3 * blockid] * grids.side + (cellid >> (2 * grids.bits)),
               blockKeys[3 * blockid + 1] * grids.side + ((cellid >> grids.bits) & (grids.side - 1)),
               blockKeys[3 * blockid + 2] * grids.side + (cellid & (grids.side - 1))};
      auto posc = (coord + (value_type)0.5) * dx;
      auto checkInKernelRange = [&posc, dx](auto &&posp) -> bool {
        for (int d = 0; d != dim; ++d)
          if (zs::abs(posp[d] - posc[d]) > dx) return false;
        return true;
      };
      coord = coord - 1;  /// move to base coord
      for (int i0 = 0; i0 < 3; ++i0)
        for (int i1 = 0; i1 < 3; ++i1)
          for (int i2 = 0; i2 < 3; ++i2) { /* ndrange<dim>(3): the last index runs fastest */
            const std::vector<int> *ids = buckets.find(coord + IV{i0, i1, i2});
            if (!ids) continue;
            for (int parid : *ids) {
              TV posp{posA[3 * parid], posA[3 * parid + 1], posA[3 * parid + 2]};
              if (!checkInKernelRange(posp)) continue;
              TV Dinv{};
              for (int d = 0; d != dim; ++d) {
                Dinv[d] = posp[d] - lower_trunc(posp[d] * dx_inv + (value_type)0.5) * dx;
                Dinv[d] = ((value_type)2 / (dx * dx - 2 * Dinv[d] * Dinv[d]));
              }
              TV vel{velA[3 * parid], velA[3 * parid + 1], velA[3 * parid + 2]};
              auto mass = massA[parid];
              TM C{};
              for (int d = 0; d != dim * dim; ++d) C[d] = BA[9 * parid + d];
              for (int d = 0; d != dim * dim; ++d) C[d] *= Dinv[d / dim];
              TM contrib{};
              if constexpr (is_same_v<model_t, EquationOfStateConfig>) {
                float J = FA[9 * parid];
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
                TM F{};
                for (int d = 0; d < 9; ++d) F[d] = FA[9 * parid + d];
                if constexpr (is_same_v<model_t, FixedCorotatedConfig>) {
                  compute_stress_fixedcorotated(model.volume, mu, lambda, F, contrib);
                } else if constexpr (is_same_v<model_t, VonMisesFixedCorotatedConfig>) {
                  compute_stress_vonmisesfixedcorotated(model.volume, mu, lambda, model.yieldStress, F, contrib);
                } else {
                  float logJp = logJpA[parid];
                  if constexpr (is_same_v<model_t, DruckerPragerConfig>) {
                    compute_stress_sand(model.volume, mu, lambda, model.cohesion, model.beta, model.yieldSurface, model.volumeCorrection, logJp,
                                        F, contrib);
                  } else if constexpr (is_same_v<model_t, NACCConfig>) {
                    compute_stress_nacc(model.volume, mu, lambda, model.bulk(), model.xi, model.beta, model.Msqr(), model.hardeningOn, logJp, F,
                                        contrib);
                  }
                  logJpA[parid] = logJp;
                }
              }
              for (int d = 0; d != dim * dim; ++d) contrib[d] *= Dinv[d / dim] * -dt;
              contrib += C * mass;
              auto xcxp = posc - posp;
              value_type Wpc = 1.f;
              auto diff = xcxp * dx_inv;
              for (int d = 0; d != dim; ++d) Wpc *= ((value_type)1. - zs::abs(diff[d]));
              m_c += mass * Wpc;
              for (int d = 0; d != dim; ++d) {
                mv_c[d] += mass * vel[d] * Wpc;
                QDinvXp_c[d] += (contrib[d] * posp[0] + contrib[3 + d] * posp[1] + contrib[6 + d] * posp[2]) * Wpc;
              }
              for (int d = 0; d != dim * dim; ++d) QDinv_c[d] += contrib[d] * Wpc;
            }
          }
      /// stage 2 (c -> i)
      coord = coord + 1;  /// move to base coord
      for (int i0 = 0; i0 < 2; ++i0)
        for (int i1 = 0; i1 < 2; ++i1)
          for (int i2 = 0; i2 < 2; ++i2) {
            const auto coordi = coord + IV{i0, i1, i2};
            const auto posi = coordi * dx;
            auto [blockCoord, local_index] = unpack_coord_in_grid(coordi, grids.side);
            float *m = grids.chan(blockCoord, 0);
            if (!m) { /* blockno < 0 */
              ++missed;
              continue;
            }
            constexpr value_type Wci = 1. / 8;
            const auto cell = grids.cellid(local_index);
            m[cell] += m_c * Wci;
            for (int d = 0; d != dim; ++d)
              grids.chan(blockCoord, 1 + d)[cell]
                  += (mv_c[d] + ((QDinv_c[d] * posi[0] + QDinv_c[3 + d] * posi[1] + QDinv_c[6 + d] * posi[2]) - QDinvXp_c[d])) * Wci;
          }
    }
  return missed;
}
/* G2C2PTransfer: v_p and B_p are accumulated (the caller zeroes them: PreG2C2PTransfer, G2C2P.hpp:215-218) */
static int ref_g2c2p_run(float dx, const RefGrid &grids, int nblocks, const int *blockKeys, size_t n, const float *posA, float *velA, float *BA) {
  using value_type = float;
  constexpr int dim = 3;
  using TV = vec<value_type, dim>;
  using TM = vec<value_type, dim * dim>;
  using IV = vec<int, dim>;
  const RefBuckets buckets = make_ref_buckets(dx, n, posA);
  value_type const dx_inv = (value_type)1 / dx;
  int missed = 0;
  for (int blockid = 0; blockid < nblocks; ++blockid)
    for (int cellid = 0; cellid < grids.ncell; ++cellid) {
      IV coord{blockKeys[3 * blockid] * grids.side + (cellid >> (2 * grids.bits)),
               blockKeys[3 * blockid + 1] * grids.side + ((cellid >> grids.bits) & (grids.side - 1)),
               blockKeys[3 * blockid + 2] * grids.side + (cellid & (grids.side - 1))};
      TV v_c{TV::zeros()};
      TM v_cross_x_c{TM::zeros()};
      for (int i0 = 0; i0 < 2; ++i0)
        for (int i1 = 0; i1 < 2; ++i1)
          for (int i2 = 0; i2 < 2; ++i2) {
            const auto coordi = coord + IV{i0, i1, i2};
            const auto posi = coordi * dx;
            auto [blockCoord, local_index] = unpack_coord_in_grid(coordi, grids.side);
            if (!grids.chan(blockCoord, 1)) {
              ++missed;
              continue;
            }
            value_type W = 1. / 8;
            const auto cell = grids.cellid(local_index);
            TV v_i{grids.chan(blockCoord, 1)[cell], grids.chan(blockCoord, 2)[cell], grids.chan(blockCoord, 3)[cell]};
            v_c += v_i * W;
            for (int d = 0; d < dim * dim; ++d) v_cross_x_c[d] += W * v_i(d % 3) * posi(d / 3);
          }
      auto posc = (coord + (value_type)0.5) * dx;
      auto checkInKernelRange = [&posc, dx](auto &&posp) -> bool {
        for (int d = 0; d != dim; ++d)
          if (zs::abs(posp[d] - posc[d]) > dx) return false;
        return true;
      };
      coord = coord - 1;
      for (int i0 = 0; i0 < 3; ++i0)
        for (int i1 = 0; i1 < 3; ++i1)
          for (int i2 = 0; i2 < 3; ++i2) {
            const std::vector<int> *ids = buckets.find(coord + IV{i0, i1, i2});
            if (!ids) continue;
            for (int parid : *ids) {
              TV posp{posA[3 * parid], posA[3 * parid + 1], posA[3 * parid + 2]};
              if (!checkInKernelRange(posp)) continue;
              auto xcxp = posc - posp;
              value_type W = 1.f;
              auto diff = xcxp * dx_inv;
              for (int d = 0; d != dim; ++d) {
                const auto xabs = zs::abs(diff[d]);
                if (xabs <= 1)
                  W *= ((value_type)1. - xabs);
                else
                  W *= 0.f;
              }
              for (int d = 0; d != dim; ++d) velA[3 * parid + d] += v_c[d] * W;
              for (int d = 0; d != dim * dim; ++d) BA[9 * parid + d] += W * (v_cross_x_c[d] - v_c(d % dim) * posp(d / dim));
            }
          }
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
/* P2C2GTransfer over every cell of the partition; returns the number of (cell, node) pairs whose node block is not in blockKeys */
int ref_mpm_p2c2g(int model, const float *prm, float dx, float dt, int side, int nblocks, const int *blockKeys, float *grid, size_t n,
                  const float *mass, const float *pos, const float *vel, const float *B, const float *F, float *logJp) {
  RefGrid g = make_ref_grid(side, nblocks, blockKeys, grid);
  return ref_with_model(model, prm, [&](const auto &m) { return ref_p2c2g_with(m, dx, dt, g, nblocks, blockKeys, n, mass, pos, vel, B, F, logJp); });
}
int ref_mpm_g2c2p(float dx, int side, int nblocks, const int *blockKeys, const float *grid, size_t n, const float *pos, float *vel, float *B) {
  RefGrid g = make_ref_grid(side, nblocks, blockKeys, const_cast<float *>(grid));
  return ref_g2c2p_run(dx, g, nblocks, blockKeys, n, pos, vel, B);
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

/* ---- bht<int, dim, int, B> and HashTable<int, dim, int> as whole functions under sequential insertion ----------------------------------
   container/Bht.hpp and HashTable.hpp include the execution-policy headers (un-vendored magic_enum: unbuildable here), so -- like
   Collider::resolveCollision and the transfer functors above -- the member bodies are spelled out over the reference's OWN pieces that do
   build: universal_hash_base (py_interop/HashUtils.hpp:7-47) with hash_combine (math/Hash.hpp), storage_key_type_impl (the padded slot,
   HashUtils.hpp:49-87), next_2pow (math/bit/Bits.h), vec; seeds from std::mt19937(2) exactly as universal_hash(std::mt19937 &) draws them
   (Bcht.hpp:39-43, Bht.hpp:165-169).  Under the SequentialExecutionPolicy atomicLoad / atomicSwitchIfEqual / atomic_add reduce to a plain
   load, "if the slot holds the sentinel store the key", and ++cnt. */
template <int dim, int B> struct RefBht {
  using key_t = vec<int, dim>;
  using slot_t = storage_key_type_impl<key_t>;
  static constexpr int threshold = B - 2;                                  /* Bht.hpp:34 */
  size_t tableSize = 0, numBuckets = 0;
  std::vector<slot_t> keys;
  std::vector<int> indices, status, activeKeys;
  int cnt = 0, success = 1;
  universal_hash_base<key_t> hf[3];
  static size_t evaluateTableSize(size_t entryCnt) {                         /* Bht.hpp:154-158 */
    if (entryCnt == 0) return 0;
    size_t n = next_2pow(entryCnt) * 2;
    return n + (B - n % B);
  }
  static key_t sentinel() {                                                  /* deduce_key_sentinel, Bht.hpp:126-131 */
    int v = 0;
    for (int i = 0; i != (int)sizeof(int); ++i) v = (v << 8) | 0x3f;
    return key_t::constant(v);
  }
  explicit RefBht(size_t nExpected) {
    tableSize = evaluateTableSize(nExpected);
    numBuckets = tableSize / B;
    keys.resize(tableSize);
    std::memset((void *)keys.data(), 0x3f, sizeof(slot_t) * tableSize);      /* Table::reset, Bht.hpp:108-112 */
    indices.assign(tableSize, 0);
    status.assign(tableSize, -1);
    activeKeys.assign(tableSize * dim, 0);
    unsigned p[6];
    ref_bht_hash_params(p);
    for (int f = 0; f < 3; ++f) hf[f] = universal_hash_base<key_t>{p[2 * f], p[2 * f + 1]};
  }
  int insert(const key_t &key) {                                             /* BHTView::insert, host execution, Bht.hpp:612-664 */
    if (numBuckets == 0) return std::numeric_limits<int>::lowest();
    const key_t key_sentinel_v = sentinel();
    int iter = 0, load = 0;
    size_t bucketOffset = hf[0](key) % numBuckets * B;
    while (iter < 3) {
      for (; load != B; ++load) {
        key_t curKey = keys[bucketOffset + load].val;
        if (curKey == key) {
          load = -1;
          break;
        } else if (curKey == key_sentinel_v)
          break;
      }
      if (load < 0)
        return -1;
      else if (load <= threshold) {
        if (keys[bucketOffset + load].val == key_sentinel_v) {                /* atomicSwitchIfEqual */
          keys[bucketOffset + load].val = key;
          int localno = cnt++;
          indices[bucketOffset + load] = localno;
          for (int d = 0; d < dim; ++d) activeKeys[(size_t)localno * dim + d] = key[d];
          if (localno >= (int)tableSize - 20) {
            success = 0;
            localno = std::numeric_limits<int>::lowest();
          }
          return localno;
        }
      } else {
        ++iter;
        load = 0;
        if (iter == 1)
          bucketOffset = hf[1](key) % numBuckets * B;
        else if (iter == 2)
          bucketOffset = hf[2](key) % numBuckets * B;
        else
          break;
      }
    }
    success = 0;
    return std::numeric_limits<int>::lowest();
  }
  int query(const key_t &key) const {                                        /* Bht.hpp:667-698 */
    if (numBuckets == 0) return -1;
    int loc = 0;
    size_t bucketOffset = hf[0](key) % numBuckets * B;
    for (int iter = 0; iter < 3;) {
      for (loc = 0; loc != B; ++loc)
        if (keys[bucketOffset + loc].val == key) break;
      if (loc != B) return indices[bucketOffset + loc];
      ++iter;
      if (iter == 1)
        bucketOffset = hf[1](key) % numBuckets * B;
      else if (iter == 2)
        bucketOffset = hf[2](key) % numBuckets * B;
    }
    return -1;
  }
};
template <int dim, int B>
static void ref_bht_seq_run(size_t nExpected, const int *keysIn, size_t n, int *ret, int *keysTable, int *indices, int *status, int *activeKeys,
                            int *cntSuccess, const int *queries, size_t nq, int *qret) {
  RefBht<dim, B> t(nExpected);
  for (size_t i = 0; i < n; ++i) {
    typename RefBht<dim, B>::key_t k;
    for (int d = 0; d < dim; ++d) k[d] = keysIn[i * dim + d];
    ret[i] = t.insert(k);
  }
  std::memcpy(keysTable, (const void *)t.keys.data(), sizeof(typename RefBht<dim, B>::slot_t) * t.tableSize);
  std::memcpy(indices, t.indices.data(), sizeof(int) * t.tableSize);
  std::memcpy(status, t.status.data(), sizeof(int) * t.tableSize);
  std::memcpy(activeKeys, t.activeKeys.data(), sizeof(int) * (size_t)t.cnt * dim);
  cntSuccess[0] = t.cnt;
  cntSuccess[1] = t.success;
  for (size_t i = 0; i < nq; ++i) {
    typename RefBht<dim, B>::key_t k;
    for (int d = 0; d < dim; ++d) k[d] = queries[i * dim + d];
    qret[i] = t.query(k);
  }
}

/* HashTable<int, dim, int>: size next_2pow(n) * 16 (HashTable.hpp:88-91), hash = hash_combine fold of the raw coordinates (do_hash,
   :496-500), linear probing with stride 127 (insert, host execution, :383-400; query :454-463 incl. its `>` wrap) */
template <int dim> struct RefHashTable {
  using key_t = vec<int, dim>;
  int tableSize = 0, cnt = 0;
  std::vector<key_t> keys;
  std::vector<int> indices, status, activeKeys;
  explicit RefHashTable(size_t nExpected) {
    tableSize = nExpected ? (int)(next_2pow(nExpected) * 16) : 0;
    keys.assign(tableSize, key_t::constant(std::numeric_limits<int>::max()));   /* key_scalar_sentinel_v */
    indices.assign(tableSize, -1);
    status.assign(tableSize, -1);
    activeKeys.assign((size_t)tableSize * dim, 0);
  }
  static int do_hash(const key_t &key) {
    size_t ret = key[0];
    for (int d = 1; d < dim; ++d) hash_combine(ret, key[d]);
    return static_cast<int>(ret);
  }
  int insert(const key_t &key) {
    const key_t key_sentinel_v = key_t::constant(std::numeric_limits<int>::max());
    int hashedentry = (do_hash(key) % tableSize + tableSize) % tableSize;
    auto cas = [&](int e) {                                                   /* atomicKeyCAS, host: :546-563 */
      key_t stored = keys[e];
      if (stored == key_sentinel_v) keys[e] = key;
      return stored;
    };
    key_t storedKey = cas(hashedentry);
    for (; !(storedKey == key_sentinel_v || storedKey == key);) {
      hashedentry = (hashedentry + 127) % tableSize;
      storedKey = cas(hashedentry);
    }
    if (storedKey == key_sentinel_v) {
      int localno = cnt++;
      indices[hashedentry] = localno;
      for (int d = 0; d < dim; ++d) activeKeys[(size_t)localno * dim + d] = key[d];
      return localno;
    }
    return -1;
  }
  int query(const key_t &key) const {
    int hashedentry = (do_hash(key) % tableSize + tableSize) % tableSize;
    while (true) {
      if (key == keys[hashedentry]) return indices[hashedentry];
      if (indices[hashedentry] == -1) return -1;
      hashedentry += 127;
      if (hashedentry > tableSize) hashedentry = hashedentry % tableSize;
    }
  }
};

extern "C" {
size_t ref_bht_table_size_b(size_t n, int B) { return B == 32 ? RefBht<3, 32>::evaluateTableSize(n) : RefBht<3, 16>::evaluateTableSize(n); }
/* keysTable: [tableSize][next_2pow(dim)] ints (the padded slots, byte for byte); activeKeys: [cnt][dim]; cntSuccess: {cnt, success} */
void ref_bht_seq(int dim, int B, size_t nExpected, const int *keysIn, size_t n, int *ret, int *keysTable, int *indices, int *status,
                 int *activeKeys, int *cntSuccess, const int *queries, size_t nq, int *qret) {
#define RB(D, BB) if (dim == D && B == BB) return ref_bht_seq_run<D, BB>(nExpected, keysIn, n, ret, keysTable, indices, status, activeKeys, cntSuccess, queries, nq, qret)
  RB(1, 16); RB(2, 16); RB(3, 16); RB(4, 16); RB(3, 32);
#undef RB
}
size_t ref_hashtable_table_size(size_t n) { return n ? next_2pow(n) * 16 : 0; }
void ref_hashtable_seq(size_t nExpected, const int *keysIn, size_t n, int *ret, int *keysTable /*[tableSize][3]*/, int *indices, int *activeKeys,
                       int *cnt, const int *queries, size_t nq, int *qret) {
  RefHashTable<3> t(nExpected);
  for (size_t i = 0; i < n; ++i) ret[i] = t.insert(vec<int, 3>{keysIn[3 * i], keysIn[3 * i + 1], keysIn[3 * i + 2]});
  for (int e = 0; e < t.tableSize; ++e)
    for (int d = 0; d < 3; ++d) keysTable[3 * (size_t)e + d] = t.keys[e][d];
  std::memcpy(indices, t.indices.data(), sizeof(int) * t.tableSize);
  std::memcpy(activeKeys, t.activeKeys.data(), sizeof(int) * (size_t)t.cnt * 3);
  *cnt = t.cnt;
  for (size_t i = 0; i < nq; ++i) qret[i] = t.query(vec<int, 3>{queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]});
}
}

/* ---- GridArena (math/curve/InterpolationKernel.hpp:271-560) over a caller-side grid view ------------------------------------------------
   GridArena is a template over the grid view it samples; the reference instantiates it with SparseGridView (geometry/SparseGrid.hpp, which
   needs the execution-policy headers: unbuildable here).  Here the REFERENCE'S OWN GridArena -- its constructors (collocated and staggered,
   base_node + the six weight functions), weight / weightsGradient, arena, isample (xlerp for the linear kernel), minimum, maximum -- is
   instantiated with a dense box of values as the caller's view type: the members GridArena asks of a view (the type aliases, worldToIndex,
   valueOr(false_c, chn, coord, default)) are provided by DenseBoxView below, which is test scaffolding, not a reference header. */
struct DenseBoxView {
  using value_type = float;
  using size_type = size_t;
  using integer_coord_component_type = int;
  using integer_coord_type = vec<int, 3>;
  using coord_component_type = float;
  using coord_type = vec<float, 3>;
  using container_type = int;  /* (not an adaptive grid: is_ag_v<int> is false) */
  static constexpr int dim = 3;
  const float *data;           /* [nchn][ext][ext][ext] */
  int lo[3], ext;
  float dx;
  template <typename VecT> constexpr coord_type worldToIndex(const VecInterface<VecT> &x) const noexcept {
    return coord_type{x[0] / dx, x[1] / dx, x[2] / dx};
  }
  template <typename VecT> constexpr float valueOr(false_type, size_t chn, const VecInterface<VecT> &c, float def) const noexcept {
    int k[3];
    for (int d = 0; d < 3; ++d) {
      k[d] = c[d] - lo[d];
      if (k[d] < 0 || k[d] >= ext) return def;
    }
    return data[((chn * ext + k[0]) * ext + k[1]) * ext + k[2]];
  }
};
template <kernel_e kt, int order>
static void ref_grid_arena_run(const DenseBoxView &gv, const float *X, int f, float def, float *out /* see gen_golden.py */) {
  using Arena = GridArena<const DenseBoxView, kt, order>;
  vec<float, 3> x{X[0], X[1], X[2]};
  Arena ar = f < 0 ? Arena{false_c, &gv, x} : Arena{false_c, &gv, x, f};
  constexpr int W = Arena::width;
  float *o = out;
  for (int d = 0; d < 3; ++d) *o++ = (float)ar.iCorner[d];
  for (int d = 0; d < 3; ++d) *o++ = ar.iLocalPos[d];
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < 4; ++k) *o++ = k < W ? get<0>(ar.weights)(d, k) : 0.f;
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < 4; ++k) {
      if constexpr (order > 0) *o++ = k < W ? get<1>(ar.weights)(d, k) : 0.f;
      else *o++ = 0.f;
    }
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < 4; ++k) {
      if constexpr (order > 1) *o++ = k < W ? get<2>(ar.weights)(d, k) : 0.f;
      else *o++ = 0.f;
    }
  *o++ = ar.isample(0, def);
  *o++ = ar.isample(1, def);
  *o++ = ar.minimum(0);
  *o++ = ar.maximum(1);
  /* weight and weightsGradient of the stencil corner nodes (0,0,0), (W-1, 0, 1 % W) and (1 % W, W-1, W-1) */
  const int locs[3][3] = {{0, 0, 0}, {W - 1, 0, 1 % W}, {1 % W, W - 1, W - 1}};
  for (int l = 0; l < 3; ++l) {
    auto loc = zs::make_tuple(locs[l][0], locs[l][1], locs[l][2]);
    *o++ = ar.weight(loc);
    if constexpr (order > 0) {
      auto g = ar.weightsGradient(loc);
      for (int d = 0; d < 3; ++d) *o++ = g[d];
    } else {
      for (int d = 0; d < 3; ++d) *o++ = 0.f;
    }
  }
}
extern "C" {
/* kt: 0 linear, 1 quadratic, 2 cubic, 3 delta2, 4 delta3, 5 delta4 (kernel_e order); order: derivative order 0..2 (B-splines only);
   f < 0: collocated, else the face whose values sit at -0.5 along axis f % 3; out: 58 floats per point */
int ref_grid_arena(int kt, int order, const float *data, int nchn, const int *lo, int ext, float dx, const float *X, size_t npts, int f, float def,
                   float *out) {
  DenseBoxView gv{data, {lo[0], lo[1], lo[2]}, ext, dx};
  (void)nchn;
  for (size_t i = 0; i < npts; ++i) {
    const float *x = X + 3 * i;
    float *o = out + 58 * i;
#define GA(K, KT, O) if (kt == K && order == O) { ref_grid_arena_run<KT, O>(gv, x, f, def, o); continue; }
    GA(0, kernel_e::linear, 0) GA(0, kernel_e::linear, 1) GA(0, kernel_e::linear, 2)
    GA(1, kernel_e::quadratic, 0) GA(1, kernel_e::quadratic, 1) GA(1, kernel_e::quadratic, 2)
    GA(2, kernel_e::cubic, 0) GA(2, kernel_e::cubic, 1) GA(2, kernel_e::cubic, 2)
    GA(3, kernel_e::delta2, 0) GA(4, kernel_e::delta3, 0) GA(5, kernel_e::delta4, 0)
#undef GA
    return -1;
  }
  return 0;
}
}
