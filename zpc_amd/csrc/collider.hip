// collider.hip -- boundary pass of the MPM sub-step for gfx950: ApplyBoundaryConditionOnGridBlocks over the sparse grid
// (simulation/grid/GridOp.hpp:111-164) with the analytic colliders of geometry/Collider.h.  One thread per (block, cell);
// 4 channels of a node are read (m, v) and 3 written when the node is inside the collider: HBM-trivial next to P2G/G2P.
// Built with -ffp-contract=off: the cuboid / cylinder normals are float central differences with eps = 1e-6 and must round
// like the reference's scalar code (collider_device.hpp).
#include "common.hpp"
#include "bht.hpp"
#include "../../include/zensim_rocm/collider_device.hpp"

namespace zsr {

template <int SIDE>
__global__ __launch_bounds__(256) void apply_boundary_kernel(ColliderDev col, const int *activeKeys, float *grid, size_t nblocks, float dx,
                                                             int kscale) {
  constexpr int NC = SIDE * SIDE * SIDE;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nblocks * NC) return;
  const size_t blk = gid / NC;
  const int cell = (int)(gid % NC);
  float *g = grid + blk * 7 * NC + cell;
  if (!(g[0] > 0.f)) return;  // block(0, cellid) > 0
  const int cc[3] = {cell / (SIDE * SIDE), (cell / SIDE) % SIDE, cell % SIDE};
  float pos[3], vel[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    // (blockkey * side + cellid_to_coord(cellid)) * dx, integer node index converted once (exact below 2^24)
    const int node = activeKeys[3 * blk + d] / kscale * SIDE + cc[d];
    pos[d] = (float)node * dx;
    vel[d] = g[(1 + d) * NC];
  }
  if (col.resolveCollision(pos, vel)) {
#pragma unroll
    for (int d = 0; d < 3; ++d) g[(1 + d) * NC] = vel[d];
  }
}

__global__ __launch_bounds__(256) void collider_resolve_kernel(ColliderDev col, const float *x, float *v, size_t n, int *inside) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
  float u[3] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
  const bool in = col.resolveCollision(p, u);
  if (in) { v[3 * i] = u[0]; v[3 * i + 1] = u[1]; v[3 * i + 2] = u[2]; }
  if (inside) inside[i] = in ? 1 : 0;
}

}  // namespace zsr

using namespace zsr;

extern "C" {

void zs_rocm_collider_init(zs_rocm_collider *c, int geometry, int type, const float *param, int nparam) {
  *c = zs_rocm_collider{};
  c->geometry = geometry;
  c->type = type;
  for (int i = 0; i < 8; ++i) c->param[i] = (param && i < nparam) ? param[i] : 0.f;
  c->s = 1.f;
  c->R[0] = c->R[4] = c->R[8] = 1.f;
}

void zs_rocm_mpm_apply_boundary(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, const zs_rocm_bht_3 *tab, float *grid, size_t nblocks,
                                const zs_rocm_collider *collider) {
  Launch L(pol, "ApplyBoundaryConditionOnGridBlocks");
  if (!nblocks || !collider) return;
  const size_t nc = (size_t)p->side * p->side * p->side;
  const int kscale = p->keyIsOrigin ? p->side : 1;
  const BhtDev t = tab->t.dev();
  if (p->side == 4)
    hipLaunchKernelGGL((apply_boundary_kernel<4>), dim3(ceil_div(nblocks * nc, 256)), dim3(256), 0, L.stream, ColliderDev(*collider),
                       (const int *)t.activeKeys, grid, nblocks, p->dx, kscale);
  else
    hipLaunchKernelGGL((apply_boundary_kernel<8>), dim3(ceil_div(nblocks * nc, 256)), dim3(256), 0, L.stream, ColliderDev(*collider),
                       (const int *)t.activeKeys, grid, nblocks, p->dx, kscale);
}

void zs_rocm_collider_resolve(zs_rocm_policy *pol, const zs_rocm_collider *collider, const float *x, float *v, size_t n, int *inside) {
  Launch L(pol, "collider_resolve");
  if (!n) return;
  hipLaunchKernelGGL(collider_resolve_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, ColliderDev(*collider), x, v, n, inside);
}

}  // extern "C"
