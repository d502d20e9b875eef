// mpm_fused.hip -- fused G2P2G entry points: zs_rocm_mpm_g2p2g_range / zs_rocm_mpm_g2p2g (see mpm_device.hpp for the kernels)
#include "mpm_device.hpp"

using namespace zsr;

extern "C" {
// G2P (from gridA) + P2G (into gridB, zeroed by the caller) in one pass; `particles.stress` must be present (it carries the
// state of the particles that take the exact path).  writeAll != 0 also stores v, C and P F^T vol of every particle.
// blocks [blockBegin, blockEnd) only.  Multi-GPU step (bench.py): the partition is numbered with the blocks near a rank boundary
// first; their range is launched first, its ghost-block sums travel on a second stream while the interior range computes.
// An interior block's exact-path particles must not reach a shared block: *driftFlag is set to 1 when a particle handled
// by the exact path sits more than one bin away from the bin it is stored in (re-bin more often then).
static int g2p2g_impl(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_particles *in, const int *order,
                      const zs_rocm_bht_3 *tab, const float *gridA, float *gridB, size_t nblocks, const int *binStart,
                      const unsigned *cellCount, const int *nbr, int writeAll, size_t blockBegin, size_t blockEnd, int *driftFlag) {
  if (!ps.n || !nblocks) return 0;
  if (!ps.stress.base || !binStart || !cellCount || !nbr) {
    fprintf(stderr, "[zs_rocm] g2p2g needs binned particles and the `stress` attribute\n");
    return -1;
  }
  long long inDelta = 0;
  if (order) {
    // the inputs come from a second particle buffer of the same layout: one element offset for every attribute
    if (!in || !in->pos.base) return -1;
    inDelta = (const float *)in->pos.base - (const float *)ps.pos.base;
    const zs_rocm_attr *a[4] = {&in->mass, &in->pos, &in->F, &in->logJp}, *b[4] = {&ps.mass, &ps.pos, &ps.F, &ps.logJp};
    for (int k = 0; k < 4; ++k) {
      if (!b[k]->base) continue;
      if (!a[k]->base || (const float *)a[k]->base - (const float *)b[k]->base != inDelta || a[k]->idx != b[k]->idx ||
          a[k]->numTileBits != b[k]->numTileBits || a[k]->tileMask != b[k]->tileMask || a[k]->numChns != b[k]->numChns) {
        fprintf(stderr, "[zs_rocm] g2p2g re-ordering step: input and output particles must share one layout\n");
        return -1;
      }
    }
  }
  if (blockEnd > nblocks) blockEnd = nblocks;
  if (blockBegin >= blockEnd) return 0;
  Launch L(pol, "G2P2GTransfer");
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  const unsigned bpb = p->side == 4 ? 1u : 8u;
  const unsigned nbins = (unsigned)((blockEnd - blockBegin) * bpb);
  const int binBase = (int)(blockBegin * bpb);
  int *staleG = (int *)L.temp(sizeof(int) * (ps.n + 64));
  int *staleP = (int *)L.temp(sizeof(int) * (ps.n + 64));
  int *counts = (int *)L.temp(sizeof(int) * 64);
  ZSR_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * 64, L.stream));
  const int lw = uniform_lane_width(ps, model_uses_logjp(p->model), true);
  if (p->model < ZS_MPM_FIXED_COROTATED || p->model > ZS_MPM_EQUATION_OF_STATE) return -1;
  const FusedArgs a{gridA, gridB, binStart, cellCount, nbr, staleG, staleP, counts, driftFlag, nbins, binBase, writeAll, lw, p->model,
                    order, inDelta};
  if (p->side == 4) g2p2g_launch_side<4>(L, mp, pd, t, a);
  else g2p2g_launch_side<8>(L, mp, pd, t, a);
  return 0;
}

// G2P (from gridA) + P2G (into gridB, zeroed by the caller) in one pass; `particles.stress` must be present (it carries the
// state of the particles that take the exact path).  writeAll != 0 also stores v, C and P F^T vol of every particle.
// blocks [blockBegin, blockEnd) only.  Multi-GPU step (bench.py): the partition is numbered with the blocks near a rank boundary
// first; their range is launched first, its ghost-block sums travel on a second stream while the interior range computes.
int zs_rocm_mpm_g2p2g_range(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, const float *gridA,
                            float *gridB, size_t nblocks, const int *binStart, const unsigned *cellCount, const int *nbr, int writeAll,
                            size_t blockBegin, size_t blockEnd, int *driftFlag) {
  return g2p2g_impl(pol, p, ps, nullptr, nullptr, tab, gridA, gridB, nblocks, binStart, cellCount, nbr, writeAll, blockBegin, blockEnd,
                    driftFlag);
}
// The re-ordering step: binStart / cellCount describe the NEW binned order (zs_rocm_mpm_bin_particles on the current positions);
// slot i of that order is read from slot order[i] of `particlesIn` and everything is written to slot i of `particlesOut`.
int zs_rocm_mpm_g2p2g_reorder_range(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles particlesOut,
                                    zs_rocm_particles particlesIn, const int *order, const zs_rocm_bht_3 *tab, const float *gridA,
                                    float *gridB, size_t nblocks, const int *binStart, const unsigned *cellCount, const int *nbr,
                                    int writeAll, size_t blockBegin, size_t blockEnd, int *driftFlag) {
  if (!order || writeAll) return -1;  // a re-ordering step does not materialise v, C, stress of every particle
  return g2p2g_impl(pol, p, particlesOut, &particlesIn, order, tab, gridA, gridB, nblocks, binStart, cellCount, nbr, writeAll, blockBegin,
                    blockEnd, driftFlag);
}

int zs_rocm_mpm_g2p2g(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, const float *gridA,
                      float *gridB, size_t nblocks, const int *binStart, const unsigned *cellCount, const int *nbr, int writeAll) {
  return zs_rocm_mpm_g2p2g_range(pol, p, ps, tab, gridA, gridB, nblocks, binStart, cellCount, nbr, writeAll, 0, nblocks, nullptr);
}

}  // extern "C"
