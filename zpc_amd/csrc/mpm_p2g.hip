// mpm_p2g.hip -- P2GTransfer entry point: zs_rocm_mpm_p2g (see mpm_device.hpp for the kernels)
#include "mpm_device.hpp"

using namespace zsr;

extern "C" {
void zs_rocm_mpm_p2g(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, float *grid,
                     size_t nblocks, const int *binStart, const unsigned *cellCount, const int *nbr) {
  Launch L(pol, "P2GTransfer");
  if (!ps.n) return;
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  const int kmodel = ps.stress.base ? MPM_CACHED_STRESS : p->model;  // cached P F^T vol (see zs_rocm_mpm_g2p) or recompute
  if (binStart && cellCount && nbr) {
    if (!nblocks) return;
    const unsigned nbins = (unsigned)(nblocks * (p->side == 4 ? 1 : 8));
    int *stale = (int *)L.temp(sizeof(int) * (ps.n + 64));
    int *staleCount = stale + ps.n + 32;
    ZSR_CHECK(hipMemsetAsync(staleCount, 0, sizeof(int), L.stream));
    const int lw = uniform_lane_width(ps, model_uses_logjp(p->model) && kmodel != MPM_CACHED_STRESS, kmodel == MPM_CACHED_STRESS);
    if (kmodel == MPM_CACHED_STRESS) {  // cached stress: the "wide" kernel, one wave per bin
      // bins per workgroup = waves that share one flush arena (8^3 blocks only; see p2g_wide_kernel).  Measured at 64 Mi particles
      // (profiles/r04_p2g.md): 1.74 / 1.71 / 1.72 ms for 1 / 2 / 4 with 1.73 / -- / 1.04 GB written: the time no longer follows the
      // traffic, so the default is the pair (least coupling between waves).  ZS_ROCM_P2G_GROUP = 1 | 2 | 4 overrides it for A/B runs.
      static const int group = [] { const char *e = getenv("ZS_ROCM_P2G_GROUP"); const int g = e ? atoi(e) : 0; return g == 1 || g == 2 || g == 4 ? g : 2; }();
#define CALL_P2G_WIDE_G(S, LWv, Gv)                                                                                                     \
  hipLaunchKernelGGL((p2g_wide_kernel<S, LWv, 1, Gv>), dim3(nbins / Gv), dim3(64 * Gv), 0, L.stream, mp, pd, t, grid, binStart, cellCount, \
                     nbr, stale, staleCount)
#define CALL_P2G_WIDE(S, M, LWv)                                                                                                       \
  do {                                                                                                                                  \
    if (S == 8 && group == 4) { CALL_P2G_WIDE_G(8, LWv, 4); }                                                                            \
    else if (S == 8 && group == 2) { CALL_P2G_WIDE_G(8, LWv, 2); }                                                                       \
    else { CALL_P2G_WIDE_G(S, LWv, 1); }                                                                                                \
    hipLaunchKernelGGL((p2g_stale_kernel<S, MPM_CACHED_STRESS>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, grid,            \
                       (const int *)stale, (const int *)staleCount);                                                                    \
  } while (0)
      if (p->side == 4) ZSR_DISPATCH_LW(lw, CALL_P2G_WIDE, 4, 0);
      else ZSR_DISPATCH_LW(lw, CALL_P2G_WIDE, 8, 0);
      return;
    }
#define CALL_P2G_BINNED3(S, M, LWv)                                                                                                  \
  hipLaunchKernelGGL((p2g_binned_kernel<S, M, LWv>), dim3(nbins), dim3(64), 0, L.stream, mp, pd, t, grid, binStart, cellCount, nbr,   \
                     stale, staleCount);                                                                                             \
  hipLaunchKernelGGL((p2g_stale_kernel<S, M>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, grid, (const int *)stale,                 \
                     (const int *)staleCount)
#define CALL_P2G_BINNED(S, M) ZSR_DISPATCH_LW(lw, CALL_P2G_BINNED3, S, M)
    ZSR_DISPATCH_SIDE_PURE(p->side, kmodel, CALL_P2G_BINNED);  // kmodel is one of the five models here
  } else {
#define CALL_P2G_GLOBAL(S, M) \
  hipLaunchKernelGGL((p2g_global_kernel<S, M>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd, t, grid)
    ZSR_DISPATCH_SIDE_MODEL(p->side, kmodel, CALL_P2G_GLOBAL);
  }
}

}  // extern "C"
