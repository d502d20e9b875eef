// mpm_g2p.hip -- G2PTransfer entry point: zs_rocm_mpm_g2p (see mpm_device.hpp for the kernels)
#include "mpm_device.hpp"

using namespace zsr;

extern "C" {
void zs_rocm_mpm_g2p(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, const float *grid,
                     size_t nblocks, const int *binStart, const unsigned *cellCount, const int *nbr) {
  Launch L(pol, "G2PTransfer");
  if (!ps.n) return;
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  // also evaluate the constitutive model for the next P2G when the particles carry `stress`; otherwise only tell the
  // kernels whether the deformation state is F or J
  const int smodel = ps.stress.base ? p->model : (p->model == ZS_MPM_EQUATION_OF_STATE ? MPM_FLUID_NO_STRESS : -1);
  if (binStart && cellCount && nbr) {
    if (!nblocks) return;
    const unsigned nbins = (unsigned)(nblocks * (p->side == 4 ? 1 : 8));
    (void)nbins;
    int *stale = (int *)L.temp(sizeof(int) * (ps.n + 64));
    int *staleCount = stale + ps.n + 32;
    ZSR_CHECK(hipMemsetAsync(staleCount, 0, sizeof(int), L.stream));
    const int lw = uniform_lane_width(ps, model_uses_logjp(smodel), smodel >= 0);  // (-2 / -1: no stress attribute needed)
    // g2p_packed_kernel: lane = particle, one workgroup per grid block (r06).  Against g2p_binned_kernel (lane = cell, one wave per bin) the
    // G2P launch of the 64 Mi-particle column takes 2.67-2.71 instead of 2.96-3.00 ms and the unfused step 4.23-4.31 instead of 4.46-4.54;
    // the P2G launch behind it has its fast mode at 1.41 instead of 1.355 ms, its slow one at 1.44 either way (profiles/r06_g2p.md).
    // -DZS_G2P_AB measurement builds: ZS_ROCM_G2P_PACKED=0 runs the lane = cell kernel.
#ifdef ZS_G2P_AB
    static const bool packed = [] { const char *e = getenv("ZS_ROCM_G2P_PACKED"); return !e || atoi(e) != 0; }();
#define CALL_G2P_BINNED3(S, M, LWv)                                                                                                  \
  if (packed)                                                                                                                        \
    hipLaunchKernelGGL((g2p_packed_kernel<S, M, LWv>), dim3((unsigned)nblocks), dim3(S == 8 ? 256 : 64), 0, L.stream, mp, pd, t, grid, \
                       binStart, nbr, stale, staleCount);                                                                            \
  else                                                                                                                               \
    hipLaunchKernelGGL((g2p_binned_kernel<S, M, LWv>), dim3(nbins), dim3(64), 0, L.stream, mp, pd, t, grid, binStart, cellCount, nbr, \
                       stale, staleCount);                                                                                           \
  hipLaunchKernelGGL((g2p_stale_kernel<S, M>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, grid, (const int *)stale,                 \
                     (const int *)staleCount)
#else
#define CALL_G2P_BINNED3(S, M, LWv)                                                                                                  \
  hipLaunchKernelGGL((g2p_packed_kernel<S, M, LWv>), dim3((unsigned)nblocks), dim3(S == 8 ? 256 : 64), 0, L.stream, mp, pd, t, grid,   \
                     binStart, nbr, stale, staleCount);                                                                              \
  hipLaunchKernelGGL((g2p_stale_kernel<S, M>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, grid, (const int *)stale,                 \
                     (const int *)staleCount)
#endif
#define CALL_G2P_BINNED(S, M) ZSR_DISPATCH_LW(lw, CALL_G2P_BINNED3, S, M)
    ZSR_DISPATCH_SIDE_SMODEL(p->side, smodel, CALL_G2P_BINNED);
  } else {
#define CALL_G2P_GLOBAL(S, M) \
  hipLaunchKernelGGL((g2p_global_kernel<S, M>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd, t, grid)
    ZSR_DISPATCH_SIDE_SMODEL(p->side, smodel, CALL_G2P_GLOBAL);
  }
}

}  // extern "C"
