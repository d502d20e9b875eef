// mpm_fused_impl.hpp -- body of g2p2g_launch_side<S> (included by mpm_fused4.hip and mpm_fused8.hip only)
#pragma once
#include "mpm_device.hpp"

namespace zsr {

template <int S> void g2p2g_launch_side(Launch &L, const MpmDev &mp, const ParticlesDev &pd, const BhtDev &t, const FusedArgs &a) {
  // A/B runs only: ZS_ROCM_G2P2G_VARIANT=1 four-wave kernel; default: role-split, one workgroup per bin
  static const int variant = [] { const char *e = getenv("ZS_ROCM_G2P2G_VARIANT"); return e ? atoi(e) : 0; }();
#define CALL_G2P2G4(SS, M, LWv, WA, RO)                                                                                               \
  hipLaunchKernelGGL((g2p2g_binned_kernel<SS, M, LWv, WA, RO>), dim3(a.nbins), dim3(256), 0, L.stream, mp, pd, t, a.gridA, a.gridB,    \
                     a.binStart, a.cellCount, a.nbr, a.staleG, a.counts, a.staleP, a.counts + 32, a.binBase, a.order, a.inDelta);     \
  hipLaunchKernelGGL((g2p2g_stale_kernel<SS, M>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, a.gridA, a.gridB,             \
                     (const int *)a.staleG, (const int *)a.counts, (const int *)a.staleP, (const int *)(a.counts + 32), a.driftFlag);  \
  hipLaunchKernelGGL((stale_scatter_coop_kernel<SS>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, a.gridB,                  \
                     (const int *)a.staleG, (const int *)a.counts, (const int *)a.staleP, (const int *)(a.counts + 32), a.driftFlag)
// the re-ordering variant exists without writeAll only (a re-ordering step never has to materialise v, C, stress)
// role-split kernel (default) + the same two exact-path kernels
#define CALL_G2P2G_RS(SS, M, LWv, WA)                                                                                                 \
  hipLaunchKernelGGL((g2p2g_rs_kernel<SS, M, LWv, WA>), dim3(a.nbins), dim3(512), 0, L.stream, mp, pd, t, a.gridA, a.gridB,            \
                     a.binStart, a.cellCount, a.nbr, a.staleG, a.counts, a.staleP, a.counts + 32, a.binBase);                          \
  hipLaunchKernelGGL((g2p2g_stale_kernel<SS, M>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, a.gridA, a.gridB,             \
                     (const int *)a.staleG, (const int *)a.counts, (const int *)a.staleP, (const int *)(a.counts + 32), a.driftFlag);  \
  hipLaunchKernelGGL((stale_scatter_coop_kernel<SS>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, a.gridB,                  \
                     (const int *)a.staleG, (const int *)a.counts, (const int *)a.staleP, (const int *)(a.counts + 32), a.driftFlag)
#define CALL_G2P2G3(SS, M, LWv)                                   \
  do {                                                            \
    if (a.order) { CALL_G2P2G4(SS, M, LWv, false, true); }        \
    else if (variant == 1) {                                      \
      if (a.writeAll) { CALL_G2P2G4(SS, M, LWv, true, false); }   \
      else { CALL_G2P2G4(SS, M, LWv, false, false); }             \
    } else if (a.writeAll) { CALL_G2P2G_RS(SS, M, LWv, true); }   \
    else { CALL_G2P2G_RS(SS, M, LWv, false); }                    \
  } while (0)
#define CALL_G2P2G(SS, M) ZSR_DISPATCH_LW(a.lw, CALL_G2P2G3, SS, M)
#ifdef ZS_FUSED_FAST_BUILD  // measurement builds (tools/ablate.sh): DruckerPrager, 64-wide tiles only -- minutes less to compile
  if (a.model != ZS_MPM_DRUCKER_PRAGER || a.lw != 64) { fprintf(stderr, "[zs_rocm] fast build: sand / lane width 64 only\n"); abort(); }
  CALL_G2P2G3(S, ZS_MPM_DRUCKER_PRAGER, 64);
#else
  ZSR_DISPATCH_PURE_(S, a.model, CALL_G2P2G)
#endif
}

}  // namespace zsr
