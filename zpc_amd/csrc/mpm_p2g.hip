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
      // m, x, v, C in 16 adjacent channels of one TileVector (lw == 64 already says: same tiles, same channel count, 16-byte aligned rows)
      const bool merged = lw == 64 && (const float *)ps.pos.base == (const float *)ps.mass.base + 64 &&
                          (const float *)ps.vel.base == (const float *)ps.mass.base + 4 * 64 && (const float *)ps.C.base == (const float *)ps.mass.base + 7 * 64;
      const bool aligned16 = ((((uintptr_t)ps.mass.base) | ((uintptr_t)ps.pos.base) | ((uintptr_t)ps.vel.base) | ((uintptr_t)ps.C.base) | ((uintptr_t)ps.stress.base)) & 15u) == 0;
      static const bool tileStreamEnv = [] { const char *e = getenv("ZS_ROCM_P2G_TILE"); return !e || atoi(e) != 0; }();
#ifndef ZS_P2GW_DEPTH
#define ZS_P2GW_DEPTH 1  // rounds of records requested ahead of the one being accumulated
#endif
#define CALL_P2G_WIDE_G(S, LWv, Gv)                                                                                                     \
  hipLaunchKernelGGL((p2g_wide_kernel<S, LWv, ZS_P2GW_DEPTH, Gv>), dim3(nbins / Gv), dim3(64 * Gv), 0, L.stream, mp, pd, t, grid, binStart, cellCount, \
                     nbr, stale, staleCount)
#ifdef ZS_PROBE_P2G  // measurement builds: extra dynamic LDS per workgroup lowers the occupancy (ZS_ROCM_P2G_DYNLDS bytes)
      static const int p2gtDynLds = [] { const char *e = getenv("ZS_ROCM_P2G_DYNLDS"); return e ? atoi(e) : 0; }();
#define P2GT_DYN_LDS p2gtDynLds
#else
#define P2GT_DYN_LDS 0
#endif
#ifndef ZS_P2GT_NB
#define ZS_P2GT_NB 3  // tile buffers per wave of the tile-stream kernel
#endif
#define CALL_P2G_TILE_GM(S, Gv, Mv)                                                                                                     \
  hipLaunchKernelGGL((p2g_tile_kernel<S, ZS_P2GT_NB, Gv, Mv>), dim3(nbins / Gv), dim3(64 * Gv), P2GT_DYN_LDS, L.stream, mp, pd, t, grid, binStart,  \
                     cellCount, nbr, stale, staleCount)
#define CALL_P2G_TILE_G(S, Gv)                                                                                                          \
  do {                                                                                                                                  \
    if (merged) { CALL_P2G_TILE_GM(S, Gv, true); }                                                                                      \
    else { CALL_P2G_TILE_GM(S, Gv, false); }                                                                                            \
  } while (0)
#define CALL_P2G_WIDE(S, M, LWv)                                                                                                       \
  do {                                                                                                                                  \
    if (LWv == 64 && tileStreamEnv && aligned16) {                                                                                                      \
      if (S == 8 && group == 4) { CALL_P2G_TILE_G(8, 4); }                                                                               \
      else if (S == 8 && group == 2) { CALL_P2G_TILE_G(8, 2); }                                                                          \
      else { CALL_P2G_TILE_G(S, 1); }                                                                                                   \
    }                                                                                                                                   \
    else if (S == 8 && group == 4) { CALL_P2G_WIDE_G(8, LWv, 4); }                                                                       \
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

#ifdef ZS_PROBE_P2G  // measurement-only build: read and clear the phase stamps of p2g_wide_kernel
void zs_rocm_p2g_probe(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zsr::g_p2g_probe), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(zsr::g_p2g_probe), z, sizeof(z));
  }
}
#endif

}  // extern "C"
