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
#ifdef ZS_P2G_AB  // measurement builds (tools/ab_build.sh): bins per workgroup / kernel choice / occupancy from the environment
    static const int group = [] { const char *e = getenv("ZS_ROCM_P2G_GROUP"); const int g = e ? atoi(e) : 0; return g == 1 || g == 2 || g == 4 ? g : 2; }();
    static const bool tileStream = [] { const char *e = getenv("ZS_ROCM_P2G_TILE"); return !e || atoi(e) != 0; }();
    static const int p2gtDynLds = [] { const char *e = getenv("ZS_ROCM_P2G_DYNLDS"); return e ? atoi(e) : 0; }();
#define P2GT_DYN_LDS p2gtDynLds
#else
    constexpr int group = 2;
    constexpr bool tileStream = true;
#define P2GT_DYN_LDS 0
#endif
    int lw = uniform_lane_width(ps, model_uses_logjp(p->model) && kmodel != MPM_CACHED_STRESS, kmodel == MPM_CACHED_STRESS);
    // Reference order (no particles.stress: the constitutive update belongs to P2G, P2G.hpp:60-101) as TWO kernels (r06): the update for
    // every particle at full lane occupancy into a stream-ordered temporary of 6 floats per particle (a TileVector<f32, 64> of its own:
    // 1.6 GB at 64 Mi particles), then the tile kernel reading it -- 1.2 + 1.45 ms where p2g_binned_kernel, which runs the 860-instruction
    // update lane = cell at the occupancy of the rounds and accumulates behind it, takes 3.5 ms.  Needs the 64-lane layout; logJp is
    // updated in place exactly as the one-kernel form does.
    const bool twoPass = tileStream && kmodel != MPM_CACHED_STRESS && lw == 64 && p->model >= ZS_MPM_FIXED_COROTATED && p->model <= ZS_MPM_EQUATION_OF_STATE &&
                         (((uintptr_t)ps.mass.base | (uintptr_t)ps.pos.base | (uintptr_t)ps.vel.base | (uintptr_t)ps.C.base) & 15u) == 0;
    if (twoPass) {
      const size_t tiles = (ps.n + 63) / 64;
      float *tmpStress = (float *)L.temp(sizeof(float) * tiles * STRESS_N * 64);
      pd.stress = Port<float>{tmpStress, 0u, 6u, 63u, (uint32_t)STRESS_N};
#define CALL_STRESS_PASS(S, M) hipLaunchKernelGGL((update_stress_kernel<M>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd)
      ZSR_DISPATCH_PURE_(0, p->model, CALL_STRESS_PASS)
    }
    if (kmodel == MPM_CACHED_STRESS || twoPass) {  // cached stress: one wave per bin, the lane's 27 x 7 node sums in registers
      // Two bins (z-neighbours of an 8^3 block) per workgroup share one flush (see p2g_tile_kernel / p2g_wide_kernel): measured at
      // 64 Mi particles 1 / 2 / 4 bins per workgroup 1.67 / 1.45-1.50 / 1.50-1.54 ms (profiles/r06_p2g.md).
      // p2g_tile_kernel needs one TileVector<f32, 64> layout for all attributes and 16-byte aligned channel rows; `merged`: m, x, v, C in
      // 16 adjacent channels (the layout of zpc_amd.mpm and of the reference's particles {m, x, v, C, ...}).  Anything else (32-lane
      // tiles, AoS / mixed iterators) takes p2g_wide_kernel.
      const bool merged = lw == 64 && (const float *)ps.pos.base == (const float *)ps.mass.base + 64 &&
                          (const float *)ps.vel.base == (const float *)ps.mass.base + 4 * 64 && (const float *)ps.C.base == (const float *)ps.mass.base + 7 * 64;
      const bool aligned16 = ((((uintptr_t)ps.mass.base) | ((uintptr_t)ps.pos.base) | ((uintptr_t)ps.vel.base) | ((uintptr_t)ps.C.base) | ((uintptr_t)ps.stress.base)) & 15u) == 0;
#ifndef ZS_P2GT_NB
#define ZS_P2GT_NB 3  // tile buffers per wave of the tile-stream kernel (2: 1.60 ms, 3: 1.50 ms, 4: 1.71 ms -- LDS then holds 6 waves per CU)
#endif
#define CALL_P2G_WIDE_G(S, LWv, Gv)                                                                                                     \
  hipLaunchKernelGGL((p2g_wide_kernel<S, LWv, 1, Gv>), dim3(nbins / Gv), dim3(64 * Gv), 0, L.stream, mp, pd, t, grid, binStart, cellCount, \
                     nbr, stale, staleCount)
#define CALL_P2G_TILE_GM(S, Gv, Mv)                                                                                                     \
  hipLaunchKernelGGL((p2g_tile_kernel<S, ZS_P2GT_NB, Gv, Mv>), dim3(nbins / Gv), dim3(64 * Gv), P2GT_DYN_LDS, L.stream, mp, pd, t, grid, binStart,  \
                     cellCount, nbr, stale, staleCount)
#define CALL_P2G_TILE_G(S, Gv)                                                                                                          \
  do {                                                                                                                                  \
    if (merged) { CALL_P2G_TILE_GM(S, Gv, true); }                                                                                      \
    else { CALL_P2G_TILE_GM(S, Gv, false); }                                                                                            \
  } while (0)
#ifdef ZS_P2G_AB
#define CALL_P2G_GROUPS(CALLG, S, ...)                                                                                                  \
  do {                                                                                                                                  \
    if (S == 8 && group == 4) { CALLG(8, ##__VA_ARGS__, 4); }                                                                            \
    else if (S == 8 && group == 2) { CALLG(8, ##__VA_ARGS__, 2); }                                                                       \
    else { CALLG(S, ##__VA_ARGS__, 1); }                                                                                                \
  } while (0)
#else
#define CALL_P2G_GROUPS(CALLG, S, ...)                                                                                                  \
  do {                                                                                                                                  \
    if (S == 8) { CALLG(8, ##__VA_ARGS__, 2); }                                                                                          \
    else { CALLG(S, ##__VA_ARGS__, 1); }                                                                                                \
  } while (0)
#endif
#define CALL_P2G_WIDE(S, M, LWv)                                                                                                       \
  do {                                                                                                                                  \
    if (LWv == 64 && tileStream && aligned16) CALL_P2G_GROUPS(CALL_P2G_TILE_G, S);                                                      \
    else CALL_P2G_GROUPS(CALL_P2G_WIDE_G, S, LWv);                                                                                      \
    hipLaunchKernelGGL((p2g_stale_kernel<S, MPM_CACHED_STRESS>), dim3(STALE_BLOCKS), dim3(256), 0, L.stream, mp, pd, t, grid,            \
                       (const int *)stale, (const int *)staleCount);                                                                    \
  } while (0)
      (void)group;
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
