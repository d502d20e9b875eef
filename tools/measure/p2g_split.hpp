// tools/measure/p2g_split.hpp -- measurement only (-DZS_ROCM_WITH_P2G_SPLIT, run with ZS_ROCM_P2G_SPLIT=1): the r01 four-wave channel-split
// stand-alone P2G (1113 VALU instructions per 64-particle round; replaced by p2g_wide_kernel in r01, kept for comparison since).
#pragma once
// ---- binned path, cached stress: channel-split workgroup.
// With the constitutive update moved to the tail of G2P, P2G has ~15 VALU ops per byte-lane left and becomes latency bound
// at two waves per SIMD (the 27 x 4 register stencil of sweep A costs 175 VGPRs).  Here one workgroup of FOUR waves owns a
// bin and each wave accumulates a subset of the 7 grid channels (all waves walk the same particles; the repeated reads of
// x / m hit L1/L2):   wave 0: m, mv_x    wave 1: mv_y, mv_z    wave 2: rhs_x, rhs_y    wave 3: rhs_z
// -> <= 54 accumulators per lane, ~4x the loads in flight per bin, and the per-bin zero/flush work spread over 256 lanes.
template <int ROLE> struct SplitRole {
  static constexpr int NA = (ROLE == 0 || ROLE == 3) ? 1 : 2;  // affine channels handled by this wave
  static constexpr bool HASMASS = ROLE == 0;                   // plus the mass channel
  static constexpr bool USEM = ROLE < 2;                       // momentum channels are weighted by W * m, stress ones by W
  static constexpr int CH0 = ROLE == 0 ? 1 : (ROLE == 1 ? 2 : (ROLE == 2 ? 4 : 6));  // first arena channel
};
template <int ROLE, int LW> struct SplitRec {
  using R = SplitRole<ROLE>;
  float pos[3], m, b[R::NA], g[R::NA][3];
  __device__ __forceinline__ void load(const ParticlesDev &ps, size_t i) {
    const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
    pload<LW, 3>(ps.pos, o, pos);
    if constexpr (R::USEM) {
      m = pload1<LW>(ps.mass, o);
      constexpr int d0 = ROLE == 0 ? 0 : 1;
#pragma unroll
      for (int k = 0; k < R::NA; ++k) {
        b[k] = pload1<LW>(ps.vel, o, d0 + k);
#pragma unroll
        for (int j = 0; j < 3; ++j) g[k][j] = pload1<LW>(ps.C, o, d0 + k + 3 * j);  // row d of the column-major C
      }
    } else {
      constexpr int d0 = ROLE == 2 ? 0 : 2;
#pragma unroll
      for (int k = 0; k < R::NA; ++k) {
        b[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) g[k][j] = pload1<LW>(ps.stress, o, d0 + k + 3 * j);
      }
    }
  }
};

// one particle's contribution of the channels of ROLE to the 27 stencil nodes of its cell (registers):
//   momentum roles:  W m (b + g . (xi - xp))      stress roles:  W kscale (g . (xi - xp))        [+ W m for the mass channel]
// evaluated as Ws * ((Px[a] + Py[b]) + Pz[c]) with the per-axis products hoisted
template <int ROLE>
__device__ __forceinline__ void split_accumulate(const MpmDev &mp, const Arena &ar, float m, float kscale,
                                                 const float (&b)[SplitRole<ROLE>::NA], const float (&g)[SplitRole<ROLE>::NA][3],
                                                 float (&accm)[27], float (&acc)[27][SplitRole<ROLE>::NA]) {
  using R = SplitRole<ROLE>;
  float Px[3][R::NA], Py[3][R::NA], Pz[3][R::NA], wzs[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float x0 = (float)k * mp.dx - ar.lp[0], x1 = (float)k * mp.dx - ar.lp[1], x2 = (float)k * mp.dx - ar.lp[2];
#pragma unroll
    for (int q = 0; q < R::NA; ++q) {
      Px[k][q] = g[q][0] * x0;
      Py[k][q] = g[q][1] * x1;
      Pz[k][q] = fmaf(g[q][2], x2, b[q]);
    }
    wzs[k] = ar.w[2][k] * (R::USEM ? m : kscale);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float wxy = ar.w[0][a] * ar.w[1][bb];
      float qv[R::NA];
#pragma unroll
      for (int q = 0; q < R::NA; ++q) qv[q] = Px[a][q] + Py[bb][q];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float Ws = wxy * wzs[c];
        const int n = (a * 3 + bb) * 3 + c;
        if constexpr (R::HASMASS) accm[n] += Ws;
#pragma unroll
        for (int q = 0; q < R::NA; ++q) acc[n][q] = fmaf(Ws, qv[q] + Pz[c][q], acc[n][q]);
      }
    }
}

template <int SIDE, int ROLE, int LW>
__device__ __forceinline__ void p2g_split_sweep(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, int start, unsigned cnt,
                                                int cx, int cy, int cz, float *a0, int *stale, int *staleCount) {
  using AL = ArenaLds;
  using R = SplitRole<ROLE>;
  const float dxi = 1.0f / mp.dx;
  const float kscale = R::USEM ? 1.f : -mp.dt * (4.f * dxi * dxi);  // contrib = -dt D_inv (P F^T vol)
  float accm[27];
  float acc[27][R::NA];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    accm[k] = 0.f;
#pragma unroll
    for (int q = 0; q < R::NA; ++q) acc[k][q] = 0.f;
  }
  RoundWalk walk(cnt, start);
  int i0, i1;
  bool any, any1;
  bool has0 = walk.next(i0, any);
  SplitRec<ROLE, LW> cur, nxt;
  if (has0) cur.load(ps, (size_t)i0);
  while (any) {
    const bool has1 = walk.next(i1, any1);
    if (has1) nxt.load(ps, (size_t)i1);
    if (has0) {
      Arena ar;
      make_arena(mp.dx, cur.pos, ar);
      if (ar.corner[0] - geo.org[0] != cx || ar.corner[1] - geo.org[1] != cy || ar.corner[2] - geo.org[2] != cz) {
        if constexpr (ROLE == 0) stale[atomicAdd(staleCount, 1)] = i0;  // exact path afterwards (queued once)
      } else {
        split_accumulate<ROLE>(mp, ar, cur.m, kscale, cur.b, cur.g, accm, acc);
      }
    }
    cur = nxt;
    has0 = has1;
    i0 = i1;
    any = any1;
  }
  // 27 conflict-free phases; every wave of the workgroup executes the same 27 barriers
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
    if constexpr (R::HASMASS) g[0] += accm[k];
#pragma unroll
    for (int q = 0; q < R::NA; ++q) g[(R::CH0 + q) * AL::CH] += acc[k][q];
    __syncthreads();
  }
}

template <int SIDE, int LW>
static __global__ __launch_bounds__(256) void p2g_binned_split_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *binStart,
                                                               const unsigned *cellCount, const int *nbr, int *stale, int *staleCount) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float arena[7 * AL::CH];
  const int bin = blockIdx.x;
  const int start = binStart[bin], end = binStart[bin + 1];
  if (start == end) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int k = tid; k < 7 * AL::CH; k += 256) arena[k] = 0.f;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const unsigned cnt = cellCount[(size_t)bin * 64 + lane];
  float *a0 = arena + AL::at(cx, cy, cz);
  __syncthreads();
  if (w == 0) p2g_split_sweep<SIDE, 0, LW>(mp, ps, geo, start, cnt, cx, cy, cz, a0, stale, staleCount);
  else if (w == 1) p2g_split_sweep<SIDE, 1, LW>(mp, ps, geo, start, cnt, cx, cy, cz, a0, stale, staleCount);
  else if (w == 2) p2g_split_sweep<SIDE, 2, LW>(mp, ps, geo, start, cnt, cx, cy, cz, a0, stale, staleCount);
  else p2g_split_sweep<SIDE, 3, LW>(mp, ps, geo, start, cnt, cx, cy, cz, a0, stale, staleCount);
  // flush (the last phase barrier has made every channel visible): thread = arena node, decoded once for all 7 channels
  if (tid < 216) {
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = nbr[(size_t)geo.block * 8 + slot];
    if (bn >= 0) {
      const float *a = arena + AL::at(x, y, z);
      float *g = grid + (size_t)bn * 7 * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        const float v = a[ch * AL::CH];
        if (v != 0.f) unsafeAtomicAdd(g + ch * NC, v);
      }
    }
  }
}

