// tools/measure/p2g_rs.hpp -- measurement only (-DZS_ROCM_WITH_P2G_RS, run with ZS_ROCM_P2G_KERNEL=rs): a role-split stand-alone P2G.
// Result (64 Mi particles at rest, cached stress): correct (tests/test_mpm_gpu.py green), 2.85 ms against 1.84 ms of p2g_wide_kernel --
// four channel-set consumers spend ~1000 instructions per 64-particle round where the wide kernel spends 733, and a consumer wave's own
// chain (8 rounds x 250 instructions) is what a bin takes; decoupling the loads does not buy that back.
#pragma once
// ---------------------------------------------------------------------------------------------------------------------------
// Role-split stand-alone P2G (r03; cached stress).  p2g_wide_kernel keeps a cell's 27 x 7 accumulators in one wave: 256 VGPRs, two waves
// per SIMD, and its memory time (25 rows of every 64-particle tile: 1.07 ms per 64 Mi particles at the measured read ceiling) and its
// arithmetic (733 instructions per round: 0.94 ms) add up instead of overlapping (1.81 ms).  Here the loads and the arithmetic sit in
// different waves, as in g2p2g_rs_kernel: waves 0-3 LOAD round 4c + w of chunk c (one chunk ahead), derive the local position and stage
// {m, lpn, v, C, P F^T} in LDS; waves 4-7 are the channel-set consumers of the fused kernels (g2p2g_rs_consumer).  Particles that no
// longer sit under the lane of their cell go the way they go in the wide kernel: in-bin -> LDS queue + post-pass, outside -> stale list.
template <int SIDE, int LW, int W>
__device__ __forceinline__ void p2g_rs_loader(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, int start, unsigned cnt, int lane,
                                              int nchunks, float *stage, unsigned long long *smask, int *mq, int *mqCount, int *stale,
                                              int *staleCount) {
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = 1.0f / mp.dx;
  RoundWalk walk(cnt, start);
  auto next_chunk = [&](int &idx, bool &has) {
    has = false;
    idx = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int i;
      bool a;
      const bool h = walk.next(i, a);
      if (r == W) {
        idx = i;
        has = h;
      }
    }
  };
  struct Rec {
    float m, pos[3], vel[3], C[9], PF[9];
    __device__ __forceinline__ void load(const ParticlesDev &ps, size_t i) {
      const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
      m = pload1<LW>(ps.mass, o);
      pload<LW, 3>(ps.pos, o, pos);
      pload<LW, 3>(ps.vel, o, vel);
      pload<LW, 9>(ps.C, o, C);
      pload<LW, 9>(ps.stress, o, PF);
    }
  };
  Rec cur, nxt;
  int i0 = 0, i1 = 0;
  bool has0 = false, has1 = false;
  if (nchunks > 0) {
    next_chunk(i1, has1);
    if (has1) nxt.load(ps, (size_t)i1);
  }
  __syncthreads();  // (the consumers clear the arena meanwhile)
  for (int it = 0; it <= nchunks; ++it) {
    if (it < nchunks) {
      const int par = it & 1;
      float *myStage = stage + (size_t)(par * 4 + W) * (G2P2G_NF * 64);
      cur = nxt;
      has0 = has1;
      i0 = i1;
      has1 = false;
      if (it + 1 < nchunks) {
        next_chunk(i1, has1);
        if (has1) nxt.load(ps, (size_t)i1);  // in flight during this chunk
      }
      bool valid = false;
      if (has0) {
        float lpn[3];
        int nc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {  // base node / local position exactly as make_arena derives them
          const float X = cur.pos[d] * dxi;
          const float fl = floorf(X - 0.5f);
          nc[d] = (int)fl - geo.org[d];
          lpn[d] = X - fl;
        }
        // under the lane of its cell, and not one of the reference arena's rounding cases (see g2p2g_rs_producer): register accumulation
        const bool here = nc[0] == cx && nc[1] == cy && nc[2] == cz && lpn[0] >= 0.5f && lpn[0] < 1.5f && lpn[1] >= 0.5f && lpn[1] < 1.5f &&
                          lpn[2] >= 0.5f && lpn[2] < 1.5f;
        if (here) {
          valid = true;
          myStage[0 * 64 + lane] = cur.m;
#pragma unroll
          for (int d = 0; d < 3; ++d) myStage[(1 + d) * 64 + lane] = lpn[d];
#pragma unroll
          for (int d = 0; d < 3; ++d) myStage[(4 + d) * 64 + lane] = cur.vel[d];
#pragma unroll
          for (int d = 0; d < 9; ++d) myStage[(7 + d) * 64 + lane] = cur.C[d];
#pragma unroll
          for (int d = 0; d < 9; ++d) myStage[(16 + d) * 64 + lane] = cur.PF[d];
        } else {
          bool queued = false;
          if ((unsigned)nc[0] < 4u && (unsigned)nc[1] < 4u && (unsigned)nc[2] < 4u) {
            const int q = atomicAdd(mqCount, 1);
            if (q < G2P2G_MQ_CAP) {
              mq[q] = i0;
              queued = true;
            }
          }
          if (!queued) stale[atomicAdd(staleCount, 1)] = i0;
        }
      }
      {
        const unsigned long long vm = __ballot(valid);
        if (lane == 0) smask[par * 4 + W] = vm;
      }
    }
    __syncthreads();
  }
}
template <int SIDE, int LW>
static __global__ __launch_bounds__(512, 4) void p2g_rs_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *binStart,
                                                        const unsigned *cellCount, const int *nbr, int *stale, int *staleCount) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float parena[7 * AL::CH];
  __shared__ float stage[2 * 4 * G2P2G_NF * 64];
  __shared__ unsigned long long smask[2 * 4];
  __shared__ int mq[G2P2G_MQ_CAP];
  __shared__ int mqCount;
  if (threadIdx.x == 0) mqCount = 0;
  const int bin = blockIdx.x;
  const int start = binStart[bin], end = binStart[bin + 1];
  if (start == end) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  const unsigned cnt = cellCount[(size_t)bin * 64 + lane];
  unsigned mx = cnt;
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)mx, sft, 64);
    mx = o > mx ? o : mx;
  }
  const int nchunks = (int)((mx + 3u) >> 2);
  if (w == 0) p2g_rs_loader<SIDE, LW, 0>(mp, ps, geo, start, cnt, lane, nchunks, stage, smask, mq, &mqCount, stale, staleCount);
  else if (w == 1) p2g_rs_loader<SIDE, LW, 1>(mp, ps, geo, start, cnt, lane, nchunks, stage, smask, mq, &mqCount, stale, staleCount);
  else if (w == 2) p2g_rs_loader<SIDE, LW, 2>(mp, ps, geo, start, cnt, lane, nchunks, stage, smask, mq, &mqCount, stale, staleCount);
  else if (w == 3) p2g_rs_loader<SIDE, LW, 3>(mp, ps, geo, start, cnt, lane, nchunks, stage, smask, mq, &mqCount, stale, staleCount);
  else if (w == 4) g2p2g_rs_consumer<0>(mp, lane, nchunks, stage, smask, parena);
  else if (w == 5) g2p2g_rs_consumer<1>(mp, lane, nchunks, stage, smask, parena);
  else if (w == 6) g2p2g_rs_consumer<2>(mp, lane, nchunks, stage, smask, parena);
  else g2p2g_rs_consumer<3>(mp, lane, nchunks, stage, smask, parena);
  __syncthreads();  // all channel sets are in the arena
  {  // post-pass: the queued in-bin particles, one thread each, added to the arena with LDS atomics (values of the exact path)
    const int nm = mqCount < G2P2G_MQ_CAP ? mqCount : G2P2G_MQ_CAP;
    const float dxi = 1.0f / mp.dx;
    const float kscale = -mp.dt * (4.f * dxi * dxi);
    for (int q = tid; q < nm; q += 512) {
      const size_t i = (size_t)mq[q];
      float pos[3], vel[3], C[9], PF[9];
      load_attr<3>(ps.pos, i, pos);
      load_attr<3>(ps.vel, i, vel);
      load_attr<9>(ps.C, i, C);
      load_attr<9>(ps.stress, i, PF);
      const float m = ps.mass.base[ps.mass.off(i)];
#pragma unroll
      for (int d = 0; d < 9; ++d) PF[d] *= kscale;
      Arena ar;
      make_arena(mp.dx, pos, ar);
      const int kx = ar.corner[0] - geo.org[0], ky = ar.corner[1] - geo.org[1], kz = ar.corner[2] - geo.org[2];
      if ((unsigned)kx >= 4u || (unsigned)ky >= 4u || (unsigned)kz >= 4u) {
        stale[atomicAdd(staleCount, 1)] = (int)i;  // (cannot differ from the loader's test: the same arithmetic; kept as a guard)
        continue;
      }
      float *b0 = parena + AL::at(kx, ky, kz);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float W = ar.w[0][a] * ar.w[1][b] * ar.w[2][c];
            const float x0 = (float)a * mp.dx - ar.lp[0], x1 = (float)b * mp.dx - ar.lp[1], x2 = (float)c * mp.dx - ar.lp[2];
            float *g = b0 + AL::at(a, b, c);
            atomicAdd(g, W * m);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              atomicAdd(g + (1 + d) * AL::CH, W * m * (vel[d] + (C[d] * x0 + C[3 + d] * x1 + C[6 + d] * x2)));
              atomicAdd(g + (4 + d) * AL::CH, (PF[d] * x0 + PF[3 + d] * x1 + PF[6 + d] * x2) * W);
            }
          }
    }
    __syncthreads();
  }
  if (tid < 216) {
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = nbr[(size_t)geo.block * 8 + slot];
    if (bn >= 0) {
      const float *a = parena + AL::at(x, y, z);
      float *g = grid + (size_t)bn * 7 * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        const float v = a[ch * AL::CH];
        if (v != 0.f) unsafeAtomicAdd(g + ch * NC, v);
      }
    }
  }
}
