// tools/measure/p2g_half.hpp -- measurement only (-DZS_ROCM_WITH_P2G_HALF, run with ZS_ROCM_P2G_KERNEL=half): the stand-alone P2G (cached
// stress) with TWO waves per bin.  Each wave keeps all seven channels of half of a cell's 27 stencil nodes (14 / 13 nodes: 98 / 91
// accumulators, ~165 VGPRs, three waves per SIMD instead of the wide kernel's two); wave 0 streams the records into a three-slot LDS ring
// (global_load_lds, two rounds ahead), both waves accumulate from it; one workgroup barrier per round.
// Result (64 Mi particles at rest): correct (tests/test_mpm_gpu.py green), 164 VGPRs / 3 waves per SIMD, 2.07-2.10 ms against 1.86-1.89 ms of
// p2g_wide_kernel.
#pragma once
template <int NH>
__device__ __forceinline__ void p2gh_accumulate(const MpmDev &mp, const Arena &ar, const float *rec, float kscale, float (&acc)[14][7]) {
  const float m = rec[0];
  float xo[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int d = 0; d < 3; ++d) xo[d][k] = (float)k * mp.dx - ar.lp[d];
  float c1m[3], c2m[3], c1f[3], c2f[3], bm[3][3], bf[3][3];  // b*[a][d]
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float v = rec[(4 + d) * 64], c0 = rec[(7 + d) * 64], f0 = rec[(16 + d) * 64];
    c1m[d] = rec[(10 + d) * 64];
    c2m[d] = rec[(13 + d) * 64];
    c1f[d] = rec[(19 + d) * 64];
    c2f[d] = rec[(22 + d) * 64];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      bm[a][d] = fmaf(c0, xo[0][a], v);
      bf[a][d] = f0 * xo[0][a];
    }
  }
  // (what belongs to nodes of the other half is dead code after unrolling)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float wxy = ar.w[0][a] * ar.w[1][b];
      const float wxym = wxy * m, wxyk = wxy * kscale;
      float qm[3], qf[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        qm[d] = fmaf(c1m[d], xo[1][b], bm[a][d]);
        qf[d] = fmaf(c1f[d], xo[1][b], bf[a][d]);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int k = (a * 3 + b) * 3 + c;
        if ((k < 14) != (NH == 0)) continue;
        float(&A)[7] = acc[k - (NH == 0 ? 0 : 14)];
        const float Wm = wxym * ar.w[2][c], Wk = wxyk * ar.w[2][c];
        A[0] += Wm;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          A[1 + d] = fmaf(Wm, fmaf(c2m[d], xo[2][c], qm[d]), A[1 + d]);
          A[4 + d] = fmaf(Wk, fmaf(c2f[d], xo[2][c], qf[d]), A[4 + d]);
        }
      }
    }
}
template <int SIDE, int LW>
static __global__ __launch_bounds__(128, 3) void p2g_half_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *binStart,
                                                          const unsigned *cellCount, const int *nbr, int *stale, int *staleCount) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  constexpr int NB = 3;
  constexpr int LDSF = NB * P2GW_NF * 64 > 7 * AL::CH ? NB * P2GW_NF * 64 : 7 * AL::CH;
  __shared__ float lds[LDSF];  // record ring during the rounds, the bin's arena afterwards
  float *arena = lds;
  float(*pbuf)[P2GW_NF * 64] = reinterpret_cast<float(*)[P2GW_NF * 64]>(lds);
  __shared__ int mq[P2GW_MQ_CAP];
  __shared__ int mqCount;
  const int bin = blockIdx.x;
  const int start = binStart[bin], end = binStart[bin + 1];
  if (start == end) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) mqCount = 0;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const unsigned cnt = cellCount[(size_t)bin * 64 + lane];
  const float dxi = 1.0f / mp.dx;
  const float kscale = -mp.dt * (4.f * dxi * dxi);
  float acc[14][7];
#pragma unroll
  for (int k = 0; k < 14; ++k)
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) acc[k][ch] = 0.f;
  const size_t tileBase = p2gw_tile_base<LW>(ps, start);
  RoundWalk lead(cnt, start), walk(cnt, start);  // both waves walk the same counts; only wave 0 issues loads and queues strays
  int li;
  bool lany = true;
  int issued = 0;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (lany) {
      const bool lh = lead.next(li, lany);
      if (lany) {
        if (w == 0) p2gw_issue<LW>(ps, (size_t)li, lh, pbuf[d], tileBase);
        ++issued;
      }
    }
  }
  int slot = 0, lslot = 2;
  int i0;
  bool any;
  bool has0 = walk.next(i0, any);
  while (any) {
    // wave 0: the current round's record has landed (only the round issued after it may still be in flight)
    if (w == 0) {
      if (issued >= 2) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // the record is visible to both waves; both have finished reading the round before
    if (lany) {       // two rounds ahead, into the slot the round before occupied
      const bool lh = lead.next(li, lany);
      if (lany) {
        if (w == 0) p2gw_issue<LW>(ps, (size_t)li, lh, pbuf[lslot], tileBase);
        lslot = lslot + 1 == NB ? 0 : lslot + 1;
        ++issued;
      }
    }
    if (has0) {
      const float *rec = pbuf[slot] + lane;
      const float pos[3] = {rec[1 * 64], rec[2 * 64], rec[3 * 64]};
      Arena ar;
      make_arena(mp.dx, pos, ar);
      const int ocx = ar.corner[0] - geo.org[0], ocy = ar.corner[1] - geo.org[1], ocz = ar.corner[2] - geo.org[2];
      if (ocx == cx && ocy == cy && ocz == cz) {
        if (w == 0) p2gh_accumulate<0>(mp, ar, rec, kscale, acc);
        else p2gh_accumulate<1>(mp, ar, rec, kscale, acc);
      } else if (w == 0) {
        bool queued = false;
        if ((unsigned)ocx < 4u && (unsigned)ocy < 4u && (unsigned)ocz < 4u) {
          const int q = atomicAdd(&mqCount, 1);
          if (q < P2GW_MQ_CAP) {
            mq[q] = i0;
            queued = true;
          }
        }
        if (!queued) stale[atomicAdd(staleCount, 1)] = i0;
      }
    }
    --issued;
    slot = slot + 1 == NB ? 0 : slot + 1;
    has0 = walk.next(i0, any);
  }
  __syncthreads();  // every record has been consumed: the region becomes the arena
  for (int k = tid; k < 7 * AL::CH; k += 128) arena[k] = 0.f;
  __syncthreads();
  float *a0 = arena + AL::at(cx, cy, cz);
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {  // the halves share nodes across cells: one wave at a time, phases ordered inside the wave
    if (s == w) {
#pragma unroll
      for (int kk = 0; kk < 14; ++kk) {
        const int k = kk + (s == 0 ? 0 : 14);
        if (k < 27) {
          float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
          for (int ch = 0; ch < 7; ++ch) g[ch * AL::CH] += acc[kk][ch];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
  }
  {  // post-pass: the queued in-bin particles, one thread each, LDS atomics (same values as the exact path)
    const int nm = mqCount < P2GW_MQ_CAP ? mqCount : P2GW_MQ_CAP;
    for (int q = tid; q < nm; q += 128) {
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
      float *b0 = arena + AL::at(ar.corner[0] - geo.org[0], ar.corner[1] - geo.org[1], ar.corner[2] - geo.org[2]);
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
  for (int node = tid; node < 216; node += 128) {
    const int x = node / 36, y = (node / 6) % 6, z = node % 6;
    int slot2, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot2, cell);
    const int bn = nbr[(size_t)geo.block * 8 + slot2];
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
