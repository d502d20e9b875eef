// tools/measure/slot_consumer_nc.hpp -- measurement only (-DZS_SLOT_WITH_NC, run with ZS_ROCM_SLOT_CONSUMERS=halves): the four consumer
// waves as 2 node halves x 2 channel groups.  Wave (NH, CG) takes nodes k < 14 (NH = 0) or k >= 14 (NH = 1) of the 27 (k = (a 3 + b) 3 + c)
// with the channels {m, mv_x, mv_y, mv_z} (CG = 0) or {f_x, f_y, f_z} (CG = 1): 56 / 42 accumulators, and per round about 200 instead of
// 250 instructions PER WAVE -- the chain between two chunk barriers that the node split with three waves left unchanged.
// Result (64 Mi moving column): correct, 8.38 ms/step against 8.09 with the four channel-set consumers; 5.76 against 5.59 at rest.
#pragma once
template <int NH, int CG>
__device__ __forceinline__ void g2p2g_consume_half(const MpmDev &mp, const float *st, int spos, float kscale, float (&acc)[14][CG == 0 ? 4 : 3]) {
  auto f = [&](int k) { return st[k * 64 + spos]; };
  float w[3][3], xo[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float d0 = f(1 + d);
    w[d][0] = 0.5f * (1.5f - d0) * (1.5f - d0);
    const float d1 = d0 - 1.0f;
    w[d][1] = 0.75f - d1 * d1;
    const float zz = 0.5f + d1;
    w[d][2] = 0.5f * zz * zz;
    const float lp = d0 * mp.dx;
#pragma unroll
    for (int k = 0; k < 3; ++k) xo[d][k] = (float)k * mp.dx - lp;
  }
  constexpr int cb = CG == 0 ? 7 : 16;
  const float scale = CG == 0 ? f(0) : kscale;
  float c0[3], c1[3], c2[3], v[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    c0[d] = f(cb + d);
    c1[d] = f(cb + 3 + d);
    c2[d] = f(cb + 6 + d);
    v[d] = CG == 0 ? f(4 + d) : 0.f;
  }
  // (everything that belongs to nodes of the other half is dead code after unrolling)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float wxs = w[0][a] * scale;
    float base[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) base[d] = fmaf(c0[d], xo[0][a], v[d]);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float wxy = wxs * w[1][b];
      float q[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) q[d] = fmaf(c1[d], xo[1][b], base[d]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        constexpr int dummy = 0;
        const int k = (a * 3 + b) * 3 + c;
        if ((k < 14) != (NH == 0)) continue;
        float(&A)[CG == 0 ? 4 : 3] = acc[k - (NH == 0 ? 0 : 14) + dummy];
        const float W = wxy * w[2][c];
        if (CG == 0) A[0] += W;
#pragma unroll
        for (int d = 0; d < 3; ++d) A[(CG == 0 ? 1 : 0) + d] = fmaf(W, fmaf(c2[d], xo[2][c], q[d]), A[(CG == 0 ? 1 : 0) + d]);
      }
    }
  }
}
template <int SIDE, int NH, int CG>
__device__ __forceinline__ void g2p2g_slot_consumer_nc(const MpmDev &mp, const BinGeom<SIDE> &geo, unsigned mask, int total, int lane, int nchunks,
                                                       const SlotShared &sh, const SlotArgs &A) {
  using AL = ArenaLds;
  constexpr int NCH = CG == 0 ? 4 : 3, CH0 = CG == 0 ? 0 : 4, NN = NH == 0 ? 14 : 13;
  constexpr int NC = SIDE * SIDE * SIDE;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = 1.0f / mp.dx;
  const float kscale = -mp.dt * (4.f * dxi * dxi);
  const unsigned long long lt = lanemask_lt();
  const float *const stage = sh.stage;
  const unsigned long long *const smask = sh.smask;
  unsigned(*const arrCnt)[64] = sh.arrCnt;
  const unsigned short(*const arrQ)[64][SL_ARRQ] = sh.arrQ;
  float *const parena = sh.parena;
  unsigned *const xCnt = sh.xCnt;
  const unsigned(*const xq)[SL_XQ] = sh.xq;
  const int *const nbrBlk = sh.nbrBlk;
  float acc[14][NCH];
#pragma unroll
  for (int k = 0; k < 14; ++k)
#pragma unroll
    for (int q = 0; q < NCH; ++q) acc[k][q] = 0.f;
  for (int k = (int)threadIdx.x - 256; k < 7 * AL::CH; k += 256) parena[k] = 0.f;  // the four consumer waves clear the bin's arena
  __syncthreads();  // (the producers fill the velocity arena meanwhile)
  int r = 0, off = 0;
  for (int it = 0; it <= nchunks; ++it) {
    if (it > 0) {
      const int par = (it - 1) % 3;
      const int produced = 256 * it < total ? 256 * it : total;
      const unsigned qn = arrCnt[par][lane];
      const int na = qn < (unsigned)SL_ARRQ ? (int)qn : SL_ARRQ;
      int ai = 0;
      if (NH == 0 && CG == 0) arrCnt[(it + 1) % 3][lane] = 0u;
#pragma unroll 1
      for (;;) {
        bool roundOk = false, has = false;
        unsigned long long occ = 0ull;
        int cnt = 0;
        if (off < total) {
          has = (mask >> r) & 1u;
          occ = __ballot(has);
          cnt = __popcll(occ);
          roundOk = off + cnt <= produced;
        }
        const bool pend = ai < na;
        if (!roundOk && __ballot(pend) == 0ull) break;
        int spos = -1;
        if (roundOk) {
          if (has) {
            const int e = off + __popcll(occ & lt);
            const int grp = (e >> 6) % SL_NG, pos = e & 63;
            if ((smask[grp] >> pos) & 1ull) spos = grp * (G2P2G_NF * 64) + pos;
          }
          off += cnt;
          ++r;
        }
        if (spos < 0 && pend) {
          const unsigned p = arrQ[par][lane][ai++];
          spos = (int)(p >> 6) * (G2P2G_NF * 64) + (int)(p & 63u);
        }
        if (spos >= 0) g2p2g_consume_half<NH, CG>(mp, stage, spos, kscale, acc);
      }
      // the chunk's global-atomic list: this wave's channel group, every second pair of entries (the other node half takes the rest)
      const int nx = xCnt[par] < (unsigned)SL_XQ ? (int)xCnt[par] : SL_XQ;
      if (NH == 0 && CG == 0 && lane == 0) xCnt[(it + 1) % 3] = 0u;
      {
        const int node = lane & 31, half = lane >> 5;
        const int sel[3] = {node / 9, (node / 3) % 3, node % 3};
        float ws[3], wt[3], wa[3], wb[3], xo[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          ws[q] = sel[q] == 0 ? -1.f : 1.f;
          wt[q] = sel[q] == 0 ? 1.5f : (sel[q] == 1 ? -1.f : -0.5f);
          wa[q] = sel[q] == 1 ? 0.75f : 0.f;
          wb[q] = sel[q] == 1 ? -1.f : 0.5f;
          xo[q] = (float)sel[q] * mp.dx;
        }
        if (node < 27 && nx > 0) {
#pragma unroll 1
          for (int k = half + 2 * NH; k < nx; k += 4) {
            const unsigned e = xq[par][k];
            const float *st = stage + (size_t)((e & 1023u) >> 6) * (G2P2G_NF * 64) + (e & 63u);
            float Wt = 1.f, xi[3];
            int g[3], code = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const float d0 = st[(1 + q) * 64];
              const float u = fmaf(ws[q], d0 - floorf(d0 - 0.5f), wt[q]);
              Wt *= fmaf(wb[q], u * u, wa[q]);
              xi[q] = fmaf(-mp.dx, d0, xo[q]);
              g[q] = (int)((e >> (10 + 3 * q)) & 7u) - 1 + geo.o[q] + sel[q];
              code = code * 3 + 1 + (g[q] >= SIDE ? 1 : 0) - (g[q] < 0 ? 1 : 0);
            }
            const int bn = nbrBlk[code];
            if (bn >= 0) {
              const int cell = ((g[0] & (SIDE - 1)) * SIDE + (g[1] & (SIDE - 1))) * SIDE + (g[2] & (SIDE - 1));
              float *gp = A.gridB + ((size_t)bn * 7 + CH0) * NC + cell;
              const float Wm = Wt * (CG == 0 ? st[0] : kscale);
              if (CG == 0 && Wm != 0.f) unsafeAtomicAdd(gp, Wm);
#pragma unroll
              for (int d = 0; d < 3; ++d) {
                const int iC = (CG == 0 ? 7 : 16) + d;
                float t = st[iC * 64] * xi[0];
                t = fmaf(st[(iC + 3) * 64], xi[1], t);
                t = fmaf(st[(iC + 6) * 64], xi[2], t);
                if (CG == 0) t += st[(4 + d) * 64];
                const float val = Wm * t;
                if (val != 0.f) unsafeAtomicAdd(gp + ((CG == 0 ? 1 : 0) + d) * NC, val);
              }
            } else if (CG == 0) {
              A.status[2] = 1;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // the two node halves of a channel group meet in the arena: first half, barrier, second half; phases ordered inside a wave
  float *a0 = parena + (size_t)CH0 * AL::CH + AL::at(cx, cy, cz);
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    if (s == NH) {
#pragma unroll
      for (int kk = 0; kk < NN; ++kk) {
        const int k = kk + (NH == 0 ? 0 : 14);
        float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
        for (int q = 0; q < NCH; ++q) g[q * AL::CH] += acc[kk][q];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    __syncthreads();
  }
}
