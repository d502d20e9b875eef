// tools/measure/slot_consumer_ns.hpp -- measurement only (included by zpc_amd/csrc/mpm_slotted.hip under -DZS_SLOT_WITH_NS; select at run
// time with ZS_ROCM_SLOT_CONSUMERS=nodes).  Result on the 64 Mi-particle moving column (profiles/r03_slot_probe.md): 11 % fewer VALU
// instructions per launch (3.25e9 instead of 3.67e9), the same step time (8.13 vs 8.12-8.18 ms), 6 % slower at rest: the step is paced by
// the producer waves' own instruction chains, not by the sum of the instructions.
#pragma once
// ---------------------------------------------------------------------------------------------------------------------------
// Node-split consumers (r03).  The four channel-set consumers above repeat the weight arithmetic of a staged particle four times
// (3 x 8 axis weights, 9 + 27 products): a third of their instructions.  Here three consumer waves share the 27 nodes instead -- wave AX
// takes the nine nodes of x-offset AX with ALL seven channels (63 accumulators) -- and the fourth wave of the group takes over the
// global-atomic list of the cross-bin movers.  The state machine over rounds and arrivals is the one of g2p2g_slot_consumer.
template <int AX>
__device__ __forceinline__ void g2p2g_consume_nodes(const MpmDev &mp, const float *st, int spos, float kscale, float (&acc)[9][7]) {
  auto f = [&](int k) { return st[k * 64 + spos]; };
  const float d0x = f(1), d0y = f(2), d0z = f(3);
  float wx;
  if (AX == 0) {
    wx = 0.5f * (1.5f - d0x) * (1.5f - d0x);
  } else if (AX == 1) {
    const float d1 = d0x - 1.0f;
    wx = 0.75f - d1 * d1;
  } else {
    const float zz = 0.5f + (d0x - 1.0f);
    wx = 0.5f * zz * zz;
  }
  float wy[3], wz[3], x1[3], x2[3];
  {
    wy[0] = 0.5f * (1.5f - d0y) * (1.5f - d0y);
    const float d1 = d0y - 1.0f;
    wy[1] = 0.75f - d1 * d1;
    const float zz = 0.5f + d1;
    wy[2] = 0.5f * zz * zz;
  }
  {
    wz[0] = 0.5f * (1.5f - d0z) * (1.5f - d0z);
    const float d1 = d0z - 1.0f;
    wz[1] = 0.75f - d1 * d1;
    const float zz = 0.5f + d1;
    wz[2] = 0.5f * zz * zz;
  }
  const float x0 = (float)AX * mp.dx - d0x * mp.dx;
  const float lpy = d0y * mp.dx, lpz = d0z * mp.dx;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    x1[k] = (float)k * mp.dx - lpy;
    x2[k] = (float)k * mp.dx - lpz;
  }
  {  // mass + momentum: W m (v + C (xi - xp))
    const float wxm = wx * f(0);
    float q[3][3], c2[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float base = fmaf(f(7 + d), x0, f(4 + d));
      const float c1 = f(10 + d);
      c2[d] = f(13 + d);
#pragma unroll
      for (int b = 0; b < 3; ++b) q[d][b] = fmaf(c1, x1[b], base);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float wxy = wxm * wy[b];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float W = wxy * wz[c];
        float(&A)[7] = acc[b * 3 + c];
        A[0] += W;
#pragma unroll
        for (int d = 0; d < 3; ++d) A[1 + d] = fmaf(W, fmaf(c2[d], x2[c], q[d][b]), A[1 + d]);
      }
    }
  }
  {  // force: -dt Dinv W (P F^T (xi - xp))
    const float wxk = wx * kscale;
    float q[3][3], c2[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float base = f(16 + d) * x0;
      const float c1 = f(19 + d);
      c2[d] = f(22 + d);
#pragma unroll
      for (int b = 0; b < 3; ++b) q[d][b] = fmaf(c1, x1[b], base);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float wxy = wxk * wy[b];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float W = wxy * wz[c];
        float(&A)[7] = acc[b * 3 + c];
#pragma unroll
        for (int d = 0; d < 3; ++d) A[4 + d] = fmaf(W, fmaf(c2[d], x2[c], q[d][b]), A[4 + d]);
      }
    }
  }
}
template <int SIDE, int AX>
__device__ __forceinline__ void g2p2g_slot_consumer_ns(const MpmDev &mp, const BinGeom<SIDE> &geo, unsigned mask, int total, int lane, int nchunks,
                                                       const SlotShared &sh, const SlotArgs &A) {
  using AL = ArenaLds;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = 1.0f / mp.dx;
  const float kscale = -mp.dt * (4.f * dxi * dxi);
  const unsigned long long lt = lanemask_lt();
  const float *const stage = sh.stage;
  const unsigned long long *const smask = sh.smask;
  unsigned(*const arrCnt)[64] = sh.arrCnt;
  const unsigned short(*const arrQ)[64][SL_ARRQ] = sh.arrQ;
  float *const parena = sh.parena;
  float acc[9][7];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[k][q] = 0.f;
  for (int k = (int)threadIdx.x - 256; k < 7 * AL::CH; k += 256) parena[k] = 0.f;  // waves 4-7 clear the bin's arena
  __syncthreads();  // (the producers fill the velocity arena meanwhile)
  int r = 0, off = 0;  // next round to consume, entry number of its first particle
  for (int it = 0; it <= nchunks; ++it) {
    if (it > 0) {
      const int par = (it - 1) % 3;
      const int produced = 256 * it < total ? 256 * it : total;
      const unsigned qn = arrCnt[par][lane];
      const int na = qn < (unsigned)SL_ARRQ ? (int)qn : SL_ARRQ;
      int ai = 0;
      if (AX == 0) arrCnt[(it + 1) % 3][lane] = 0u;  // the counters the NEXT chunk will use (last read one iteration ago)
#pragma unroll 1
      for (;;) {
        bool roundOk = false, has = false;
        unsigned long long occ = 0ull;
        int cnt = 0;
        if (off < total) {
          has = (mask >> r) & 1u;
          occ = __ballot(has);
          cnt = __popcll(occ);
          roundOk = off + cnt <= produced;  // else: the round's last entries belong to the chunk in production
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
        if (spos < 0 && pend) {  // a lane without a particle of its own in this round takes an arrival
          const unsigned p = arrQ[par][lane][ai++];
          spos = (int)(p >> 6) * (G2P2G_NF * 64) + (int)(p & 63u);
        }
        if (spos >= 0) g2p2g_consume_nodes<AX>(mp, stage, spos, kscale, acc);
      }
    }
    __syncthreads();
  }
  // the three waves' nodes overlap in the arena (cell cx + offset AX): one wave at a time, phases ordered inside the wave
  float *a0 = parena + AL::at(cx + AX, cy, cz);
#pragma unroll 1
  for (int s = 0; s < 3; ++s) {
    if (s == AX) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        float *g = a0 + AL::at(0, k / 3, k % 3);
#pragma unroll
        for (int q = 0; q < 7; ++q) g[q * AL::CH] += acc[k][q];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    __syncthreads();
  }
}
// the fourth wave of the consumer group: the chunk's list of movers whose new cell is not a lane of this bin (or whose cell's arrival queue
// was full, or whose local position left [0.5, 1.5) by rounding): all seven channels of their 27 node terms straight to the grid, two
// entries per pass (lane = entry parity x stencil node)
template <int SIDE>
__device__ __forceinline__ void g2p2g_slot_lister(const MpmDev &mp, const BinGeom<SIDE> &geo, int lane, int nchunks, const SlotShared &sh,
                                                  const SlotArgs &A) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  const float dxi = 1.0f / mp.dx;
  const float kscale = -mp.dt * (4.f * dxi * dxi);
  const float *const stage = sh.stage;
  unsigned *const xCnt = sh.xCnt;
  const unsigned(*const xq)[SL_XQ] = sh.xq;
  const int *const nbrBlk = sh.nbrBlk;
  float *const parena = sh.parena;
  for (int k = (int)threadIdx.x - 256; k < 7 * AL::CH; k += 256) parena[k] = 0.f;
  __syncthreads();
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
  for (int it = 0; it <= nchunks; ++it) {
    if (it > 0) {
      const int par = (it - 1) % 3;
      const int nx = xCnt[par] < (unsigned)SL_XQ ? (int)xCnt[par] : SL_XQ;
      if (lane == 0) xCnt[(it + 1) % 3] = 0u;
      if (node < 27 && nx > 0) {
#pragma unroll 1
        for (int k = half; k < nx; k += 2) {
          const unsigned e = xq[par][k];
          const float *st = stage + (size_t)((e & 1023u) >> 6) * (G2P2G_NF * 64) + (e & 63u);
          float Wt = 1.f, xi[3];
          int g[3], code = 0;
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const float d0 = st[(1 + q) * 64];
            const float u = fmaf(ws[q], d0 - floorf(d0 - 0.5f), wt[q]);  // the reference's second base_node (see `edge` in the producer)
            Wt *= fmaf(wb[q], u * u, wa[q]);
            xi[q] = fmaf(-mp.dx, d0, xo[q]);
            g[q] = (int)((e >> (10 + 3 * q)) & 7u) - 1 + geo.o[q] + sel[q];
            code = code * 3 + 1 + (g[q] >= SIDE ? 1 : 0) - (g[q] < 0 ? 1 : 0);
          }
          const int bn = nbrBlk[code];
          if (bn >= 0) {
            const int cell = ((g[0] & (SIDE - 1)) * SIDE + (g[1] & (SIDE - 1))) * SIDE + (g[2] & (SIDE - 1));
            float *gp = A.gridB + (size_t)bn * 7 * NC + cell;
            const float Wm = Wt * st[0], Wk = Wt * kscale;
            if (Wm != 0.f) unsafeAtomicAdd(gp, Wm);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              float t = st[(7 + d) * 64] * xi[0];
              t = fmaf(st[(10 + d) * 64], xi[1], t);
              t = fmaf(st[(13 + d) * 64], xi[2], t);
              t += st[(4 + d) * 64];  // v_d + (C . xi), the association of P2G.hpp:112
              const float val = Wm * t;
              if (val != 0.f) unsafeAtomicAdd(gp + (1 + d) * NC, val);
              float u2 = st[(16 + d) * 64] * xi[0];
              u2 = fmaf(st[(19 + d) * 64], xi[1], u2);
              u2 = fmaf(st[(22 + d) * 64], xi[2], u2);
              const float fv = Wk * u2;
              if (fv != 0.f) unsafeAtomicAdd(gp + (4 + d) * NC, fv);
            }
          } else {
            A.status[2] = 1;  // mass for a node whose block is not in the partition
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll 1
  for (int s = 0; s < 3; ++s) __syncthreads();  // the consumers' three flush stages
}

