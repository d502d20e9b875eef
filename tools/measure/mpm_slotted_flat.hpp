// tools/measure/mpm_slotted_flat.hpp -- measurement only (-DZS_SLOT_WITH_FLAT, run with ZS_ROCM_SLOT_SCHEDULE=flat): the slotted fused step
// with a FLAT schedule.  Correct (tests/test_mpm_gpu.py slotted cases green) and SLOWER: 8.79-8.84 ms/step against 8.05-8.07 on the moving
// 64 Mi column, 5.89 against 5.61 at rest (17 VGPR spills; the two waves of a channel set flush one after the other).
//
// g2p2g_slot_kernel pipelines a bin's chunks through four producer and four consumer waves: the consumers work on chunk c - 1 while the
// producers work on chunk c.  A bin of the 64 Mi column has 7.5 groups of 64 entries = 1.9 chunks, so the pipeline is two stages deep and a
// third of a workgroup's life is fill (consumers idle during chunk 0) and drain (producers idle during the last chunk) --
// profiles/r03_slot_probe.md: the consumers wait 39 k of 94 k cycles, the producers 10 k at the end.  Here every wave does both jobs, one
// after the other: all EIGHT waves produce (a chunk = 8 groups = 512 entries: the whole bin, usually), barrier, all eight consume -- the two
// waves that share a channel set take the loop iterations of the round / arrival state machine in turns and the entries of the
// global-atomic list alternately -- then flush one after the other into the bin's arena.  Heterogeneous work still overlaps on a CU: its
// two workgroups are in different phases.  The per-particle and per-round code is that of g2p2g_slot_producer / g2p2g_slot_consumer.
#pragma once

template <int SIDE, int SMODEL, bool WRITE_ALL, int W>
__device__ __forceinline__ void g2p2g_slot_flat(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, int bin, unsigned mask,
                                                int total, int lane, const SlotShared &sh, const SlotArgs &A) {
  constexpr int CS = W & 3, HALF = W >> 2;
  using S = ConsumerSet<CS>;
  using AL = ArenaLds;
  constexpr int LW = 64;
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool FLUID = model_is_fluid(SMODEL);
  constexpr int NC = SIDE * SIDE * SIDE;
  const float dxi = 1.0f / mp.dx;
  const float D_inv = 4.f * dxi * dxi;
  const size_t rowBase = (size_t)bin * (size_t)A.K;
  const unsigned kmask = A.K >= 32 ? 0xffffffffu : ((1u << A.K) - 1u);
  float *const varena = sh.varena, *const stage = sh.stage;
  unsigned long long *const smask = sh.smask;
  const unsigned short *const tab = sh.tab;
  unsigned *const mask0 = sh.mask0, *const clr = sh.clr, *const arrLocal = sh.arrLocal;
  const int *const nbrBin = sh.nbrBin;
  unsigned(*const arrCnt)[64] = sh.arrCnt;
  unsigned short(*const arrQ)[64][SL_ARRQ] = sh.arrQ;
  unsigned *const xCnt = sh.xCnt;
  unsigned(*const xq)[SL_XQ] = sh.xq;
  int *const outCount = sh.outCount, *const sent = sh.sent, *const homed = sh.homed, *const xOver = sh.xOver;
  const unsigned long long *const smaskc = sh.smask;
  const float kscale = -mp.dt * (4.f * dxi * dxi);
  float *const parena = sh.parena;
  const int *const nbrBlk = sh.nbrBlk;
  const int nchunks = (total + 511) >> 9;
  RecG<LW, DP, FLUID> cur;
  bool has0 = false;
  size_t i0 = 0;
  unsigned code0 = 0;
  {
    const int j = 64 * W + lane;
    has0 = j < total;
    if (has0) {
      code0 = tab[j];
      i0 = (rowBase + (size_t)(code0 >> 6)) * 64 + (size_t)(code0 & 63u);
      cur.load(ps, i0);
    }
  }
  {
    const int tid = (int)threadIdx.x;
    if (tid < 216) {
      const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
      int slot, cell;
      arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
      const int bn = A.nbr[(size_t)geo.block * 8 + slot];
      float *a = varena + AL::at(x, y, z);
      const float *g = A.gridA + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) a[ch * AL::CH] = bn >= 0 ? g[ch * NC] : 0.f;
    }
    for (int k = tid; k < 7 * AL::CH; k += 512) parena[k] = 0.f;
  }
  __syncthreads();
  int r = 0, off = 0;  // consumer state: next round to consume, entry number of its first particle
  for (int it = 0; it < nchunks; ++it) {
    {  // ---------------------------------------------------------------- produce group 8 it + W
      const int grp = 8 * it + W;
      const int par = it % 3;
      float *myStage = stage + (size_t)(grp % SL_NG) * (G2P2G_NF * 64);
      if (it > 0) {  // (the first chunk's record was requested in front of the arena fill)
        const int j = 64 * grp + lane;
        has0 = j < total;
        if (has0) {
          code0 = tab[j];
          i0 = (rowBase + (size_t)(code0 >> 6)) * 64 + (size_t)(code0 & 63u);
          cur.load(ps, i0);
        }
      }
      bool valid = false;
      if (has0) {
        const int cell = (int)(code0 & 63u), r = (int)(code0 >> 6);
        const int cx = cell >> 4, cy = (cell >> 2) & 3, cz = cell & 3;
        Arena ar;
        make_arena(mp.dx, cur.pos, ar);
        const int ocx = ar.corner[0] - geo.org[0], ocy = ar.corner[1] - geo.org[1], ocz = ar.corner[2] - geo.org[2];
        if (ocx != cx || ocy != cy || ocz != cz) {
          A.status[4] = 1;  // the storage invariant is broken (the caller moved particles without re-slotting them)
        } else {
          float vel[3], C[9];
          g2p_gather_lds(mp, ar, varena + AL::at(ocx, ocy, ocz), D_inv, vel, C);
          float pos[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) pos[d] = cur.pos[d] + vel[d] * mp.dt;
          float F[9], PF[9];
          advance_state<FLUID>(cur.F, C, mp.dt, F);
          float lpn[3];
          int nc[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) {  // base node / local position of the NEW position, exactly as make_arena derives them
            const float X = pos[d] * dxi;
            const float fl = floorf(X - 0.5f);
            nc[d] = (int)fl - geo.org[d];
            lpn[d] = X - fl;
          }
          const float pm = cur.m;
          float plj = 0.f;
          if constexpr (DP) plj = cur.logJp;
#ifdef ZS_X_NOMOVE
          const bool moved = false;
#else
          const bool moved = nc[0] != cx || nc[1] != cy || nc[2] != cz;
#endif
          // X - floor(X - 0.5) rounded up to 1.5, or X - 0.5 rounded up to an integer and left it below 0.5 (|X| < 1 only): the reference
          // applies base_node to the local position once more and takes the weights of d0 -+ 1 on the unchanged corner
          // (InterpolationKernel.hpp:108 on simulation/Utils.hpp:59-60; make_arena restates it).  The lane = cell consumers take the staged
          // lpn as d0; such a particle is scattered through the consumers' list instead, which folds d0 as the reference does.
#ifdef ZS_X_NOEDGE  // measurement / test-of-the-test build: the consumers take every staged lpn as d0
          const bool edge = false;
#else
          const bool edge = !(lpn[0] >= 0.5f && lpn[0] < 1.5f && lpn[1] >= 0.5f && lpn[1] < 1.5f && lpn[2] >= 0.5f && lpn[2] < 1.5f);
#endif
          bool outbox = false;   // it gets an outbox record (new cell in a neighbour bin: slot_rehome_kernel finds its slot; or fallback scatter)
          bool staged = !moved;  // {m, x', v', C', P F^T} staged for the consumers
          bool home = false;     // mover with a new slot inside this bin
          bool byList = false;   // stayer scattered by the consumers' list (see `edge`)
          bool keep = false;     // mover that found no new home (cell full, outbox full, moved too far): it stays in its OLD slot with its new
                                 // state, occupancy bit set -- reported (status [0] / [1] / [4]); re-slotting the storage recovers it
          unsigned recFlag = 0u; // record word SLR_FLAG: the record's grid contributions are still to be added (after the loop)
          float *rec = nullptr;
          POff<LW> o = particle_offset<LW>(ps.pos.chns, i0);  // where the particle lives after the step
          if (moved) {
            int code = 0;
            bool far = false;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              far = far || (unsigned)(nc[d] + 1) > 5u;
              code = code * 3 + (nc[d] < 0 ? 0 : (nc[d] > 3 ? 2 : 1));
            }
            const int dl = ((nc[0] & 3) * 4 + (nc[1] & 3)) * 4 + (nc[2] & 3);
            unsigned dcell = 0xffffffffu;  // destination cell of a record whose home slot_rehome_kernel has to find
            bool viaX = true;              // its grid contributions: consumers' global-atomic list (else: arrival queue of its new cell)
            if (far) {
              A.status[4] = 1;  // moved more than one cell in one step (CFL violated): not representable (scattered nowhere)
              viaX = false;
              keep = true;
            } else if (code == 13) {  // new cell inside this bin: a ticket of its LDS counter = a free round, from the bottom
              const int rr = nth_low_bit(~mask0[dl] & kmask, atomicAdd(&arrLocal[dl], 1u));
              if (rr >= 0) {
                home = true;
                o = particle_offset<LW>(ps.pos.chns, (rowBase + (size_t)rr) * 64 + (size_t)dl);
              } else {
                A.status[1] = 1;  // cell full: the particle is scattered but has no new slot
                keep = true;
              }
#ifndef ZS_X_NOINBIN
              const unsigned q = edge ? (unsigned)SL_ARRQ : atomicAdd(&arrCnt[par][dl], 1u);
              if (q < (unsigned)SL_ARRQ) {  // the lane of the new cell scatters it (arrival queue of the chunk)
                viaX = false;
                staged = true;
                arrQ[par][dl][q] = (unsigned short)((grp % SL_NG) * 64 + lane);
              }
#endif
            } else {
              outbox = true;
              const int dbin = nbrBin[code];
              if (dbin >= 0) dcell = (unsigned)dbin * 64u + (unsigned)dl;
              else A.status[2] = 1;  // the destination block is not in the partition: nowhere to live (sent != homed)
            }
            if (viaX) {
              const unsigned k = atomicAdd(&xCnt[par], 1u);
              if (k < (unsigned)SL_XQ) {
                staged = true;
                xq[par][k] = (unsigned)((grp % SL_NG) * 64 + lane) | ((unsigned)(nc[0] + 1) << 10) | ((unsigned)(nc[1] + 1) << 13) |
                             ((unsigned)(nc[2] + 1) << 16);
              } else {  // list full: a full record, scattered after the loop
                outbox = true;
                recFlag = 1u;
                atomicAdd(xOver, 1);
              }
            }
            if (outbox) {
              const int k = atomicAdd(outCount, 1);
              if (k < A.cap) {
                rec = A.moverRec + ((size_t)bin * A.cap + (size_t)k) * SL_REC;
#pragma unroll
                for (int d = 0; d < 3; ++d) rec[1 + d] = pos[d];
#pragma unroll
                for (int d = 0; d < 9; ++d) rec[4 + d] = F[d];
                if (WRITE_ALL || recFlag) {
#pragma unroll
                  for (int d = 0; d < 3; ++d) rec[SLR_V + d] = vel[d];
#pragma unroll
                  for (int d = 0; d < 9; ++d) rec[SLR_C + d] = C[d];
                }
                reinterpret_cast<unsigned *>(rec)[SLR_DCELL] = dcell;
                reinterpret_cast<unsigned *>(rec)[SLR_FLAG] = recFlag;
              } else {
                A.status[0] = 1;  // outbox full -- reported, the caller must react (raise outboxCap, re-slot)
                keep = !home;     // (a mover that already has its new slot only loses the fallback scatter of its grid terms)
              }
            }
            if (home || keep) {
              pstore_state<LW, FLUID>(ps.F, o, F);
              pstore<LW, 3>(ps.pos, o, pos);
              if (WRITE_ALL) {
                pstore<LW, 3>(ps.vel, o, vel);
                pstore<LW, 9>(ps.C, o, C);
              }
              if (home) atomicAdd(homed, 1);
            }
            if (!keep) {
              atomicOr(&clr[cell], 1u << r);  // its slot becomes a hole
              atomicAdd(sent, 1);
            }
          } else {
            pstore_state<LW, FLUID>(ps.F, o, F);
            pstore<LW, 3>(ps.pos, o, pos);
            if (WRITE_ALL) {
              pstore<LW, 3>(ps.vel, o, vel);
              pstore<LW, 9>(ps.C, o, C);
            }
            if (edge) {  // (see `edge`) a stayer, scattered by the list; a full list leaves it to its lane
              const unsigned k = atomicAdd(&xCnt[par], 1u);
              if (k < (unsigned)SL_XQ) {
                byList = true;
                xq[par][k] = (unsigned)((grp % SL_NG) * 64 + lane) | ((unsigned)(nc[0] + 1) << 10) | ((unsigned)(nc[1] + 1) << 13) |
                             ((unsigned)(nc[2] + 1) << 16);
              }
            }
          }
          {  // the plastic models may project the local copy of F (the stored / recorded F is the unprojected one, P2G.hpp:101)
            float lj = plj;
            model_stress<SMODEL>(mp.mat, lj, F, PF, C);
            if (outbox) {
              if (rec) {
                rec[0] = pm;
                rec[13] = lj;
                if (WRITE_ALL || recFlag) {
#pragma unroll
                  for (int d = 0; d < 9; ++d) rec[SLR_PF + d] = PF[d];
                }
              }
            }
            if (!moved || home || keep) {
              if constexpr (DP) pstore1<LW>(ps.logJp, o, lj);
              if (WRITE_ALL) pstore<LW, 9>(ps.stress, o, PF);
              if (moved) pstore1<LW>(ps.mass, o, pm);
            }
          }
          if (staged) {
            // staged AFTER the constitutive update, as in g2p2g_rs_producer: with m, x', v', C' dead before it the compiler
            // reuses their registers for the SVD at once and waits for the particle stores just issued (s_waitcnt vmcnt(1)
            // in front of the SVD: 2 ms per 64 Mi particles)
            valid = !moved && !byList;  // an in-bin mover is consumed by the lane of its NEW cell (arrival queue), not by the lane of its entry
            myStage[0 * 64 + lane] = pm;
#pragma unroll
            for (int d = 0; d < 3; ++d) myStage[(1 + d) * 64 + lane] = lpn[d];
#pragma unroll
            for (int d = 0; d < 3; ++d) myStage[(4 + d) * 64 + lane] = vel[d];
#pragma unroll
            for (int d = 0; d < 9; ++d) myStage[(7 + d) * 64 + lane] = C[d];
#pragma unroll
            for (int d = 0; d < 9; ++d) myStage[(16 + d) * 64 + lane] = PF[d];
          }
        }
      }
      {
        const unsigned long long vm = __ballot(valid);
        if (lane == 0) smask[grp % SL_NG] = vm;
      }

    }
    __syncthreads();  // the chunk is staged
    {  // ---------------------------------------------------------------- consume the rounds the chunk completed, its arrivals and its list
      // lane-derived values of this phase are re-derived here from an opaque copy of the lane number: hoisted out of the chunk loop they
      // would stay live through the produce phase, which has no register to spare
      int lane_c = lane;
      asm volatile("" : "+v"(lane_c));
      const unsigned long long lt = lanemask_lt();
      float acc[27][S::NA];
#pragma unroll
      for (int k = 0; k < 27; ++k)
#pragma unroll
        for (int q = 0; q < S::NA; ++q) acc[k][q] = 0.f;
      int iter = 0;
      const unsigned long long *const smask = smaskc;
      const int par = it % 3;
      const int produced = 512 * (it + 1) < total ? 512 * (it + 1) : total;
      const unsigned qn = arrCnt[par][lane_c];
#ifdef ZS_X_NOARR
      const int na = 0;
#else
      const int na = qn < (unsigned)SL_ARRQ ? (int)qn : SL_ARRQ;
#endif
      int ai = 0;
      if (CS == 0 && HALF == 0) arrCnt[(it + 1) % 3][lane_c] = 0u;  // the counters the NEXT chunk will use (last read one iteration ago)
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
        if (spos < 0 && pend) {  // a lane_c without a particle of its own in this round takes an arrival
          const unsigned p = arrQ[par][lane_c][ai++];
          spos = (int)(p >> 6) * (G2P2G_NF * 64) + (int)(p & 63u);
        }
        if (spos >= 0 && (iter & 1) == HALF) g2p2g_consume_set<CS>(mp, stage, spos, kscale, acc);  // the set's two waves take turns
        ++iter;
      }
      // movers of the chunk whose new cell is not a lane_c of this bin (or whose cell's arrival queue was full): this set's channels
      // of their 27 node terms straight to the grid.  Two list entries per pass: lane_c = (entry parity, stencil node); the channels of
      // the set are a compile-time loop, so only the node's weight formula (alpha + beta (s d0 + t)^2 per axis) is a per-lane_c constant
      const int nx = xCnt[par] < (unsigned)SL_XQ ? (int)xCnt[par] : SL_XQ;
      if (CS == 0 && HALF == 0 && lane_c == 0) xCnt[(it + 1) % 3] = 0u;
      {
        constexpr int NC = SIDE * SIDE * SIDE;
        const int node = lane_c & 31, half = lane_c >> 5;
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
#ifdef ZS_X_NOXQ
        if (false) {
#else
        if (node < 27 && nx > 0) {
#endif
#pragma unroll 1
          for (int k = half + 2 * HALF; k < nx; k += 4) {
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
              float *gp = A.gridB + ((size_t)bn * 7 + S::CH0) * NC + cell;
              const float Wm = Wt * (S::STRESS ? kscale : st[0]);
#pragma unroll
              for (int q = 0; q < S::NA; ++q) {
                float val;
                if (S::MASS && q == 0) {
                  val = Wm;  // mass
                } else {
                  // momentum d: m (v_d + C[., d] . xi); force d: -dt Dinv (P F^T[., d] . xi)
                  const int d = S::D0 + q - (S::MASS ? 1 : 0);
                  const int iC = (S::STRESS ? 16 : 7) + d;
                  float t = st[iC * 64] * xi[0];
                  t = fmaf(st[(iC + 3) * 64], xi[1], t);
                  t = fmaf(st[(iC + 6) * 64], xi[2], t);
                  if (!S::STRESS) t += st[(4 + d) * 64];  // v_d + (C . xi), the association of P2G.hpp:112
                  val = Wm * t;
                }
#ifndef ZS_X_NOXATOMIC
                if (val != 0.f) unsafeAtomicAdd(gp + q * NC, val);
#else
                if (val == 1234.5f) A.status[7] = cell;
#endif
              }
            } else if (S::MASS) {
              A.status[2] = 1;  // mass for a node whose block is not in the partition
            }
          }
        }
      }

      // the set's channels of the bin's arena belong to its two waves: first one, then (behind the barrier) the other; phases ordered
      // inside a wave (see g2p2g_body)
      float *a0 = parena + (size_t)S::CH0 * AL::CH + AL::at(lane_c >> 4, (lane_c >> 2) & 3, lane_c & 3);
      if (HALF == 0) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
          float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
          for (int q = 0; q < S::NA; ++q) g[q * AL::CH] += acc[k][q];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      __syncthreads();  // every wave has finished reading the staging ring; the first halves' sums are in the arena
      if (HALF == 1) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
          float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
          for (int q = 0; q < S::NA; ++q) g[q * AL::CH] += acc[k][q];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
    }
  }
}
