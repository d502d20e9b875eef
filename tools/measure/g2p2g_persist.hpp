// g2p2g_persist.hpp -- MEASUREMENT CODE, not part of the product library: the persistent variant of the role-split fused G2P2G kernel
// (r02: 4.82 ms against 4.56 ms for one workgroup per bin).  Textually included by zpc_amd/csrc/mpm_device.hpp when a
// measurement build defines ZS_ROCM_WITH_PERSIST (tools/ablate.sh does); it uses the declarations of that header and is meaningless alone.
// ---------------------------------------------------------------------------------------------------------------------------
// Persistent role-split pass (compiled with -DZS_ROCM_WITH_PERSIST only: a measured alternative, not the default -- DESIGN.md).  Cycle stamps of g2p2g_rs_kernel (tools/ablate.sh PROBE, 64 Mi particles, per
// workgroup): 36 300 cycles from entry to exit, of which only the 23 700 of the chunk loop issue VALU work -- 8 800 go to the head
// of the bin (bin start / cell counts / block key / nbr row, then grid velocities and the first records: dependent HBM round
// trips), 5 900 to the drain iteration and the barriers, 1 300 to the tail, and ~5 000 more pass between two workgroups on a
// CU slot (dispatch).  One wave alone issues a VALU instruction every ~5 cycles, so a SIMD holding four such waves, each busy a
// third of the time, runs at under half of its issue rate.  Here a workgroup stays resident and walks a sequence of bins in
// groups of eight (one 8^3 block): the group's metadata is read once into LDS, and inside the group the producer / consumer
// pipeline of g2p2g_rs_kernel never stops -- the producers run on into the next bin (its velocity arena is filled one slot
// ahead into the other LDS buffer, its first records are requested one chunk ahead) while the consumers finish the previous
// one: flush of the register accumulators, then the in-bin movers, then the global flush, one per slot, each consumer wave
// flushing the arena channels it owns.  A "slot" is one chunk (four rounds) of one bin; every bin takes at least two slots so
// that the three stages of two consecutive bins never meet in the single LDS arena.
constexpr int PG_GB = 8;        // bins per group
constexpr int PG_MQ_CAP = 256;  // in-bin movers per bin taken through the LDS queue (more: exact path)
struct PersistMeta {            // one group's metadata in LDS
  int start[PG_GB + 1];
  int key[PG_GB][3];
  int nbr[PG_GB][8];
  int chunks[PG_GB];
  unsigned cnt[PG_GB][64];
};
struct PersistShared {
  float varena[2][3 * ArenaLds::CH];
  float parena[7 * ArenaLds::CH];
  float stage[2 * 4 * G2P2G_NF * 64];
  unsigned long long smask[2 * 4];
  int mq[2][PG_MQ_CAP];
  int mqCount[2];
  PersistMeta meta[2];
};
struct PersistArgs {
  const float *gridA;
  float *gridB;
  const int *binStart;
  const unsigned *cellCount;
  const int *nbr;
  int *staleG, *staleGCount, *staleP, *stalePCount;
  int binBase, nbins;
};

// ROLE 0..3: producer of round 4c + ROLE; ROLE 4..7: consumer of channel set ROLE - 4.  ONE function for both so that every wave
// executes the same sequence of barriers by construction.
template <int SIDE, int SMODEL, int LW, bool WRITE_ALL, int ROLE>
__device__ __forceinline__ void g2p2g_persist_role(const MpmDev &mp, const ParticlesDev &ps, const BhtDev &t, const PersistArgs &A,
                                                   PersistShared &sh) {
  using AL = ArenaLds;
  constexpr bool PROD = ROLE < 4;
  constexpr int W = ROLE & 3;
  constexpr int CS = ROLE & 3;
  using CSet = ConsumerSet<CS>;
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool FLUID = model_is_fluid(SMODEL);
  constexpr int NC = SIDE * SIDE * SIDE;
  constexpr int BPB = bins_per_block<SIDE>();
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int ctid = tid - 256;  // consumer thread index 0..255 (consumers only)
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = 1.0f / mp.dx;
  const float D_inv = 4.f * dxi * dxi;
  const float kscale = -mp.dt * D_inv;
  const int kk = SIDE / mp.kscale;  // block key -> cells
  // values read back from LDS are wave-uniform here, but the compiler cannot know: keep them (and the control flow that hangs
  // on them) scalar
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  // the rarely executed stages derive addresses from the lane / thread index; the values are loop invariant, and hoisted out of the
  // slot loop they would sit in dozens of VGPRs for the whole kernel: launder the index where such a stage starts
  auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };

  // producer state
  RecG<LW, DP, FLUID> cur, nxt;
  int i0 = 0, i1 = 0;
  bool has0 = false, has1 = false;
  RoundWalk walk(0u, 0);
  // consumer state
  float acc[PROD ? 1 : 27][PROD ? 1 : CSet::NA];
  if constexpr (!PROD) {
#pragma unroll
    for (int k = 0; k < 27; ++k)
#pragma unroll
      for (int q = 0; q < CSet::NA; ++q) acc[k][q] = 0.f;
  }
  int nmPending = 0;
  // pipeline bookkeeping (identical in every wave)
  int s = 0;                       // slot counter: stage / smask parity
  int q = 0;                       // bins started so far: varena / mq parity
  int e1 = -1, e2 = -1, e3 = -1;   // bins whose last slot was s-1 / s-2 / s-3: (metaParity << 8) | (j << 4) | (q & 1)
  bool prevWork = false;

  // origin (cells) of bin j of the group whose metadata is M, and its offset inside its block
  auto bin_geom = [&](const PersistMeta &M, int binAbs, int j, int (&o)[3], int (&org)[3]) {
    const int sub = binAbs % BPB;
    o[0] = SIDE == 4 ? 0 : ((sub >> 2) & 1) * 4;
    o[1] = SIDE == 4 ? 0 : ((sub >> 1) & 1) * 4;
    o[2] = SIDE == 4 ? 0 : (sub & 1) * 4;
#pragma unroll
    for (int d = 0; d < 3; ++d) org[d] = uni(M.key[j][d]) * kk + o[d];
  };
  // ---- consumer stages of one slot (e3: global flush of the channels this wave owns, e2: in-bin movers, e1: accumulator flush)
  auto consumer_stages = [&](int gFirstBinOf[2]) {
    if constexpr (!PROD) {
      if (prevWork) {
        const int par = (s - 1) & 1;
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
          const unsigned long long vm = sh.smask[par * 4 + rr];
          if (vm == 0ull) continue;
          if ((vm >> lane) & 1ull) g2p2g_consume_set<CS>(mp, sh.stage + (size_t)(par * 4 + rr) * (G2P2G_NF * 64), lane, kscale, acc);
        }
      }
#ifndef ZS_DBG_NO_E3
      if (e3 >= 0) {  // global flush of my channels, then they are cleared for the next bin
        const PersistMeta &M = sh.meta[e3 >> 8];
        const int j = (e3 >> 4) & 15;
        const int binAbs = A.binBase + gFirstBinOf[e3 >> 8] + j;
        int o[3], org[3];
        bin_geom(M, binAbs, j, o, org);
        for (int n = fresh(lane); n < 216; n += 64) {
          const int x = n / 36, y = (n / 6) % 6, z = n % 6;
          int slot, cell;
          arena_to_grid<SIDE>(o, x, y, z, slot, cell);
          const int bn = M.nbr[j][slot];
          float *a = sh.parena + (size_t)CSet::CH0 * AL::CH + AL::at(x, y, z);
          if (bn >= 0) {
            float *g = A.gridB + ((size_t)bn * 7 + CSet::CH0) * NC + cell;
#pragma unroll
            for (int c = 0; c < CSet::NA; ++c) {
              const float v = a[c * AL::CH];
              if (v != 0.f) unsafeAtomicAdd(g + c * NC, v);
            }
          } else if (CS == 0 && a[0] != 0.f) {
            A.staleGCount[9] = 1;  // mass for a node whose block is not in the partition
          }
#pragma unroll
          for (int c = 0; c < CSet::NA; ++c) a[c * AL::CH] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
#endif
      if (e1 >= 0) {  // the bin's last chunk has just been consumed: accumulators -> my channels of the arena
        const int l2 = fresh(lane);
        float *a0 = sh.parena + (size_t)CSet::CH0 * AL::CH + AL::at(l2 >> 4, (l2 >> 2) & 3, l2 & 3);
#pragma unroll
        for (int k = 0; k < 27; ++k) {
          float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
          for (int c = 0; c < CSet::NA; ++c) {
            g[c * AL::CH] += acc[k][c];
            acc[k][c] = 0.f;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
    }
  };
  // ---- in-bin movers (slot e2 of a bin: every consumer has flushed its accumulators, the global flush comes one slot later).
  // Run by the PRODUCER threads: they hold the particle ports anyway, and the pass is a handful of particles per bin.
  auto mover_stage = [&](int gFirstBinOf[2]) {
    if constexpr (PROD) {
      if (e1 >= 0) {
        const int c = uni(sh.mqCount[e1 & 1]);  // complete: the producers finished this bin before the last barrier
        nmPending = c < PG_MQ_CAP ? c : PG_MQ_CAP;
      }
      if (e2 >= 0) {  // in-bin movers of that bin: dense pass of the 256 producer threads, LDS atomics on all channels
        const PersistMeta &M = sh.meta[e2 >> 8];
        const int j = (e2 >> 4) & 15, qp = e2 & 1;
        const int binAbs = A.binBase + gFirstBinOf[e2 >> 8] + j;
        int o[3], org[3];
        bin_geom(M, binAbs, j, o, org);
        const int nm = nmPending;
        for (int qi = fresh(tid); qi < nm; qi += 256) {
          const size_t i = (size_t)sh.mq[qp][qi];
          auto cload = [&](const Port<float> &p, int comp) {
            return __hip_atomic_load(p.base + p.off(i) + (size_t)comp * p.cstride(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          };
          const float m = ps.mass.base[ps.mass.off(i)];
          float pos[3], vel[3], C[9], PF[9];
#pragma unroll
          for (int d = 0; d < 3; ++d) { pos[d] = cload(ps.pos, d); vel[d] = cload(ps.vel, d); }
#pragma unroll
          for (int d = 0; d < 9; ++d) { C[d] = cload(ps.C, d); PF[d] = cload(ps.stress, d) * kscale; }
          Arena ar;
          make_arena(mp.dx, pos, ar);
          const int kx = ar.corner[0] - org[0], ky = ar.corner[1] - org[1], kz = ar.corner[2] - org[2];
          if ((unsigned)kx >= 4u || (unsigned)ky >= 4u || (unsigned)kz >= 4u) {
            A.staleP[atomicAdd(A.stalePCount, 1)] = (int)i;
            continue;
          }
          float *a0 = sh.parena + AL::at(kx, ky, kz);
          // a ROLLED loop over the 27 nodes (weights picked with selects): this path is rare, and unrolled it would push the
          // consumer's 54 accumulators out of the register file
#pragma unroll 1
          for (int nd = 0; nd < 27; ++nd) {
            const int a = nd / 9, b = (nd / 3) % 3, c = nd % 3;
            const float wa = a == 0 ? ar.w[0][0] : (a == 1 ? ar.w[0][1] : ar.w[0][2]);
            const float wb = b == 0 ? ar.w[1][0] : (b == 1 ? ar.w[1][1] : ar.w[1][2]);
            const float wc = c == 0 ? ar.w[2][0] : (c == 1 ? ar.w[2][1] : ar.w[2][2]);
            const float Wt = wa * wb * wc;
            const float x0 = (float)a * mp.dx - ar.lp[0], x1 = (float)b * mp.dx - ar.lp[1], x2 = (float)c * mp.dx - ar.lp[2];
            float *g = a0 + (a * AL::SX + b * AL::SY + c);
            atomicAdd(g, Wt * m);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              atomicAdd(g + (1 + d) * AL::CH, Wt * m * (vel[d] + (C[d] * x0 + C[3 + d] * x1 + C[6 + d] * x2)));
              atomicAdd(g + (4 + d) * AL::CH, (PF[d] * x0 + PF[3 + d] * x1 + PF[6 + d] * x2) * Wt);
            }
          }
        }
        if (tid == 0) sh.mqCount[qp] = 0;  // every producer thread read its count one slot ago
      }
    }
  };

  int gFirstBinOf[2] = {0, 0};  // first bin (relative to binBase) of the group in each metadata buffer
  // ONE loop, one call site per stage: with the stages inlined at several places the compiler carries a second copy of the
  // accumulators between them (54 more VGPRs).  An iteration = [group head, when a new group starts] + exactly one slot.
  enum { NEED_GROUP = 0, IN_GROUP = 1, DRAIN = 2 };
  int mode = NEED_GROUP, drainLeft = 0;
  int g = (int)blockIdx.x, gp = 0, j = 0, c = 0;
  float vnext[3] = {0.f, 0.f, 0.f};
  for (;;) {
    if (mode == NEED_GROUP) {
      const bool pending = (e1 >= 0 && (e1 >> 8) == gp) || (e2 >= 0 && (e2 >> 8) == gp) || (e3 >= 0 && (e3 >> 8) == gp);
      if (g * PG_GB >= A.nbins) {
        mode = DRAIN;
        drainLeft = 4;  // consume, flush, movers, global flush of the last bin
      } else if (!pending) {  // (pending: stages that still need the metadata buffer about to be overwritten run first, in an empty slot)
        PersistMeta &M = sh.meta[gp];
        gFirstBinOf[gp] = g * PG_GB;
        {  // ---- the group's metadata: wave jj reads bin jj (512 threads = 8 bins x 64 cells)
          const int jj = tid >> 6;
          const int binRel = g * PG_GB + jj;
          const bool live = binRel < A.nbins;
          const int binAbs = A.binBase + (live ? binRel : A.nbins - 1);
          const unsigned cc = live ? A.cellCount[(size_t)binAbs * 64 + lane] : 0u;
          int st0 = 0, st1 = 0, key = 0, nb = 0;
          if (lane == 0) { st0 = A.binStart[binAbs]; st1 = A.binStart[binAbs + 1]; }
          const int block = binAbs / BPB;
          if (lane < 3) key = t.activeKeys[3 * (size_t)block + lane];
          if (lane < 8) nb = A.nbr[(size_t)block * 8 + lane];
          unsigned mx = cc;
#pragma unroll
          for (int sft = 32; sft >= 1; sft >>= 1) {
            const unsigned o2 = (unsigned)__shfl_xor((int)mx, sft, 64);
            mx = o2 > mx ? o2 : mx;
          }
          M.cnt[jj][lane] = cc;
          if (lane == 0) {
            M.start[jj] = st0;
            if (jj == PG_GB - 1 || binRel + 1 >= A.nbins) M.start[jj + 1] = st1;
            M.chunks[jj] = live && st0 != st1 ? (int)((mx + 3u) >> 2) : 0;
          }
          if (lane < 3) M.key[jj][lane] = key;
          if (lane < 8) M.nbr[jj][lane] = nb;
        }
        __syncthreads();
        j = 0;
        while (j < PG_GB && uni(M.chunks[j]) == 0) ++j;
        if (j >= PG_GB) {  // nothing in this group
          g += (int)gridDim.x;
          continue;
        }
        // ---- head of the group: first records + velocity arena of its first bin
        if constexpr (PROD) {
          walk = RoundWalk(M.cnt[j][lane], uni(M.start[j]));
          has0 = false;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int i;
            bool a;
            const bool h = walk.next(i, a);
            if (r == W) { i0 = i; has0 = h; }
          }
          if (has0) cur.load(ps, (size_t)i0);
          if (tid < 216) {
            int o[3], org[3];
            bin_geom(M, A.binBase + g * PG_GB + j, j, o, org);
            const int tt = fresh(tid);
            const int x = tt / 36, y = (tt / 6) % 6, z = tt % 6;
            int slot, cell;
            arena_to_grid<SIDE>(o, x, y, z, slot, cell);
            const int bn = M.nbr[j][slot];
            float *a = sh.varena[q & 1] + AL::at(x, y, z);
            const float *gsrc = A.gridA + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) a[ch * AL::CH] = bn >= 0 ? gsrc[ch * NC] : 0.f;
          }
        }
        __syncthreads();
        c = 0;
        mode = IN_GROUP;
      }
    }
    if (mode == DRAIN) {
      if (drainLeft == 0) break;
      --drainLeft;
    }
    // ------------------------------------------------------------------------------------------------ one slot
    const bool haveBin = mode == IN_GROUP;
    PersistMeta &M = sh.meta[gp];
    const int nch = haveBin ? uni(M.chunks[j]) : 0;
    const int nslots = nch < 2 ? 2 : nch;
    const bool work = haveBin && c < nch;
    const bool lastSlot = haveBin && c == nslots - 1;
    int jn = PG_GB;
    if (haveBin) {
      jn = j + 1;
      while (jn < PG_GB && uni(M.chunks[jn]) == 0) ++jn;
    }
    if constexpr (PROD) {
      mover_stage(gFirstBinOf);
      if (work) {
          // ---- look-ahead: the chunk after this one (same bin, or the first chunk of the next bin of the group)
          has1 = false;
          const bool sameBin = c + 1 < nch;
          if (sameBin || jn < PG_GB) {
            if (!sameBin) walk = RoundWalk(M.cnt[jn][lane], uni(M.start[jn]));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int i;
              bool a;
              const bool h = walk.next(i, a);
              if (r == W) { i1 = i; has1 = h; }
            }
            if (has1) nxt.load(ps, (size_t)i1);  // in flight during this chunk
          }
          // ---- this chunk
          const int par = s & 1;
          float *myStage = sh.stage + (size_t)(par * 4 + W) * (G2P2G_NF * 64);
          const float *varena = sh.varena[q & 1];
          int *mq = sh.mq[q & 1];
          int *mqCount = &sh.mqCount[q & 1];
          int o[3], org[3];
          bin_geom(M, A.binBase + g * PG_GB + j, j, o, org);
          bool valid = false;
          if (has0) {
            Arena ar;
            make_arena(mp.dx, cur.pos, ar);
            const int ocx = ar.corner[0] - org[0], ocy = ar.corner[1] - org[1], ocz = ar.corner[2] - org[2];
            if ((unsigned)ocx >= 4u || (unsigned)ocy >= 4u || (unsigned)ocz >= 4u) {
              A.staleG[atomicAdd(A.staleGCount, 1)] = i0;  // outside the bin: exact gather + scatter afterwards
              if ((unsigned)(ocx + 4) >= 12u || (unsigned)(ocy + 4) >= 12u || (unsigned)(ocz + 4) >= 12u) A.staleGCount[8] = 1;
            } else {
              float vel[3], C[9];
              g2p_gather_lds(mp, ar, varena + AL::at(ocx, ocy, ocz), D_inv, vel, C);
              const POff<LW> po = particle_offset<LW>(ps.pos.chns, (size_t)i0);
              float pos[3];
#pragma unroll
              for (int d = 0; d < 3; ++d) pos[d] = cur.pos[d] + vel[d] * mp.dt;
              float F[9], PF[9];
              advance_state<FLUID>(cur.F, C, mp.dt, F);
              pstore_state<LW, FLUID>(ps.F, po, F);
              pstore<LW, 3>(ps.pos, po, pos);
              float lpn[3];
              int nc[3];
#pragma unroll
              for (int d = 0; d < 3; ++d) {  // base node / local position of the NEW position, exactly as make_arena derives them
                const float X = pos[d] * dxi;
                const float fl = floorf(X - 0.5f);
                nc[d] = (int)fl - org[d];
                lpn[d] = X - fl;
              }
              const int ncx = nc[0], ncy = nc[1], ncz = nc[2];
              const bool moved = ncx != cx || ncy != cy || ncz != cz;
              // everything that does not need the stress leaves the registers BEFORE the constitutive update (the 3x3 SVD is the
              // register peak of a producer: m, x', v', C' would otherwise stay live across it)
              if (WRITE_ALL || moved) {
                pstore<LW, 3>(ps.vel, po, vel);
                pstore<LW, 9>(ps.C, po, C);
              }
              if (!moved) {
                myStage[0 * 64 + lane] = cur.m;
#pragma unroll
                for (int d = 0; d < 3; ++d) myStage[(1 + d) * 64 + lane] = lpn[d];
#pragma unroll
                for (int d = 0; d < 3; ++d) myStage[(4 + d) * 64 + lane] = vel[d];
#pragma unroll
                for (int d = 0; d < 9; ++d) myStage[(7 + d) * 64 + lane] = C[d];
              }
              {  // F has been stored above: the plastic models may project this local copy
                float lj = 0.f;
                if constexpr (DP) lj = cur.logJp;
                model_stress<SMODEL>(mp.mat, lj, F, PF, C);
                if constexpr (DP) pstore1<LW>(ps.logJp, po, lj);
              }
              if (WRITE_ALL || moved) pstore<LW, 9>(ps.stress, po, PF);
              if (moved) {
                bool queued = false;
                if ((unsigned)ncx < 4u && (unsigned)ncy < 4u && (unsigned)ncz < 4u) {
                  const int slotq = atomicAdd(mqCount, 1);
                  if (slotq < PG_MQ_CAP) {
                    mq[slotq] = i0;
                    queued = true;
                  }
                }
                if (!queued) {
                  A.staleP[atomicAdd(A.stalePCount, 1)] = i0;  // left the bin during this step: exact scatter afterwards
                  if ((unsigned)(ncx + 4) >= 12u || (unsigned)(ncy + 4) >= 12u || (unsigned)(ncz + 4) >= 12u) A.staleGCount[8] = 1;
                }
              } else {
                valid = true;
#pragma unroll
                for (int d = 0; d < 9; ++d) myStage[(16 + d) * 64 + lane] = PF[d];
              }
            }
          }
          {
            const unsigned long long vm = __ballot(valid);
            if (lane == 0) sh.smask[par * 4 + W] = vm;
          }
          cur = nxt;
          has0 = has1;
          i0 = i1;
      }
    } else {
      // velocity arena of the next bin, by the consumer threads (they have the slack): requested at the END of the bin's
      // second-to-last slot (every bin has at least two), stored to the other buffer at the START of its last slot -- the values
      // are in flight across the barrier only, not across the accumulation
      if (lastSlot && jn < PG_GB && ctid < 216) {
        const int ct = fresh(ctid);
        const int x = ct / 36, y = (ct / 6) % 6, z = ct % 6;
        float *a = sh.varena[(q + 1) & 1] + AL::at(x, y, z);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) a[ch * AL::CH] = vnext[ch];
      }
      consumer_stages(gFirstBinOf);
      if (haveBin && c == nslots - 2 && jn < PG_GB && ctid < 216) {
        int o[3], org[3];
        bin_geom(M, A.binBase + g * PG_GB + jn, jn, o, org);
        const int ct = fresh(ctid);
        const int x = ct / 36, y = (ct / 6) % 6, z = ct % 6;
        int slot, cell;
        arena_to_grid<SIDE>(o, x, y, z, slot, cell);
        const int bn = M.nbr[jn][slot];
        const float *gsrc = A.gridA + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) vnext[ch] = bn >= 0 ? gsrc[ch * NC] : 0.f;
      }
    }
    __syncthreads();
    e3 = e2;
    e2 = e1;
    e1 = lastSlot ? ((gp << 8) | (j << 4) | (q & 1)) : -1;
    prevWork = work;
    ++s;
    if (haveBin) {
      if (lastSlot) {
        j = jn;
        c = 0;
        ++q;
        if (j >= PG_GB) {
          mode = NEED_GROUP;
          g += (int)gridDim.x;
          gp ^= 1;
        }
      } else {
        ++c;
      }
    }
  }
}

template <int SIDE, int SMODEL, int LW, bool WRITE_ALL>
static __global__ __launch_bounds__(512, 4) void g2p2g_persist_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, PersistArgs A) {
  __shared__ PersistShared sh;
  const int tid = (int)threadIdx.x, w = tid >> 6;
  for (int k = tid; k < 7 * ArenaLds::CH; k += 512) sh.parena[k] = 0.f;
  if (tid < 2) sh.mqCount[tid] = 0;
  __syncthreads();
#ifdef ZS_DBG_ROLE  // register-pressure triage: compile one role only
  g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, ZS_DBG_ROLE>(mp, ps, t, A, sh);
#else
  if (w == 0) g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 0>(mp, ps, t, A, sh);
  else if (w == 1) g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 1>(mp, ps, t, A, sh);
  else if (w == 2) g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 2>(mp, ps, t, A, sh);
  else if (w == 3) g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 3>(mp, ps, t, A, sh);
  else if (w == 4) g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 4>(mp, ps, t, A, sh);
  else if (w == 5) g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 5>(mp, ps, t, A, sh);
  else if (w == 6) g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 6>(mp, ps, t, A, sh);
  else g2p2g_persist_role<SIDE, SMODEL, LW, WRITE_ALL, 7>(mp, ps, t, A, sh);
#endif
}
