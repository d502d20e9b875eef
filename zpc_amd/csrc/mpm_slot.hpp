// mpm_slot.hpp -- device code shared by the kernels of the fused G2P2G step on SLOTTED particle storage (mpm_slotted.hip: one workgroup
// per bin; mpm_slotblk.hip: one workgroup per 8^3 block, its bins pipelined back to back).  Storage, mover protocol: mpm_slotted.hip.
#pragma once
#include "mpm_device.hpp"

namespace zsr {

constexpr int SL_REC = 48;    // floats per outbox record (192 bytes = three 64-byte lines).  First line, all slot_rehome_kernel reads without
                              // writeAll: [0] m, [1] x(3), [4] F(9), [13] logJp, [14] destination cell (bin * 64 + lane; ~0: none),
                              // [15] bit 0: its grid contributions are still to be added, bits 1..: the slot it left inside its bin
                              // (round * 64 + cell: where slot_rehome_kernel puts it back when the destination cell has no free round);
                              // then [16] v(3), [19] C(9), [28] P F^T(9) (written for writeAll steps and for flagged records only)
constexpr int SLR_DCELL = 14, SLR_FLAG = 15, SLR_V = 16, SLR_C = 19, SLR_PF = 28;

#ifdef ZS_SLOT_PROBE  // measurement-only build (tools/ablate_slot.sh PROBE): cycle stamps of a workgroup's phases, summed over sampled workgroups
static __device__ unsigned long long g_slot_probe[32];
#define SLP_SAMPLED ((blockIdx.x & 63) == 0)
#define SLP_T0(name) const unsigned long long name = __builtin_readcyclecounter()
#define SLP_ADD(slot, t0) do { if ((threadIdx.x & 63) == 0 && SLP_SAMPLED) atomicAdd(&g_slot_probe[slot], (unsigned long long)(__builtin_readcyclecounter() - (t0))); } while (0)
#define SLP_ACC(var, t0) var += __builtin_readcyclecounter() - (t0)
#define SLP_PUT(slot, v) do { if ((threadIdx.x & 63) == 0 && SLP_SAMPLED) atomicAdd(&g_slot_probe[slot], (unsigned long long)(v)); } while (0)
#define SLP_SEG(k) do { if (seg) { const unsigned long long tn_ = __builtin_readcyclecounter(); seg[k] += tn_ - tseg_; tseg_ = tn_; } } while (0)
#define SLP_SEG0() unsigned long long tseg_ = __builtin_readcyclecounter()
#else
#define SLP_T0(name) do { } while (0)
#define SLP_ADD(slot, t0) do { } while (0)
#define SLP_ACC(var, t0) do { } while (0)
#define SLP_PUT(slot, v) do { } while (0)
#define SLP_SEG(k) do { } while (0)
#define SLP_SEG0() do { } while (0)
#endif

struct SlotArgs {
  const float *gridA;
  float *gridB;
  unsigned *cellMask;   // [nbins][64] occupancy of the K rounds of every cell
  int K;
  const int *nbr;       // [nblocks][8]  blocks at offsets {0,1}^3 (arena -> grid)
  const int *nbr27;     // [nblocks][27] blocks at offsets {-1,0,1}^3 (movers: arena flush, destination bin)
  int *moverCount;      // [nbins] movers of the bin in this step (diagnostic)
  unsigned *claim;      // [2][nbinsAll][64]: [0] this step's arrivals (high 16 bits: from inside the bin, low 16: from other bins), [1] rounds
                        // vacated in this step; zero between steps (slot_commit_kernel folds both into cellMask)
  float *moverRec;      // [nbins][cap][SL_REC] outbox records: movers that left their bin (or found the arrival queue of their cell full)
  int *status;          // [0] outbox full, [1] cell full (K), [2] mass / a mover for a block outside the partition, [3] a particle lives in a
                        // block next to the partition's edge (blockEdge: re-partition soon), [4] a particle was not stored under its cell;
                        // [8 .. 8 + 256) movers sent, [264 .. 264 + 256) movers re-homed (running sums spread over 256 words each; equal
                        // after every step: a mover that finds no new home keeps or gets back its old slot -- no particle is ever dropped)
  int binBase, nbins, nbinsAll;
  int cap;              // outbox records per bin and step (caller's choice: a bin holds 512 particles at 8 per cell)
  const unsigned char *blockEdge;  // [nblocks] or NULL: 1 = a block of {-1..2}^3 around this one is not in the partition (zs_rocm_mpm_partition_edge)
  unsigned long long *signal;      // NULL, or a running count: every workgroup of the blocks [0, signalBlocks) adds 1 when its grid sums are in L2
  int signalBlocks;                // (8^3 blocks only: the multi-GPU step's exchange stream starts behind the boundary blocks of ONE launch, dist.hip)
};

constexpr int SL_NCTR = 256, SL_SENT = 8, SL_DELIVERED = 8 + SL_NCTR;  // layout of the status words (zs_rocm.h: ZS_ROCM_SLOT_STATUS_WORDS)
// bin next to `bin` in direction code (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1), dx, dy, dz in {-1, 0, 1}: same block or the block next to it
template <int SIDE> __device__ __forceinline__ int neighbour_bin(const int *nbr27, int block, int bin, int code) {
  if constexpr (SIDE == 4) {
    return nbr27[(size_t)block * 27 + code];
  } else {
    const int dd[3] = {code / 9 - 1, (code / 3) % 3 - 1, code % 3 - 1};
    const int sub = bin & 7;
    int sx[3] = {((sub >> 2) & 1) + dd[0], ((sub >> 1) & 1) + dd[1], (sub & 1) + dd[2]}, bo[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      bo[d] = sx[d] < 0 ? -1 : (sx[d] > 1 ? 1 : 0);
      sx[d] &= 1;
    }
    const int nb = nbr27[(size_t)block * 27 + ((bo[0] + 1) * 9 + (bo[1] + 1) * 3 + (bo[2] + 1))];
    return nb < 0 ? -1 : nb * 8 + ((sx[0] * 2 + sx[1]) * 2 + sx[2]);
  }
}

// ------------------------------------------------------------------------------------------------------------------ main kernel
// Packed rounds.  The storage keeps lane = cell (a consumer lane accumulates the 27 nodes of ITS cell in registers), so a bin has as
// many rounds as its fullest cell: 10-12 for a falling column that averages 7.5 particles per cell, i.e. a third of all (round, lane)
// slots are holes.  The producers -- gather, advection, F, SVD + model: two thirds of the step's instructions -- do not need lane =
// cell (they read the velocity arena by the particle's own base node), so they walk the bin's OCCUPIED slots instead, 64 at a time in
// round-major order: entry e of the bin <-> (round r, cell c) through a table built from the occupancy words at the head of the
// kernel.  Results are staged by entry; a consumer lane finds the entry of (r, its cell) by the same enumeration
// (off[r] + popcount of the occupied cells below it) and consumes a round as soon as all of its entries have been produced.
//
// Movers (r03; r02 ran a second kernel in which every bin pulled from the outboxes of its 27 neighbours and added the arrivals to the
// grid: 1.8 ms per step of the 64 Mi-particle column for 5 % of the particles, a chain of dependent loads per bin).  A particle whose
// base node changes is finished by the workgroup that moved it:
//   * new cell inside the bin (three of four movers of a drifting cloud): the producer draws a ticket from the destination cell's LDS
//     counter -- ticket t = the t-th lowest round that was free at the start of the step --, stores the particle's state straight into
//     that slot, stages {m, x', v', C', P F^T} like a stayer's and puts the staged position on the destination cell's arrival queue.
//     The consumer lane of that cell takes queued arrivals in iterations in which it has no particle of its own (a third of a lane's
//     round slots are holes): the grid contributions of an in-bin mover cost no additional instruction slot;
//   * new cell in a neighbour bin (or arrival queue full): the 40-float record goes to the bin's outbox.  In the LAST iteration of the
//     chunk loop, in which the producer waves have nothing to produce (the consumers still work on the last chunk), they read the
//     records back and add their 27 x 7 node terms straight to the grid with lane = (node, channel): 3 global float atomics per
//     record and wave-instruction, fire and forget -- the same atomics the arena flush issues, no LDS read-modify-write chain, no
//     dependence between records.  (Measured dead ends: ds_add_f32 into a wider LDS arena, ~3 cycles per lane: +4 ms; returning
//     global ticket atomics in the producer loop: ~5 us round trip under load, +2.6 ms; one wave per channel walking the records with
//     plain LDS read-add-write: latency-bound, 6 us per workgroup.)  The record's new home is found by slot_rehome_kernel after the
//     step: one thread per record, ticket from the destination cell's global counter, rounds above the in-bin arrivals.
// Departures and ticket counts are folded into the occupancy words by slot_commit_kernel.
#ifndef ZS_PROD_XLIST
#define ZS_PROD_XLIST 1  // the list of the last chunk is scattered by the (then idle) producer waves
#endif
constexpr int SL_NG = 9;       // staged entry groups of 64: a chunk being produced (4) + the chunk being consumed (4) + a straddling round
constexpr int SL_KMAX = 32;    // rounds per bin the 32-bit occupancy words allow
#ifndef ZS_SL_ARRQ
#define ZS_SL_ARRQ 4
#endif
constexpr int SL_ARRQ = ZS_SL_ARRQ;     // in-bin arrivals one cell takes per chunk through the consumers' queue
constexpr int SL_XQ = 64;      // movers per chunk whose grid contributions the consumers add with global atomics (new cell in another bin, or
                               // arrival queue full); more: scattered from their outbox records after the loop

// t-th (0-based) lowest set bit of w, or -1
__device__ __forceinline__ int nth_low_bit(unsigned w, unsigned t) {
  for (unsigned k = 0; k < t && w; ++k) w &= w - 1u;
  return w ? __ffs((int)w) - 1 : -1;
}
struct SlotShared {  // views of the kernel's LDS arrays
  float *varena;                     // [3 * ArenaLds::CH] node velocities of the bin
  float *parena;                     // [7 * ArenaLds::CH] the bin's P2G arena
  float *stage;                      // [SL_NG * G2P2G_QF * 64]
  unsigned long long *smask;         // [SL_NG]
  unsigned short *tab;               // [SL_KMAX * 64] entry -> round * 64 + cell
  unsigned *mask0;                   // [64] occupancy of the bin's cells at the start of the step
  unsigned *clr;                     // [64] rounds vacated in this step
  unsigned *arrLocal;                // [64] in-bin arrivals of the cell in this step (ticket counter)
  const int *nbrBlk;                 // [27] the grid blocks around this bin's block (-1: not in the partition)
  const int *nbrBin;                 // [27] the bins around this one (direction code (dx + 1) 9 + (dy + 1) 3 + dz + 1)
  unsigned (*arrCnt)[64];            // [3][64] arrivals queued for the consumers, by chunk number mod 3
  unsigned short (*arrQ)[64][SL_ARRQ];  // [3][64][SL_ARRQ]
  unsigned *xCnt;                    // [3] entries of xq, by chunk number mod 3
  unsigned (*xq)[SL_XQ];             // [3][SL_XQ] staged position | (new cell + 1, 3 bits per axis) << 10
  int *outCount, *sent, *homed, *xOver;
};

// One stencil node x one grid channel of ONE particle, straight to the grid (P2G.hpp:104-124): node in 0..26, ch in 0..6; the
// particle's new base node is `nc` cells from the origin of the bin's block, its position inside it `d0` (normalised, 0.5 .. 1.5);
// fm / fv / fC / fP: accessors of m, v_d, C[k], P F^T[k]
template <int SIDE, class FM, class FV, class FC, class FP>
__device__ __forceinline__ void scatter_node_task(const MpmDev &mp, int node, int ch, const int (&nc)[3], const float (&d0)[3], FM fm, FV fv, FC fC,
                                                  FP fP, const int *nbrBlk, float *gridB, int *status) {
  constexpr int NC = SIDE * SIDE * SIDE;
  const int sel[3] = {node / 9, (node / 3) % 3, node % 3};
  float Wt = 1.f, xi[3];
  int g[3], code = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float u = sel[d] == 0 ? 1.5f - d0[d] : (sel[d] == 1 ? d0[d] - 1.0f : d0[d] - 0.5f);
    Wt *= sel[d] == 1 ? 0.75f - u * u : 0.5f * u * u;
    xi[d] = (float)sel[d] * mp.dx - d0[d] * mp.dx;
    g[d] = nc[d] + sel[d];
    code = code * 3 + (g[d] < 0 ? 0 : (g[d] >= SIDE ? 2 : 1));
  }
  float val;
  if (ch == 0) {
    val = fm() * Wt;
  } else if (ch < 4) {
    const int d = ch - 1;
    val = Wt * fm() * (fv(d) + (fC(d) * xi[0] + fC(3 + d) * xi[1] + fC(6 + d) * xi[2]));
  } else {
    const int d = ch - 4;
    const float dxi = mp.dxi;
    const float kscale = mp.fscale;
    val = (fP(d) * kscale * xi[0] + fP(3 + d) * kscale * xi[1] + fP(6 + d) * kscale * xi[2]) * Wt;
  }
  const int bn = nbrBlk[code];
  if (bn >= 0) {
    const int cell = ((g[0] & (SIDE - 1)) * SIDE + (g[1] & (SIDE - 1))) * SIDE + (g[2] & (SIDE - 1));
    if (val != 0.f) unsafeAtomicAdd(gridB + ((size_t)bn * 7 + ch) * NC + cell, val);
  } else if (ch == 0) {
    status[2] = 1;  // mass for a node whose block is not in the partition
  }
}

// After the loop (rare: more than SL_XQ such movers in one chunk): the outbox records rec[0 .. n) that ask for it -> grid.  Record j
// belongs to wave j mod 8; lane task t = pass * 64 + lane < 189 = (stencil node t mod 27, grid channel t / 27).
template <int SIDE, class GEO>
__device__ __forceinline__ void outbox_scatter_global(const MpmDev &mp, const GEO &geo, const float *recs, int n, int w, int lane,
                                                      const int *nbrBlk, float *gridB, int *status) {
  const float dxi = mp.dxi;
#pragma unroll 1
  for (int q = 0;; ++q) {
    const int j = w + 8 * (q / 3), p = q % 3;
    if (j >= n) break;
    const int t = p * 64 + lane;
    const float *rc = recs + (size_t)j * SL_REC;
    if (t >= 189 || (reinterpret_cast<const unsigned *>(rc)[SLR_FLAG] & 1u) == 0u) continue;
    int nc[3];
    float d0[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float X = rc[1 + d] * dxi;
      const float fl = floorf(X - 0.5f);
      d0[d] = X - fl;
      nc[d] = (int)fl - geo.org[d] + geo.o[d];
    }
    scatter_node_task<SIDE>(mp, t % 27, t / 27, nc, d0, [&]() { return rc[0]; }, [&](int d) { return rc[SLR_V + d]; },
                            [&](int k) { return rc[SLR_C + k]; }, [&](int k) { return rc[SLR_PF + k]; }, nbrBlk, gridB, status);
  }
}

// Movers whose new cell is not a lane of this bin (or whose cell's arrival queue was full), and stayers whose local position rounded
// onto 1.5: the channels of set CS of their 27 node terms straight to the grid.  Two list entries per pass: lane = (entry parity,
// stencil node); the channels of the set are a compile-time loop, so only the node's weight formula (alpha + beta (s d0 + t)^2 per
// axis) and its offset from the centre node are per-lane constants.  Staged record: stage_qform.
template <int SIDE, int CS, class GEO>
__device__ __forceinline__ void slot_xlist_scatter(const MpmDev &mp, const GEO &geo, const float *stage, const unsigned *xq, int nx, int lane,
                                                   const int *nbrBlk, const SlotArgs &A) {
  using S = ConsumerSet<CS>;
  constexpr int NC = SIDE * SIDE * SIDE;
  const int node = lane & 31, half = lane >> 5;
  if (node >= 27 || nx <= 0) return;
  const int sel[3] = {node / 9, (node / 3) % 3, node % 3};
  float ws[3], wt[3], wa[3], wb[3], oc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    ws[q] = sel[q] == 0 ? -1.f : 1.f;
    wt[q] = sel[q] == 0 ? 1.5f : (sel[q] == 1 ? -1.f : -0.5f);
    wa[q] = sel[q] == 1 ? 0.75f : 0.f;
    wb[q] = sel[q] == 1 ? -1.f : 0.5f;
    oc[q] = (float)(sel[q] - 1);
  }
#pragma unroll 1
  for (int k = half; k < nx; k += 2) {
    const unsigned e = xq[k];
    const float *st = stage + (size_t)((e & 1023u) >> 6) * (G2P2G_QF * 64) + (e & 63u);
    float Wt = 1.f;
    int g[3], code = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float d0 = st[(1 + q) * 64];
      const float u = fmaf(ws[q], d0 - floorf(d0 - 0.5f), wt[q]);  // the reference's second base_node (see `edge` in the producer)
      Wt *= fmaf(wb[q], u * u, wa[q]);
      g[q] = (int)((e >> (10 + 3 * q)) & 7u) - 1 + geo.o[q] + sel[q];
      code = code * 3 + 1 + (g[q] >= SIDE ? 1 : 0) - (g[q] < 0 ? 1 : 0);
    }
    const int bn = nbrBlk[code];
    if (bn >= 0) {
      const int cell = ((g[0] & (SIDE - 1)) * SIDE + (g[1] & (SIDE - 1))) * SIDE + (g[2] & (SIDE - 1));
      float *gp = A.gridB + ((size_t)bn * 7 + S::CH0) * NC + cell;
#pragma unroll
      for (int q = 0; q < S::NA; ++q) {
        float val;
        if (S::MASS && q == 0) {
          val = Wt * st[0];  // mass
        } else {
          const float *c = st + (4 + 4 * ((S::STRESS ? 3 : 0) + S::D0 + q - (S::MASS ? 1 : 0))) * 64;
          val = Wt * fmaf(c[192], oc[2], fmaf(c[128], oc[1], fmaf(c[64], oc[0], c[0])));
        }
        if (val != 0.f) unsafeAtomicAdd(gp + q * NC, val);
      }
    } else if (S::MASS) {
      A.status[2] = 1;  // mass for a node whose block is not in the partition
    }
  }
}


// Per-bin view of the per-particle code: where the bin's particles, tickets and counters live (LDS pointers), which velocity arena its
// cells read.  VA = layout of that arena (ArenaLds: the bin's own 6^3 nodes; ArenaBlk: the 10^3 nodes of the bin's 8^3 block).
struct SlotBinView {
  int bin;              // global bin number (outbox, row base)
  int org[3];           // origin of the bin in world cells
  size_t rowBase;       // bin * K: first storage row (tile) of the bin
  unsigned kmask;       // the K valid round bits
  const float *va;      // velocity arena at the bin's cell (0, 0, 0)
  unsigned *mask0;      // [64] occupancy of the bin's cells at the start of the step
  unsigned *clr;        // [64] rounds vacated in this step
  unsigned *arrLocal;   // [64] in-bin arrivals of the cell in this step (ticket counter)
  const int *nbrBin;    // [27] the bins around this one (direction code (dx + 1) 9 + (dy + 1) 3 + dz + 1)
  int *outCount, *sent, *homed, *xOver;
};

// One occupied slot of a bin: G2P from the velocity arena, advection, F update, constitutive update, the particle's new state to its
// (old or new) slot or to an outbox record, {m, d0, Q-form coefficients} staged at `myStage` (staged position `spos` = ring group * 64 +
// lane) after beforeStage() -- the hook in which the block kernel waits until the consumers have released the ring slot.  code0 = round * 64 + cell of the slot, i0 = its element index.  Returns whether the lane of the slot's cell consumes the staged
// record (a stayer); movers are queued for the lane of their new cell (arrCnt / arrQ) or for the global-atomic list (xCnt / xq).
template <int SIDE, int SMODEL, bool WRITE_ALL, class VA, class REC, class WAIT>
__device__ __forceinline__ bool slot_produce_entry(const MpmDev &mp, const ParticlesDev &ps, const REC &cur, unsigned code0, size_t i0, int lane,
                                                   unsigned spos, float *myStage, const SlotBinView &bv, const SlotArgs &A, unsigned *arrCnt,
                                                   unsigned short (*arrQ)[SL_ARRQ], unsigned *xCnt, unsigned *xq, WAIT beforeStage,
                                                   unsigned long long *seg = nullptr) {  // seg: probe builds only (cycles per segment)
  constexpr int LW = 64;
  SLP_SEG0();
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool FLUID = model_is_fluid(SMODEL);
  const float dxi = mp.dxi;
  const float D_inv = mp.D_inv;
  bool valid = false;
  const int cell = (int)(code0 & 63u), r = (int)(code0 >> 6);
  const int cx = cell >> 4, cy = (cell >> 2) & 3, cz = cell & 3;
  Arena ar;
  make_arena(mp.dx, mp.dxi, cur.pos, ar);
  const int ocx = ar.corner[0] - bv.org[0], ocy = ar.corner[1] - bv.org[1], ocz = ar.corner[2] - bv.org[2];
  if (ocx != cx || ocy != cy || ocz != cz) {
    A.status[4] = 1;  // the storage invariant is broken (the caller moved particles without re-slotting them)
  } else {
    float vel[3], C[9];
    g2p_gather_lds<VA>(mp, ar, bv.va + VA::at(ocx, ocy, ocz), D_inv, vel, C);
    SLP_SEG(0);  // arena set-up + gather
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
      nc[d] = (int)fl - bv.org[d];
      lpn[d] = X - fl;
    }
    SLP_SEG(1);  // advection, F update, new base node
    const float pm = cur.m;
    float plj = 0.f;
    if constexpr (DP) plj = cur.logJp;
    const bool moved = nc[0] != cx || nc[1] != cy || nc[2] != cz;
    // X - floor(X - 0.5) rounded up to 1.5, or X - 0.5 rounded up to an integer and left it below 0.5 (|X| < 1 only): the reference
    // applies base_node to the local position once more and takes the weights of d0 -+ 1 on the unchanged corner
    // (InterpolationKernel.hpp:108 on simulation/Utils.hpp:59-60; make_arena restates it).  The lane = cell consumers take the staged
    // lpn as d0; such a particle is scattered through the consumers' list instead, which folds d0 as the reference does.
    const bool edge = !(lpn[0] >= 0.5f && lpn[0] < 1.5f && lpn[1] >= 0.5f && lpn[1] < 1.5f && lpn[2] >= 0.5f && lpn[2] < 1.5f);
        bool outbox = false;   // it gets an outbox record (new cell in a neighbour bin: slot_rehome_kernel finds its slot; or fallback scatter)
    bool staged = !moved;  // {m, x', v', C', P F^T} staged for the consumers
    bool home = false;     // mover with a new slot inside this bin
    bool byList = false;   // stayer scattered by the consumers' list (see `edge`)
    bool lowered = false;  // stayer re-homed into a lower round of its own cell
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
        const int rr = nth_low_bit(~bv.mask0[dl] & bv.kmask, atomicAdd(&bv.arrLocal[dl], 1u));
        if (rr >= 0) {
          home = true;
          o = particle_offset<LW>(ps.pos.chns, (bv.rowBase + (size_t)rr) * 64 + (size_t)dl);
        } else {
          A.status[1] = 1;  // cell full: the particle is scattered but has no new slot
          keep = true;
        }
        const unsigned q = edge ? (unsigned)SL_ARRQ : atomicAdd(&arrCnt[dl], 1u);
        if (q < (unsigned)SL_ARRQ) {  // the lane of the new cell scatters it (arrival queue of the chunk)
          viaX = false;
          staged = true;
          arrQ[dl][q] = (unsigned short)spos;
        }
      } else {
        const int dbin = bv.nbrBin[code];
        if (dbin >= 0) {
          outbox = true;
          dcell = (unsigned)dbin * 64u + (unsigned)dl;
        } else {
          A.status[2] = 1;  // the destination block is not in the partition: the particle keeps its old slot with its new state
          keep = true;      // (never dropped: G2P.hpp:67-82 writes every particle back); the caller re-partitions and re-slots
        }
      }
      if (viaX) {
        const unsigned k = atomicAdd(xCnt, 1u);
        if (k < (unsigned)SL_XQ) {
          staged = true;
          xq[k] = spos | ((unsigned)(nc[0] + 1) << 10) | ((unsigned)(nc[1] + 1) << 13) |
                       ((unsigned)(nc[2] + 1) << 16);
        } else {  // list full: a full record, scattered after the loop
          outbox = true;
          recFlag = 1u;
          atomicAdd(bv.xOver, 1);
        }
      }
      if (outbox) {
        const int k = atomicAdd(bv.outCount, 1);
        if (k < A.cap) {
          rec = A.moverRec + ((size_t)bv.bin * A.cap + (size_t)k) * SL_REC;
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
          reinterpret_cast<unsigned *>(rec)[SLR_FLAG] = recFlag | (code0 << 1);  // code0 = round * 64 + cell: the slot it leaves
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
        if (home) atomicAdd(bv.homed, 1);
      }
      if (!keep) {
        atomicOr(&bv.clr[cell], 1u << r);  // its slot becomes a hole
        atomicAdd(bv.sent, 1);
      }
    } else {
      {
        // A stayer that holds the TOP round of its cell while a round below is free re-homes downwards: the rounds of a bin are set by
        // the highest occupied round of its cells, and a particle that once sat on top of a crowd keeps the bin tall long after the
        // crowd has left (200 steps of the falling column: 11.6 rounds for a fullest cell of 9.1).  It draws a ticket like an in-bin
        // arrival (whatever round the ticket gives is taken, so that tickets and stored particles stay one to one) and is still
        // consumed from the staging ring by the lane of its cell: only where it is stored changes.  One per cell and step, and only
        // while the cell has rounds to spare: the vacated round is not reusable before the step's commit, so every re-homing costs
        // the step's arrivals a free round (lowering every particle above the cell's compact height filled crowded cells).
        const unsigned m0 = bv.mask0[cell];
        if (r > 0 && (m0 >> r) == 1u && (m0 & ((1u << r) - 1u)) != ((1u << r) - 1u) && __popc(~m0 & bv.kmask) >= 8) {
          const int rr = nth_low_bit(~m0 & bv.kmask, atomicAdd(&bv.arrLocal[cell], 1u));
          if (rr >= 0) {
            o = particle_offset<LW>(ps.pos.chns, (bv.rowBase + (size_t)rr) * 64 + (size_t)cell);
            atomicOr(&bv.clr[cell], 1u << r);
            lowered = true;
          }
        }
      }
      pstore_state<LW, FLUID>(ps.F, o, F);
      pstore<LW, 3>(ps.pos, o, pos);
      if (WRITE_ALL) {
        pstore<LW, 3>(ps.vel, o, vel);
        pstore<LW, 9>(ps.C, o, C);
      }
      if (edge) {  // (see `edge`) a stayer, scattered by the list; a full list leaves it to its lane
        const unsigned k = atomicAdd(xCnt, 1u);
        if (k < (unsigned)SL_XQ) {
          byList = true;
          xq[k] = spos | ((unsigned)(nc[0] + 1) << 10) | ((unsigned)(nc[1] + 1) << 13) |
                       ((unsigned)(nc[2] + 1) << 16);
        }
      }
    }
    SLP_SEG(2);  // tickets / queues / records of the movers, the particle's stores
    {  // the plastic models may project the local copy of F (the stored / recorded F is the unprojected one, P2G.hpp:101)
      float lj = plj;
      model_stress<SMODEL>(mp.mat, lj, F, PF, C);
      SLP_SEG(3);  // constitutive update
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
        if (WRITE_ALL) {
          float S[STRESS_N];
          stress_pack(PF, S);
          pstore<LW, STRESS_N>(ps.stress, o, S);
        }
        if (moved || lowered) pstore1<LW>(ps.mass, o, pm);
      }
    }
    SLP_SEG(4);  // logJp / stress / mass stores, record tail
    if (staged) {
      // staged AFTER the constitutive update, as in g2p2g_rs_producer: with m, x', v', C' dead before it the compiler
      // reuses their registers for the SVD at once and waits for the particle stores just issued (s_waitcnt vmcnt(1)
      // in front of the SVD: 2 ms per 64 Mi particles)
      valid = !moved && !byList;  // an in-bin mover is consumed by the lane of its NEW cell (arrival queue), not by the lane of its entry
      beforeStage();
      SLP_SEG(5);  // wait for the ring slot
      stage_qform(mp, myStage, pm, lpn, vel, C, PF);
      SLP_SEG(6);  // staging
    }
  }
  return valid;
}

// producer wave W (0..3): entries [64 (4c + W), +64) of every chunk c
template <int SIDE, int SMODEL, bool WRITE_ALL, int W>
__device__ __forceinline__ void g2p2g_slot_producer(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, int bin, int total,
                                                    int lane, int nchunks, const SlotShared &sh, const SlotArgs &A) {
  using AL = ArenaLds;
  constexpr int LW = 64;
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool FLUID = model_is_fluid(SMODEL);
  constexpr int NC = SIDE * SIDE * SIDE;
  const float dxi = mp.dxi;
  const float D_inv = mp.D_inv;
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
  const SlotBinView bv{bin, {geo.org[0], geo.org[1], geo.org[2]}, rowBase, kmask, sh.varena, sh.mask0, sh.clr, sh.arrLocal, sh.nbrBin,
                       sh.outCount, sh.sent, sh.homed, sh.xOver};
  RecG<LW, DP, FLUID> cur, nxt;
  bool has0 = false, has1 = false;
  size_t i0 = 0, i1 = 0;
  unsigned code0 = 0, code1 = 0;
  // The record of a chunk is requested one chunk ahead and handed over (nxt -> cur) at the TOP of the iteration that uses it: a `cur` that
  // is loaded directly on the entry path makes the compiler place its in-order vmcnt waits at the first uses inside the loop, where they
  // also drain the prefetch issued a few hundred instructions earlier (seen in the ISA: s_waitcnt vmcnt(7..0) in front of the F update)
  if (nchunks > 0) {
    const int j = 64 * W + lane;
    has1 = j < total;
    if (has1) {
      code1 = tab[j];
      i1 = (rowBase + (size_t)(code1 >> 6)) * 64 + (size_t)(code1 & 63u);
      nxt.load(ps, i1);
    }
  }
  {
    const int tid = (int)threadIdx.x;  // the four producer waves are threads 0..255
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
  }
  SLP_T0(tFill);
  __syncthreads();
  if (W == 0) SLP_ADD(2, tFill);
#ifdef ZS_SLOT_PROBE
  unsigned long long tWork = 0, tBar = 0;
#endif
  for (int it = 0; it < nchunks; ++it) {
    SLP_T0(tIt);
    {
      const int grp = 4 * it + W;
      const int par = it % 3;
      float *myStage = stage + (size_t)(grp % SL_NG) * (G2P2G_QF * 64);
      cur = nxt;
      has0 = has1;
      i0 = i1;
      code0 = code1;
      has1 = false;
      if (it + 1 < nchunks) {
        const int j1 = 64 * (grp + 4) + lane;
        has1 = j1 < total;
        if (has1) {
          code1 = tab[j1];
          i1 = (rowBase + (size_t)(code1 >> 6)) * 64 + (size_t)(code1 & 63u);
          nxt.load(ps, i1);  // in flight during this chunk
        }
      }
      bool valid = false;
      if (has0)
        valid = slot_produce_entry<SIDE, SMODEL, WRITE_ALL, ArenaLds>(mp, ps, cur, code0, i0, lane, (unsigned)((grp % SL_NG) * 64 + lane), myStage + lane, bv, A,
                                                                      arrCnt[par], arrQ[par], &xCnt[par], xq[par], [] {});
      {
        const unsigned long long vm = __ballot(valid);
        if (lane == 0) smask[grp % SL_NG] = vm;
      }
    }
    SLP_ACC(tWork, tIt);
    SLP_T0(tB);
    __syncthreads();
    SLP_ACC(tBar, tB);
  }
  {  // nothing left to produce (the consumers accumulate the rounds of the last chunk): the last chunk's global-atomic list
    SLP_T0(tB);
    if (ZS_PROD_XLIST && nchunks > 0) {
      const int par = (nchunks - 1) % 3;
      const int nx = xCnt[par] < (unsigned)SL_XQ ? (int)xCnt[par] : SL_XQ;
      slot_xlist_scatter<SIDE, W>(mp, geo, stage, xq[par], nx, lane, sh.nbrBlk, A);
    }
    __syncthreads();
    if (W == 0) SLP_ADD(6, tB);
  }
  if (W == 0) {
    SLP_PUT(3, tWork);
    SLP_PUT(4, tBar);
  }
}
// consumer wave of channel set CS: lane = cell; after chunk c has been produced every round whose last entry lies below 256 (c + 1)
// is complete and is consumed while the producers work on chunk c + 1 -- together with the in-bin arrivals queued during chunk c
template <int SIDE, int CS>
__device__ __forceinline__ void g2p2g_slot_consumer(const MpmDev &mp, const BinGeom<SIDE> &geo, unsigned mask, int total, int lane, int nchunks,
                                                    const SlotShared &sh, const SlotArgs &A) {
  using S = ConsumerSet<CS>;
  using AL = ArenaLds;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const unsigned long long lt = lanemask_lt();
  const float *const stage = sh.stage;
  const unsigned long long *const smask = sh.smask;
  unsigned(*const arrCnt)[64] = sh.arrCnt;
  const unsigned short(*const arrQ)[64][SL_ARRQ] = sh.arrQ;
  float *const parena = sh.parena;
  unsigned *const xCnt = sh.xCnt;
  const unsigned(*const xq)[SL_XQ] = sh.xq;
  const int *const nbrBlk = sh.nbrBlk;
  float acc[27][S::NA];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int q = 0; q < S::NA; ++q) acc[k][q] = 0.f;
  __syncthreads();  // (the producers fill the velocity arena meanwhile)
#ifdef ZS_SLOT_PROBE
  unsigned long long tWork = 0, tBar = 0;
#endif
  int r = 0, off = 0;  // next round to consume, entry number of its first particle
  for (int it = 0; it <= nchunks; ++it) {
    SLP_T0(tIt);
    if (it > 0) {
      const int par = (it - 1) % 3;
      const int produced = 256 * it < total ? 256 * it : total;
      const unsigned qn = arrCnt[par][lane];
      const int na = qn < (unsigned)SL_ARRQ ? (int)qn : SL_ARRQ;
      int ai = 0;
      if (CS == 0) arrCnt[(it + 1) % 3][lane] = 0u;  // the counters the NEXT chunk will use (last read one iteration ago)
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
            if ((smask[grp] >> pos) & 1ull) spos = grp * (G2P2G_QF * 64) + pos;
          }
          off += cnt;
          ++r;
        }
        if (spos < 0 && pend) {  // a lane without a particle of its own in this round takes an arrival
          const unsigned p = arrQ[par][lane][ai++];
          spos = (int)(p >> 6) * (G2P2G_QF * 64) + (int)(p & 63u);
        }
        if (spos >= 0) g2p2g_consume_set<CS>(mp, stage, spos, acc);
        if (CS == 0) SLP_PUT(15, 1);  // [15] consumer: loop iterations (rounds + extra rounds for arrivals)
      }
      if (CS == 0) SLP_ADD(13, tIt);  // [13] consumer: rounds loop (incl. in-bin arrivals)
      SLP_T0(tX);
      // movers of the chunk whose new cell is not a lane of this bin (or whose cell's arrival queue was full): slot_xlist_scatter.  The
      // list of the LAST chunk is taken by the producer waves, which have nothing left to produce in that iteration
      const int nx = xCnt[par] < (unsigned)SL_XQ ? (int)xCnt[par] : SL_XQ;
      if (CS == 0 && lane == 0) xCnt[(it + 1) % 3] = 0u;
      if (!ZS_PROD_XLIST || it < nchunks) slot_xlist_scatter<SIDE, CS>(mp, geo, stage, xq[par], nx, lane, nbrBlk, A);
      if (CS == 0) {
        SLP_ADD(14, tX);  // [14] consumer: global-atomic list
        SLP_PUT(12, nx);  // [12] entries of the list
      }
    }
    SLP_ACC(tWork, tIt);
    SLP_T0(tB);
    __syncthreads();
    SLP_ACC(tBar, tB);
  }
  if (CS == 0) {
    SLP_PUT(7, tWork);
    SLP_PUT(8, tBar);
  }
  SLP_T0(tFl);
  // the set's channels of the bin's arena belong to this wave alone; phases ordered inside the wave (see g2p2g_body)
  // (the arena lives in the staging ring, which nobody reads after the loop's last barrier: cleared here, by the wave that owns the channels)
  for (int k = lane; k < S::NA * AL::CH; k += 64) parena[(size_t)S::CH0 * AL::CH + k] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  float *a0 = parena + (size_t)S::CH0 * AL::CH + AL::at(cx, cy, cz);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
    for (int q = 0; q < S::NA; ++q) g[q * AL::CH] += acc[k][q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (CS == 0) SLP_ADD(9, tFl);
}

// mpm_slotblk.hip: the step's main kernel for 8^3 blocks, one workgroup per block (blocks [A.binBase / 8, + A.nbins / 8))
void launch_g2p2g_slotblk(hipStream_t stream, int model, bool writeAll, const MpmDev &mp, const ParticlesDev &pd, const BhtDev &t, const SlotArgs &A);

}  // namespace zsr
