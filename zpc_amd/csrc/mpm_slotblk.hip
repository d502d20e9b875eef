// mpm_slotblk.hip -- the fused G2P2G step on slotted storage with ONE WORKGROUP PER 8^3 GRID BLOCK (r05).
//
// Why.  g2p2g_slot_kernel (mpm_slotted.hip) gives every bin (4^3 cells, ~512 particles = 1.9 chunks of 256) a workgroup of its own: a
// producer -> consumer pipeline that is filled and drained once per bin, in front of it a head (occupancy words, entry table, velocity
// arena: three dependent memory round trips) and behind it a tail (arena flush, grid atomics).  The cycle stamps of that kernel
// (profiles/r05_slot_probe.md) put head + drain + flush + tail at half of a workgroup's life; the two workgroups of a CU overlap them only by
// chance.  Here the 8 bins of a block are one pipeline:
//   * the chunk sequence runs through all bins of the block (bin 0 chunk 0, bin 0 chunk 1, bin 1 chunk 0, ...): producers prefetch the
//     records of chunk g + 1 -- whichever bin it belongs to -- while they compute chunk g; the consumers flush a bin's accumulators and
//     send its arena to the grid while the producers are already in the next bin.  Fill and drain are paid once per block;
//   * head once per block: the occupancy words of all 8 bins in one load, ONE velocity arena of 10^3 nodes for the whole block
//     (12 KB; the bins' own 6^3 arenas overlap by half);
//   * the staging ring holds ONE chunk (+ the group a round may straddle): 5 groups of 64 records instead of 9 (35 KB, which is what pays
//     for the block arena).  A producer waits, right before it stages, until all four consumers have counted the previous chunk done (an
//     LDS counter, no barrier: the consumers finish a chunk in half the time the producers need for the next); one barrier per chunk
//     hands the staged chunk over;
//   * per-bin state (entry table, tickets, departures, outbox count, neighbour bins) is double-buffered by bin parity and kept by the
//     CONSUMER waves, which have the time: the entry table of a bin is built three chunks ahead, its neighbour bins two chunks ahead,
//     a finished bin's claim words are written out with its last chunk (blk_consumer spells out when each buffer is free); everything
//     the loops need to know about a chunk sits in one packed word (ChunkDesc), read once per iteration into SGPRs;
//   * 128 VGPRs and NO scratch: the uniform constants of the per-particle code come from the host (MpmDev::dxi, D_inv, fscale;
//     Material::smu, dpCoef, expCohesion -- kernel arguments live in SGPRs), one producer body serves the four producer waves.  On gfx9
//     loads and stores share vmcnt, so a spill reload inside the chunk loop waits for the record prefetch of the next chunk and for the
//     particle stores just issued (profiles/r05_block_kernel.md, section 7; tests/test_host_cpu.py keeps it that way);
//   * the bin's P2G arena has 8^3 nodes (ArenaBin8): a mover into a neighbour bin adds its 27 node terms there (plain LDS
//     read-add-write) and they reach the grid with the bin's one flush -- no global atomics per mover.
// The per-particle code (slot_produce_entry), the consumers' accumulation (g2p2g_consume_set), the mover protocol, slot_rehome_kernel
// and slot_commit_kernel are those of mpm_slotted.hip; results differ only in summation order.
#include "mpm_slot.hpp"

namespace zsr {

constexpr int SB_NG = 5;       // ring: the four groups of a chunk + the group a straddling round keeps alive
constexpr int SB_MAXCH = 64;   // chunks per block: 8 bins x (32 rounds x 64 cells / 256)

struct SubGeom {  // what the shared scatter helpers need of a bin: origin inside the block, origin in world cells
  int o[3], org[3];
};
__device__ __forceinline__ SubGeom sub_geom(const int (&borg)[3], int b) {
  SubGeom g;
  g.o[0] = ((b >> 2) & 1) * 4;
  g.o[1] = ((b >> 1) & 1) * 4;
  g.o[2] = (b & 1) * 4;
#pragma unroll
  for (int d = 0; d < 3; ++d) g.org[d] = borg[d] + g.o[d];
  return g;
}

struct BlkShared {
  float *varena, *parena, *stage;
  unsigned long long *smask;
  unsigned short (*tab)[SL_KMAX * 64];       // [2]
  unsigned (*masks)[64];                     // [8]
  unsigned (*clr)[64], (*arrLocal)[64];      // [2]
  unsigned (*arrCnt)[64];                    // [3]
  unsigned short (*arrQ)[64][SL_ARRQ];       // [3]
  unsigned *xCnt;                            // [3]
  unsigned (*xq)[SL_XQ];                     // [3]
  int *nbrBlk, (*nbrBin)[27], *nbr8;
  int (*cnt)[4];                             // [2]: outCount, sent, homed, xOver of the bin of that parity
  int *total, *gbase, *nch, *binQ, *oc;      // [8] per bin: occupied slots, first chunk number, chunks, position among the block's non-empty bins,
                                             // outbox records (after the bin has been finished)
  unsigned char *chBin, *chIdx;              // [SB_MAXCH] chunk -> bin of the block, chunk number inside the bin
  const unsigned *desc;                      // [SB_MAXCH + 5] desc[g + 1] = packed descriptor of chunk g (ChunkDesc), zero outside [0, G)
  unsigned *done;                            // consumer waves x chunks consumed
  int *sums;                                 // [0] sent, [1] homed, [2] bins whose outbox holds records that still have to be scattered
};

// Everything the chunk loop needs to know about a chunk, in one word: the loops read desc[g .. g + 3] (chunks g - 1 .. g + 2) in ONE LDS access
// and keep the fields in SGPRs.  (Read field by field -- chBin[g], then binQ / total / gbase / nch of that bin, for three chunks -- the top of
// a producer iteration was ~20 dependent LDS round trips behind the consumers' traffic: 4.1 k of its 17 k cycles, r05 stamps.)
struct ChunkDesc {
  unsigned w;
  __device__ static unsigned pack(int b, int c, int qp, bool last, int total, int gbase) {
    return (unsigned)b | ((unsigned)c << 3) | ((unsigned)(qp & 1) << 7) | ((unsigned)last << 8) | ((unsigned)total << 9) | ((unsigned)gbase << 21) | (1u << 31);
  }
  __device__ int bin() const { return (int)(w & 7u); }             // bin of the block
  __device__ int idx() const { return (int)((w >> 3) & 15u); }     // chunk number inside the bin
  __device__ int qp() const { return (int)((w >> 7) & 1u); }       // parity of the bin among the block's non-empty bins
  __device__ bool last() const { return (w >> 8) & 1u; }           // last chunk of its bin
  __device__ int total() const { return (int)((w >> 9) & 4095u); } // occupied slots of the bin
  __device__ int gbase() const { return (int)((w >> 21) & 127u); } // number of the bin's first chunk
};
template <int N> __device__ __forceinline__ void blk_chunk_descs(const unsigned *desc, int g, ChunkDesc (&d)[N]) {  // d[k] = chunk g - 1 + k
#pragma unroll
  for (int k = 0; k < N; ++k) d[k].w = desc[g + k];
#pragma unroll
  for (int k = 0; k < N; ++k) d[k].w = (unsigned)__builtin_amdgcn_readfirstlane((int)d[k].w);
}

// entry table of one bin: round-major enumeration of its occupied slots; wave w of nw writes the rows of rounds = w mod nw
__device__ __forceinline__ void blk_build_tab(unsigned mask, int lane, int w, int nw, unsigned short *tab) {
  const unsigned long long lt = lanemask_lt();
  int total = 0;
#pragma unroll 1
  for (int r = 0; r < SL_KMAX; ++r) {  // (ballots only: no cross-lane data movement, nothing that waits for the record loads in flight)
    if (__ballot((mask >> r) != 0u) == 0ull) break;
    const bool has = (mask >> r) & 1u;
    const unsigned long long occ = __ballot(has);
    if ((r % nw) == w && has) tab[total + __popcll(occ & lt)] = (unsigned short)(r * 64 + lane);
    total += __popcll(occ);
  }
}
// the 27 bins around bin b of this block (direction code (dx + 1) 9 + (dy + 1) 3 + dz + 1), from the block's 27 neighbour blocks
__device__ __forceinline__ int blk_neighbour_bin(const int *nbrBlk, int blk, int b, int code) {
  if (code == 13) return blk * 8 + b;
  const int dd[3] = {code / 9 - 1, (code / 3) % 3 - 1, code % 3 - 1};
  int sx[3] = {((b >> 2) & 1) + dd[0], ((b >> 1) & 1) + dd[1], (b & 1) + dd[2]}, bo[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    bo[d] = sx[d] < 0 ? -1 : (sx[d] > 1 ? 1 : 0);
    sx[d] &= 1;
  }
  const int nb = nbrBlk[(bo[0] + 1) * 9 + (bo[1] + 1) * 3 + (bo[2] + 1)];
  return nb < 0 ? -1 : nb * 8 + ((sx[0] * 2 + sx[1]) * 2 + sx[2]);
}

// a finished bin (all of its chunks produced by all producer waves): departures and in-bin arrivals of its cells for slot_rehome_kernel /
// slot_commit_kernel, its outbox count; the parity's counters are zero again.  One wave, lane = cell; every LDS read is issued before the
// first use (ONE round trip: as a chain of lane-0 reads this was 4.9 k cycles per bin on the producers' critical path, r05 stamps).
__device__ __forceinline__ void blk_finish_bin(const BlkShared &sh, const SlotArgs &A, int bin0, int b, int qp, int lane) {
  const int bin = bin0 + b;
  const unsigned c = sh.clr[qp][lane], nl = sh.arrLocal[qp][lane];
  const int cv = sh.cnt[qp][lane & 3];  // outCount, sent, homed, xOver
  if (c) A.claim[((size_t)A.nbinsAll + (size_t)bin) * 64 + lane] = c;
  if (nl) A.claim[(size_t)bin * 64 + lane] = nl << 16;  // (the low half -- arrivals from other bins -- is counted by slot_rehome_kernel)
  sh.clr[qp][lane] = 0u;
  sh.arrLocal[qp][lane] = 0u;
  const int c0 = __builtin_amdgcn_readlane(cv, 0), c1 = __builtin_amdgcn_readlane(cv, 1), c2 = __builtin_amdgcn_readlane(cv, 2),
            c3 = __builtin_amdgcn_readlane(cv, 3);
  if (lane == 0) {
    const int oc = c0 < A.cap ? c0 : A.cap;
    A.moverCount[bin] = oc;
    sh.oc[b] = oc;
  }
  if (lane < 3) {  // sums: sent, homed, bins whose outbox holds records that still have to be scattered (one bit per bin: + is |)
    const int add = lane == 0 ? c1 : (lane == 1 ? c2 : (c3 > 0 ? 1 << b : 0));
    if (add) atomicAdd(&sh.sums[lane], add);
  }
  if (lane < 4) sh.cnt[qp][lane] = 0;
}

// P2G arena of ONE bin in the block kernel: 8^3 nodes = the bin's 6^3 stencil nodes + one layer around them, origin at the bin's node -1.
// A mover whose new cell lies in a neighbour bin (base node -1 .. 4 per axis) has its 27 nodes inside, so its grid terms are added HERE
// (plain LDS read-add-write) and reach the grid with the bin's one flush -- instead of 189 global float atomics per mover (0.4 ms and
// 1.8 GB of write-through sectors per step of the 64 Mi-particle column).  Strides: for a fixed stencil offset the 64 cells of a bin land
// on 32 distinct banks per 32-lane half (z + 8 y + 68 x: x adds 4 mod 32).
struct ArenaBin8 {
  static constexpr int W = 8;
  static constexpr int SY = 8, SX = 68, CH = W * SX;
  __device__ static constexpr int at(int x, int y, int z) { return x * SX + y * SY + z; }
};

// The chunk's list (movers into neighbour bins, arrival-queue overflow, stayers whose local position rounded onto 1.5): the channels of
// set CS of their 27 node terms into the bin's 8^3 arena `pa` (this wave's channels).  Two list entries per pass, lane = (entry parity,
// stencil node); the two entries of a pass may share nodes, so the halves do their read-add-write one after the other (LDS operations of
// a wave execute in order).  Weights and staged record as slot_xlist_scatter.
template <int CS>
__device__ __forceinline__ void blk_xlist_lds(const float *stage, const unsigned *xq, int nx, int lane, float *pa) {
  using S = ConsumerSet<CS>;
  using A8 = ArenaBin8;
  const int node = lane & 31, half = lane >> 5;
  if (nx <= 0) return;
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
  for (int k0 = 0; k0 < nx; k0 += 2) {
    const int k = k0 + half;
    const bool act = node < 27 && k < nx;
    float val[S::NA];
    int an = 0;
    if (act) {
      const unsigned e = xq[k];
      const float *st = stage + (size_t)((e & 1023u) >> 6) * (G2P2G_QF * 64) + (e & 63u);
      float Wt = 1.f;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float d0 = st[(1 + q) * 64];
        const float u = fmaf(ws[q], d0 - floorf(d0 - 0.5f), wt[q]);  // the reference's second base_node (see `edge` in the producer)
        Wt *= fmaf(wb[q], u * u, wa[q]);
      }
      // (the entry carries new cell + 1 per axis = the arena coordinate of the stencil's node 0)
      an = A8::at((int)((e >> 10) & 7u) + sel[0], (int)((e >> 13) & 7u) + sel[1], (int)((e >> 16) & 7u) + sel[2]);
#pragma unroll
      for (int q = 0; q < S::NA; ++q) {
        if (S::MASS && q == 0) {
          val[q] = Wt * st[0];  // mass
        } else {
          const float *c = st + (4 + 4 * ((S::STRESS ? 3 : 0) + S::D0 + q - (S::MASS ? 1 : 0))) * 64;
          val[q] = Wt * fmaf(c[192], oc[2], fmaf(c[128], oc[1], fmaf(c[64], oc[0], c[0])));
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (act && half == h) {
#pragma unroll
        for (int q = 0; q < S::NA; ++q) pa[q * A8::CH + an] += val[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

// producer wave W (0..3): entries [64 (4 c + W), +64) of every chunk c of every bin of the block.  W is a wave-uniform RUNTIME value: one
// copy of the 17 KB producer body for the four waves (as a template parameter the kernel carried four: 87 KB of code, 70 KB of it hot in
// every chunk -- more than the instruction cache two CUs share)
template <int SMODEL, bool WRITE_ALL>
__device__ __forceinline__ void blk_producer(const MpmDev &mp, const ParticlesDev &ps, const int (&borg)[3], int blk, int lane, int G,
                                             const BlkShared &sh, const SlotArgs &A, const int W) {
  constexpr int LW = 64;
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool FLUID = model_is_fluid(SMODEL);
  const int bin0 = blk * 8;
  const unsigned kmask = A.K >= 32 ? 0xffffffffu : ((1u << A.K) - 1u);
  RecG<LW, DP, FLUID> cur, nxt;
  bool has0 = false, has1 = false;
  unsigned code0 = 0, code1 = 0;  // round * 64 + cell of the entry (its element index is rebuilt from it where needed: two VGPRs less across the loop)
  auto elem = [&](int b, unsigned code) { return ((size_t)(bin0 + b) * (size_t)A.K + (size_t)(code >> 6)) * 64 + (size_t)(code & 63u); };
  {
    ChunkDesc d[2];
    blk_chunk_descs(sh.desc, 0, d);
    const int b = d[1].bin();
    const int j = 64 * W + lane;
    has1 = j < d[1].total();
    if (has1) {
      code1 = sh.tab[d[1].qp()][j];
      nxt.load(ps, elem(b, code1));
    }
  }
  __syncthreads();  // every producer has read the first bin's entry table (iteration 0 may rebuild that buffer for the bin of chunk 2)
#ifdef ZS_SLOT_PROBE
  unsigned long long tWork = 0, tBar = 0, tRing = 0, tPre = 0, segv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long *const seg = segv;
  unsigned long long topv[4] = {0, 0, 0, 0}, tTop = 0;
#define SLP_TOP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); topv[k] += tn_ - tTop; tTop = tn_; } while (0)
#else
#define SLP_TOP(k) do { } while (0)
  unsigned long long *const seg = nullptr;
#endif
  for (int g = 0; g < G; ++g) {
    SLP_T0(tIt);
#ifdef ZS_SLOT_PROBE
    tTop = tIt;
#endif
    ChunkDesc d[3];  // chunks g - 1, g, g + 1
    blk_chunk_descs(sh.desc, g, d);
    const int b = d[1].bin();
    const int qp = d[1].qp();
    const int grp = 4 * g + W, slot = grp % SB_NG, par = g % 3;
    float *myStage = sh.stage + (size_t)slot * (G2P2G_QF * 64);
    cur = nxt;
    has0 = has1;
    code0 = code1;
    has1 = false;
    SLP_TOP(0);  // descriptors, hand-over
    if (g + 1 < G) {  // the records of the next chunk (this bin's or the next bin's): in flight during this chunk
      const int b1 = d[2].bin();
      const int j1 = 256 * d[2].idx() + 64 * W + lane;
      has1 = j1 < d[2].total();
      if (has1) {
        code1 = sh.tab[d[2].qp()][j1];
        nxt.load(ps, elem(b1, code1));
      }
    }
    SLP_TOP(1);  // record requests
    // (per-bin state around the bin boundaries -- a finished bin's counters, the next bins' entry tables and neighbour bins -- is kept by the
    // consumer waves, which have the time: see blk_consumer)
    const SubGeom sg = sub_geom(borg, b);
    const SlotBinView bv{bin0 + b, {sg.org[0], sg.org[1], sg.org[2]}, (size_t)(bin0 + b) * (size_t)A.K, kmask,
                         sh.varena + ArenaBlk::at(sg.o[0], sg.o[1], sg.o[2]), sh.masks[b], sh.clr[qp], sh.arrLocal[qp], sh.nbrBin[qp],
                         &sh.cnt[qp][0], &sh.cnt[qp][1], &sh.cnt[qp][2], &sh.cnt[qp][3]};
    // the ring slot this wave stages into was last read by the consumers of chunk g - 1 (its groups 0..2, or the straddle group of
    // chunk g - 2, consumed with chunk g - 1): all four consumer waves have counted that chunk done
    auto ringFree = [&] {
      while (__hip_atomic_load(sh.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * (unsigned)g) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    bool valid = false;
    SLP_ACC(tPre, tIt);
    if (has0)
      valid = slot_produce_entry<8, SMODEL, WRITE_ALL, ArenaBlk>(mp, ps, cur, code0, elem(b, code0), lane, (unsigned)(slot * 64 + lane), myStage + lane, bv, A,
                                                                 sh.arrCnt[par], sh.arrQ[par], &sh.xCnt[par], sh.xq[par], ringFree, seg);
    SLP_T0(tR);
    ringFree();
    SLP_ACC(tRing, tR);
    {
      const unsigned long long vm = __ballot(valid);
      if (lane == 0) sh.smask[slot] = vm;
    }
    SLP_ACC(tWork, tIt);
#ifdef ZS_SLOT_PROBE
    {  // how long do the acknowledgements of this iteration's stores (and the next chunk's records) take from here?
      SLP_T0(tV);
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      SLP_ACC(segv[7], tV);
    }
#endif
    SLP_T0(tB);
    __syncthreads();  // chunk g is staged
    SLP_ACC(tBar, tB);
  }
  SLP_T0(tF);
  __syncthreads();  // the consumers have accumulated the last chunk and flushed the last bin
  if (W == 0) {
    SLP_ADD(6, tF);
    SLP_PUT(3, tWork);
    SLP_PUT(4, tBar);
    SLP_PUT(2, tRing);
    SLP_PUT(5, G);
#ifdef ZS_SLOT_PROBE
    SLP_PUT(16, tPre);
    for (int k = 0; k < 7; ++k) SLP_PUT(17 + k, segv[k]);
    SLP_PUT(24, segv[7]);
    for (int k = 0; k < 2; ++k) SLP_PUT(25 + k, topv[k]);
#endif
  }
}

// consumer wave of channel set CS: lane = cell of the current bin; chunk g is consumed while the producers work on chunk g + 1
template <int CS>
__device__ __forceinline__ void blk_consumer(const MpmDev &mp, const int (&borg)[3], int blk, int lane, int G, const BlkShared &sh, const SlotArgs &A) {
  using S = ConsumerSet<CS>;
  using AL = ArenaBin8;
  constexpr int NC = 512;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const unsigned long long lt = lanemask_lt();
  const float *const stage = sh.stage;
  float *const pa = sh.parena + (size_t)S::CH0 * AL::CH;  // this wave's channels of the bin's arena
  float acc[27][S::NA];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int q = 0; q < S::NA; ++q) acc[k][q] = 0.f;
  unsigned mask = 0u;
  int r = 0, off = 0;  // next round to consume, entry number of its first particle
  __syncthreads();  // (the producers' first record requests are out)
  // Per-bin state, kept by the consumers (off the producers' critical path; every buffer named here is dead for its previous owner, see the
  // file comment).  In iteration g, i.e. between the barriers "chunk g staged" and "chunk g + 1 staged", while the producers work on
  // chunk g + 1:
  //   * entry table of the bin that starts with chunk g + 3 (its first records are requested at the top of the producers' iteration g + 2;
  //     the table of the bin two bins earlier, same parity, was last read at the top of their iteration g at the latest), all four waves;
  //   * neighbour bins of the bin that starts with chunk g + 2 (used by the producers from their iteration g + 2 on), wave of set 2;
  //   * the counters of the bin that ENDS with chunk g (all of it was produced before the barrier "chunk g staged"; the parity's next
  //     owner starts with chunk g + 2 at the earliest), wave of set 3.
  // The kernel head has done the bins that start with chunks 0 and 1; the one that starts with chunk 2 is "iteration -1", here.
  auto binState = [&](int g, const ChunkDesc &c2, const ChunkDesc &c3) {  // c2 / c3: chunks g + 2 / g + 3
    if (g + 3 < G && c3.idx() == 0) blk_build_tab(sh.masks[c3.bin()][lane], lane, CS, 4, sh.tab[c3.qp()]);
    if (CS == 2 && g >= 0 && g + 2 < G && c2.idx() == 0 && lane < 27) sh.nbrBin[c2.qp()][lane] = blk_neighbour_bin(sh.nbrBlk, blk, c2.bin(), lane);
  };
  {
    ChunkDesc d[5];
    blk_chunk_descs(sh.desc, -1 + 1, d);  // chunks -1 .. 3
    binState(-1, d[2], d[3]);
  }
#ifdef ZS_SLOT_PROBE
  unsigned long long tWork = 0, tBar = 0, tFlush = 0;
#endif
  for (int g = 0; g < G; ++g) {
    SLP_T0(tB);
    __syncthreads();  // chunk g is staged
    SLP_ACC(tBar, tB);
    SLP_T0(tIt);
    ChunkDesc d[5];  // chunks g - 1 .. g + 3
    blk_chunk_descs(sh.desc, g, d);
    const int b = d[1].bin(), c = d[1].idx(), total = d[1].total(), par = g % 3;
    const int gb = 4 * d[1].gbase();  // ring group number of the bin's entry 0
    if (c == 0) {
      r = 0;
      off = 0;
      mask = sh.masks[b][lane];
      for (int k = lane; k < S::NA * AL::CH; k += 64) pa[k] = 0.f;  // the bin's arena (this wave's channels): the chunk lists add into it first
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    const int produced = 256 * (c + 1) < total ? 256 * (c + 1) : total;
    const unsigned qn = sh.arrCnt[par][lane];
    const int na = qn < (unsigned)SL_ARRQ ? (int)qn : SL_ARRQ;
    int ai = 0;
    if (CS == 0) sh.arrCnt[(g + 2) % 3][lane] = 0u;  // the counters chunk g + 2 will use (last read with chunk g - 1, which every consumer has left)
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
          const int slot = (gb + (e >> 6)) % SB_NG, pos = e & 63;
          if ((sh.smask[slot] >> pos) & 1ull) spos = slot * (G2P2G_QF * 64) + pos;
        }
        off += cnt;
        ++r;
      }
      if (spos < 0 && pend) {  // a lane without a particle of its own in this round takes an arrival
        const unsigned p = sh.arrQ[par][lane][ai++];
        spos = (int)(p >> 6) * (G2P2G_QF * 64) + (int)(p & 63u);
      }
      if (spos >= 0) g2p2g_consume_set<CS>(mp, stage, spos, acc);
    }
    {
      const int nx = sh.xCnt[par] < (unsigned)SL_XQ ? (int)sh.xCnt[par] : SL_XQ;
      if (CS == 0 && lane == 0) sh.xCnt[(g + 2) % 3] = 0u;
      blk_xlist_lds<CS>(stage, sh.xq[par], nx, lane, pa);
    }
    // this wave has read what it needs of chunk g: the producers may stage chunk g + 1 over it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(sh.done, 1u);
    binState(g, d[3], d[4]);
    if (CS == 3 && d[1].last()) blk_finish_bin(sh, A, blk * 8, b, d[1].qp(), lane);
    SLP_ACC(tWork, tIt);
    SLP_T0(tFl);
    if (d[1].last()) {
      // last chunk of the bin: the set's channels of the bin's arena belong to this wave alone -- add the 27 register planes on top of
      // the lists' terms (phases ordered inside the wave), send the arena's nodes to the grid; no other wave is involved
      float *a0 = pa + AL::at(cx + 1, cy + 1, cz + 1);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        float *gp = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
        for (int q = 0; q < S::NA; ++q) {
          gp[q * AL::CH] += acc[k][q];
          acc[k][q] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      const SubGeom sg = sub_geom(borg, b);
      for (int n = lane; n < 512; n += 64) {
        const int x = n >> 6, y = (n >> 3) & 7, z = n & 7;
        const int g[3] = {sg.o[0] - 1 + x, sg.o[1] - 1 + y, sg.o[2] - 1 + z};  // node in cells of this block: -1 .. 10
        const int code = ((g[0] < 0 ? 0 : (g[0] >= 8 ? 2 : 1)) * 3 + (g[1] < 0 ? 0 : (g[1] >= 8 ? 2 : 1))) * 3 + (g[2] < 0 ? 0 : (g[2] >= 8 ? 2 : 1));
        const int bn = sh.nbrBlk[code];
        const float *a = pa + AL::at(x, y, z);
        if (bn >= 0) {
          float *gq = A.gridB + ((size_t)bn * 7 + S::CH0) * NC + (((g[0] & 7) * 8 + (g[1] & 7)) * 8 + (g[2] & 7));
#pragma unroll
          for (int q = 0; q < S::NA; ++q) {
            const float v = a[q * AL::CH];
            if (v != 0.f) unsafeAtomicAdd(gq + q * NC, v);
          }
        } else if (S::MASS && a[0] != 0.f) {
          A.status[2] = 1;  // mass for a node whose block is not in the partition
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the arena is cleared again for the next bin only after these reads)
    }
    SLP_ACC(tFlush, tFl);
  }
  SLP_T0(tE);
  __syncthreads();  // the last bin is flushed
  if (CS == 0) {
    SLP_PUT(7, tWork);
    SLP_PUT(8, tBar);
    SLP_PUT(9, tFlush);
    SLP_ADD(10, tE);
  }
}

template <int SMODEL, bool WRITE_ALL>
static __global__ __launch_bounds__(512, 4) void g2p2g_slotblk_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, SlotArgs A) {
  constexpr int NC = 512;
  __shared__ float s_varena[3 * ArenaBlk::CH];
  __shared__ float s_parena[7 * ArenaBin8::CH];
  __shared__ float s_stage[SB_NG * G2P2G_QF * 64];
  __shared__ unsigned long long s_smask[SB_NG];
  __shared__ unsigned short s_tab[2][SL_KMAX * 64];
  __shared__ unsigned s_masks[8][64];
  __shared__ unsigned s_clr[2][64], s_arrLocal[2][64], s_arrCnt[3][64];
  __shared__ unsigned short s_arrQ[3][64][SL_ARRQ];
  __shared__ unsigned s_xCnt[3], s_xq[3][SL_XQ];
  __shared__ int s_nbrBlk[27], s_nbrBin[2][27], s_nbr8[8];
  __shared__ int s_cnt[2][4];
  __shared__ int s_total[8], s_gbase[8], s_nch[8], s_binQ[8], s_oc[8];
  __shared__ unsigned char s_chBin[SB_MAXCH], s_chIdx[SB_MAXCH];
  __shared__ unsigned s_desc[SB_MAXCH + 5];
  __shared__ unsigned s_done;
  __shared__ int s_sums[3], s_G;
  const BlkShared sh{s_varena, s_parena, s_stage, s_smask, s_tab, s_masks, s_clr, s_arrLocal, s_arrCnt, s_arrQ, s_xCnt, s_xq, s_nbrBlk, s_nbrBin, s_nbr8,
                     s_cnt, s_total, s_gbase, s_nch, s_binQ, s_oc, s_chBin, s_chIdx, s_desc, &s_done, s_sums};
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int blk = (int)xcd_chunked(blockIdx.x, gridDim.x) + A.binBase / 8;
  const int bin0 = blk * 8;
  SLP_T0(tStart);
  // occupancy of the block's 8 bins: wave w <-> bin w
  const unsigned mask = A.cellMask[(size_t)(bin0 + w) * 64 + lane];
  s_masks[w][lane] = mask;
  {
    int n = __popc(mask);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) n += __shfl_xor(n, sft, 64);
    if (lane == 0) s_total[w] = n;
  }
  if (tid < 128) {
    s_clr[tid >> 6][tid & 63] = 0u;
    s_arrLocal[tid >> 6][tid & 63] = 0u;
  }
  if (tid < 192) s_arrCnt[tid >> 6][tid & 63] = 0u;
  if (tid < 3) {
    s_xCnt[tid] = 0u;
    s_sums[tid] = 0;
  }
  if (tid < 8) (&s_cnt[0][0])[tid] = 0;
  if (tid == 0) s_done = 0u;
  if (tid >= 64 && tid < 64 + 27) s_nbrBlk[tid - 64] = A.nbr27[(size_t)blk * 27 + (tid - 64)];
  if (tid >= 96 && tid < 104) s_nbr8[tid - 96] = A.nbr[(size_t)blk * 8 + (tid - 96)];
  int borg[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) borg[d] = t.activeKeys[3 * (size_t)blk + d] * (8 / mp.kscale);
  __syncthreads();
  if ((s_total[0] | s_total[1] | s_total[2] | s_total[3] | s_total[4] | s_total[5] | s_total[6] | s_total[7]) == 0) {
    // an empty block (the partition's apron: a quarter to a half of all blocks) leaves before it gathers a velocity arena; it has no grid
    // sums, so it reports to the one-launch schedule's counter at once
    if (tid < 8) A.moverCount[bin0 + tid] = 0;
    if (A.signal && blk < A.signalBlocks && tid == 0) atomicAdd(A.signal, 1ull);
    return;
  }
  if (tid == 0) {  // the block's chunk sequence
    int G = 0, q = 0;
    for (int b = 0; b < 8; ++b) {
      const int nch = (s_total[b] + 255) >> 8;
      s_nch[b] = nch;
      s_gbase[b] = G;
      s_binQ[b] = q;
      s_oc[b] = 0;
      if (nch) ++q;
      for (int c = 0; c < nch; ++c) {
        s_chBin[G] = (unsigned char)b;
        s_chIdx[G] = (unsigned char)c;
        s_desc[G + 1] = ChunkDesc::pack(b, c, s_binQ[b], c == nch - 1, s_total[b], s_gbase[b]);
        ++G;
      }
    }
    s_desc[0] = 0u;
    for (int k = G + 1; k < G + 5; ++k) s_desc[k] = 0u;
    s_G = G;
    // early warning of the closed-loop re-partition: the block holds particles and a block within {-1..2}^3 of it is missing
    if (G && A.blockEdge && A.blockEdge[blk]) A.status[3] = 1;
  }
  // the block's velocity arena: 10^3 nodes of grid A (this block and the 7 blocks at offsets {0,1}^3)
  for (int n = tid; n < 1000; n += 512) {
    const int x = n / 100, y = (n / 10) % 10, z = n % 10;
    const int slot = ((x >= 8) << 2) | ((y >= 8) << 1) | (z >= 8);
    const int cell = ((x & 7) * 8 + (y & 7)) * 8 + (z & 7);
    const int bn = s_nbr8[slot];
    float *a = s_varena + ArenaBlk::at(x, y, z);
    const float *g = A.gridA + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) a[ch * ArenaBlk::CH] = bn >= 0 ? g[ch * NC] : 0.f;
  }
  __syncthreads();
  const int G = s_G;
  if (tid < 8 && s_total[tid] == 0) A.moverCount[bin0 + tid] = 0;
  {  // entry tables and neighbour bins of the bins of chunks 0 and 1 (all eight waves; later ones: the producers, two chunks ahead)
    const int b0 = s_chBin[0];
    blk_build_tab(s_masks[b0][lane], lane, w, 8, s_tab[s_binQ[b0] & 1]);
    if (tid < 27) s_nbrBin[s_binQ[b0] & 1][tid] = blk_neighbour_bin(s_nbrBlk, blk, b0, tid);
    if (G > 1 && s_chIdx[1] == 0) {
      const int b1 = s_chBin[1];
      blk_build_tab(s_masks[b1][lane], lane, w, 8, s_tab[s_binQ[b1] & 1]);
      if (tid >= 64 && tid < 64 + 27) s_nbrBin[s_binQ[b1] & 1][tid - 64] = blk_neighbour_bin(s_nbrBlk, blk, b1, tid - 64);
    }
  }
  __syncthreads();
  if (w == 0) SLP_ADD(1, tStart);
  if (w < 4) blk_producer<SMODEL, WRITE_ALL>(mp, ps, borg, blk, lane, G, sh, A, __builtin_amdgcn_readfirstlane(w));
  else {
    if (w == 4) blk_consumer<0>(mp, borg, blk, lane, G, sh, A);
    else if (w == 5) blk_consumer<1>(mp, borg, blk, lane, G, sh, A);
    else if (w == 6) blk_consumer<2>(mp, borg, blk, lane, G, sh, A);
    else blk_consumer<3>(mp, borg, blk, lane, G, sh, A);
  }
  __syncthreads();  // the last bin is finished (blk_finish_bin by the consumer wave of set 3)
  if (w == 0) {
    SLP_ADD(0, tStart);
    SLP_PUT(11, 1);
  }
  if (tid == 0) {
    // movers sent / re-homed: running sums spread over SL_NCTR words (one device-wide word serves ~90 atomics per microsecond)
    if (s_sums[0]) atomicAdd(&A.status[SL_SENT + (blk & (SL_NCTR - 1))], s_sums[0]);
    if (s_sums[1]) atomicAdd(&A.status[SL_DELIVERED + (blk & (SL_NCTR - 1))], s_sums[1]);
  }
  if (s_sums[2]) {  // (rare) a chunk had more than SL_XQ movers for the consumers' list: the flagged records of those bins -> grid, all eight waves
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores have reached L2 ...
    __syncthreads();                                    // ... and so have everybody else's
    constexpr int RB = SB_NG * G2P2G_QF * 64 / SL_REC;  // records per batch: the staging ring is free now
    for (int b = 0; b < 8; ++b) {
      if (!((s_sums[2] >> b) & 1)) continue;
      const SubGeom sg = sub_geom(borg, b);
      const int oc = s_oc[b];
      for (int j0 = 0; j0 < oc; j0 += RB) {
        const int nb = oc - j0 < RB ? oc - j0 : RB;
        const float *src = A.moverRec + ((size_t)(bin0 + b) * A.cap + (size_t)j0) * SL_REC;
        // agent-scope loads: served by L2, where the stores are (never by an L1 line of this CU)
        for (int k = tid; k < nb * SL_REC; k += 512) s_stage[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        outbox_scatter_global<8>(mp, sg, s_stage, nb, w, lane, s_nbrBlk, A.gridB, A.status);
        __syncthreads();
      }
    }
  }
  if (A.signal && blk < A.signalBlocks) {  // (workgroup-uniform) one of the blocks whose grid sums somebody is waiting for
    // this wave's atomic adds to grid B have been performed (they are device-wide operations: nothing of them sits in this XCD's L2, so
    // no release fence -- an agent-scope fence writes the whole L2 back, and a thousand workgroups doing that cost a third of the step) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // ... and every wave's
    if (tid == 0) atomicAdd(A.signal, 1ull);
  }
}

template <int M, bool WA> static void launch_blk(hipStream_t stream, unsigned nblk, const MpmDev &mp, const ParticlesDev &pd, const BhtDev &t, const SlotArgs &A) {
  hipLaunchKernelGGL((g2p2g_slotblk_kernel<M, WA>), dim3(nblk), dim3(512), 0, stream, mp, pd, t, A);
}

// launched by zs_rocm_mpm_g2p2g_slots (mpm_slotted.hip) for 8^3 blocks: blocks [A.binBase / 8, + A.nbins / 8)
void launch_g2p2g_slotblk(hipStream_t stream, int model, bool writeAll, const MpmDev &mp, const ParticlesDev &pd, const BhtDev &t, const SlotArgs &A) {
  const unsigned nblk = (unsigned)(A.nbins / 8);
  if (!nblk) return;
#define ZSR_BLK(M)                                        \
  case M:                                                 \
    if (writeAll) launch_blk<M, true>(stream, nblk, mp, pd, t, A); \
    else launch_blk<M, false>(stream, nblk, mp, pd, t, A);         \
    break;
  switch (model) {
    ZSR_BLK(ZS_MPM_FIXED_COROTATED)
    ZSR_BLK(ZS_MPM_DRUCKER_PRAGER)
    ZSR_BLK(ZS_MPM_VONMISES_FIXED_COROTATED)
    ZSR_BLK(ZS_MPM_NACC)
    ZSR_BLK(ZS_MPM_EQUATION_OF_STATE)
  }
#undef ZSR_BLK
}

}  // namespace zsr

#if defined(ZS_SLOT_PROBE) && defined(ZS_SLOT_PROBE_BLK)  // measurement-only build: read (and clear) the phase stamps of g2p2g_slotblk_kernel
extern "C" void zs_rocm_slot_probe(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zsr::g_slot_probe), sizeof(unsigned long long) * 32);  // (the block kernel's reader: 32 slots)
  if (reset) {
    unsigned long long z[32] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(zsr::g_slot_probe), z, sizeof(z));
  }
}
#endif
