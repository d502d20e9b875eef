// mpm_slotted.hip -- the motion-robust form of the fused G2P2G step: SLOTTED particle storage.
//
// Why.  The binned kernels key a lane to a cell and keep its 27-node stencil in registers, which is what makes them fast -- and
// what made them fragile: with the compact round-robin order of zs_rocm_mpm_bin_particles a particle that leaves its cell stays
// where it is stored, is scattered through LDS float atomics (same bin) or global atomics (other bin) from then on, and a full
// re-bin (3 ms per 64 Mi particles) is the only repair.  Measured at 0.05 cell per step (a 1 m/s drift at dx = 1/512, dt = 1e-4:
// 5 % of the particles cross a cell face per step) the step went from 4.6 ms to 10 ms right after a re-bin and then grew by 7 ms
// per step.  Here the storage order is an invariant the step itself maintains:
//
//   * storage = bins x K rounds x 64 lanes: slot (bin, r, lane) is element (bin K + r) 64 + lane of a TileVector<f32, 64>, so a
//     round of a bin is exactly one tile row (every load / store of the binned kernels stays one 256-B row per channel), and a
//     32-bit mask per cell says which of its K <= 32 rounds hold a particle.  A particle is ALWAYS stored under the cell of its
//     base node;
//   * the main kernel (role-split, as g2p2g_rs_kernel) does G2P + advection + constitutive update of every particle and scatters
//     the ones that stay in their cell from registers.  The slot of a particle whose base node changes ("mover") becomes a hole;
//   * r03: the workgroup that moves a particle finishes it (r02 ran a second kernel in which every bin pulled from the outboxes of its 27
//     neighbours and accumulated the arrivals: 1.8 ms per step of the 64 Mi-particle column, a chain of dependent loads per bin).  New
//     cell inside the bin: a ticket of the cell's LDS counter names a free round, the state is stored there at once, and the lane of
//     the new cell adds the particle's grid terms from the staging ring in an iteration in which it has no particle of its own.  New
//     cell in a neighbour bin: the consumer waves add its 27 x 7 node terms with global float atomics (lane = node x channel), its
//     state goes to an outbox record, and slot_rehome_kernel (one lane per record, after the main kernel) draws a ticket from the
//     destination cell's global counter and copies the state into that round.  slot_commit_kernel folds departures and tickets into
//     the occupancy words.  Details and the measured dead ends: comment of the main kernel.
//
// The result of a step is the same sum of the same per-particle terms as zs_rocm_mpm_g2p2g (different summation order); bins,
// re-bins and exact-path queues disappear from the time loop.  Capacity limits (K rounds per cell, `cap` records per outbox)
// are reported in the status words, never dropped silently.
#include "mpm_slot.hpp"

namespace zsr {

// ------------------------------------------------------------------------------------------------------------------ slotting
template <int SIDE>
static __global__ __launch_bounds__(256) void slot_assign_kernel(BhtDev t, Port<float> pos, size_t n, float dx, unsigned *cellCount, int K,
                                                                 int *srcOf, int *status, int kscale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[3];
  load_attr<3>(pos, i, p);
  int key[3], loc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int c = (int)floorf(p[d] * (1.0f / dx) - 0.5f);
    loc[d] = c & (SIDE - 1);
    key[d] = (c - loc[d]) / SIDE * kscale;
  }
  const int b = bht_query<3>(t, key);
  if (b < 0) {
    status[2] = 1;
    return;
  }
  const int sub = SIDE == 4 ? 0 : (((loc[0] >> 2) * 2 + (loc[1] >> 2)) * 2 + (loc[2] >> 2));
  const unsigned bin = (unsigned)b * bins_per_block<SIDE>() + sub;
  const unsigned lane = (unsigned)(((loc[0] & 3) * 4 + (loc[1] & 3)) * 4 + (loc[2] & 3));
  const unsigned r = atomicAdd(&cellCount[bin * 64u + lane], 1u);
  if (r >= (unsigned)K) {
    status[1] = 1;
    return;
  }
  srcOf[((size_t)bin * K + r) * 64 + lane] = (int)i;
}
static __global__ __launch_bounds__(256) void slot_mask_kernel(unsigned *cellCountToMask, size_t ncells, int K) {
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  unsigned n = cellCountToMask[c];
  if (n > (unsigned)K) n = (unsigned)K;
  cellCountToMask[c] = n >= 32u ? 0xffffffffu : ((1u << n) - 1u);
}
// dst(:, s) = srcOf[s] >= 0 ? src(:, srcOf[s]) : 0 for every slot s (tile width 64 on both sides)
static __global__ __launch_bounds__(256) void slot_gather_kernel(const float *src, float *dst, size_t nslots, int C, const int *srcOf) {
  size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  const int i = srcOf[s];
  float *d = dst + (s >> 6) * (size_t)C * 64 + (s & 63);
  if (i < 0) {
    for (int c = 0; c < C; ++c) d[(size_t)c * 64] = 0.f;
    return;
  }
  const float *q = src + ((size_t)i >> 6) * (size_t)C * 64 + ((size_t)i & 63);
  for (int c = 0; c < C; ++c) d[(size_t)c * 64] = q[(size_t)c * 64];
}
// list of the occupied slots in slot order: per cell popcount -> scan (host) -> emit
static __global__ __launch_bounds__(256) void slot_popc_kernel(const unsigned *cellMask, size_t ncells, unsigned *cnt) {
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncells) cnt[c] = (unsigned)__popc(cellMask[c]);
}
static __global__ __launch_bounds__(256) void slot_emit_kernel(const unsigned *cellMask, const unsigned *start, size_t ncells, int K, int *slots) {
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  unsigned m = cellMask[c], o = start[c];
  const size_t bin = c >> 6, lane = c & 63;
  while (m) {
    const int r = __ffs((int)m) - 1;
    m &= m - 1;
    slots[o++] = (int)((bin * (size_t)K + (size_t)r) * 64 + lane);
  }
}
static __global__ __launch_bounds__(256) void build_neighbors27_kernel(BhtDev t, int nblocks, int *nbr27, int kscale) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nblocks * 27) return;
  const int i = (int)(g / 27), o = (int)(g % 27);
  int k[3] = {t.activeKeys[3 * (size_t)i] + (o / 9 - 1) * kscale, t.activeKeys[3 * (size_t)i + 1] + ((o / 3) % 3 - 1) * kscale,
              t.activeKeys[3 * (size_t)i + 2] + (o % 3 - 1) * kscale};
  nbr27[g] = bht_query<3>(t, k);
}

// edge[i] = 1 if a block at offset [lo, hi]^3 of block i is not in the partition
static __global__ __launch_bounds__(256) void partition_edge_kernel(BhtDev t, int nblocks, unsigned char *edge, int kscale, int lo, int hi) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= nblocks) return;
  const int k0[3] = {t.activeKeys[3 * (size_t)i], t.activeKeys[3 * (size_t)i + 1], t.activeKeys[3 * (size_t)i + 2]};
  int missing = 0;
  for (int a = lo; a <= hi; ++a)
    for (int b = lo; b <= hi; ++b)
      for (int c = lo; c <= hi; ++c) {
        int k[3] = {k0[0] + a * kscale, k0[1] + b * kscale, k0[2] + c * kscale};
        missing |= bht_query<3>(t, k) < 0;
      }
  edge[i] = (unsigned char)missing;
}

template <int SIDE, int SMODEL, bool WRITE_ALL>
static __global__ __launch_bounds__(512, 4) void g2p2g_slot_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, SlotArgs A) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float s_varena[3 * AL::CH];
  __shared__ float s_stage[SL_NG * G2P2G_QF * 64];
  float *const s_parena = s_stage;  // the bin's P2G arena (7 * AL::CH floats) is filled after the last chunk has been consumed: it shares the ring
  __shared__ unsigned long long s_smask[SL_NG];
  __shared__ unsigned short s_tab[SL_KMAX * 64];
  __shared__ unsigned s_mask0[64], s_clr[64], s_arrLocal[64], s_arrCnt[3][64];
  __shared__ unsigned short s_arrQ[3][64][SL_ARRQ];
  __shared__ int s_nbrBlk[27], s_nbrBin[27];
  __shared__ unsigned s_xCnt[3], s_xq[3][SL_XQ];
  __shared__ int s_outCount, s_sent, s_homed, s_xOver;
  const SlotShared sh{s_varena, s_parena, s_stage, s_smask, s_tab, s_mask0, s_clr, s_arrLocal, s_nbrBlk, s_nbrBin, s_arrCnt, s_arrQ,
                      s_xCnt, s_xq, &s_outCount, &s_sent, &s_homed, &s_xOver};
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bin = (int)xcd_chunked(blockIdx.x, gridDim.x) + A.binBase;
  SLP_T0(tStart);
  const unsigned mask = A.cellMask[(size_t)bin * 64 + lane];
  // round-major enumeration of the occupied slots (every wave walks the rounds; wave w fills the table rows of rounds = w mod 8)
  unsigned any = mask;
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) any |= (unsigned)__shfl_xor((int)any, sft, 64);
  if (any == 0u) {
    if (tid == 0) A.moverCount[bin] = 0;
    return;
  }
  const int nrounds = 32 - __clz((int)any);
  int total = 0;
  {
    const unsigned long long lt = lanemask_lt();
    for (int r = 0; r < nrounds; ++r) {
      const bool has = (mask >> r) & 1u;
      const unsigned long long occ = __ballot(has);
      if ((r & 7) == w && has) s_tab[total + __popcll(occ & lt)] = (unsigned short)(r * 64 + lane);
      total += __popcll(occ);
    }
  }
  const int nchunks = (total + 255) >> 8;
  if (tid < 64) {
    s_mask0[tid] = mask;  // (tid < 64: lane == tid, `mask` is cell tid's)
    s_clr[tid] = 0u;
    s_arrLocal[tid] = 0u;
    s_arrCnt[0][tid] = s_arrCnt[1][tid] = s_arrCnt[2][tid] = 0u;
  }
  if (tid == 0) s_outCount = s_sent = s_homed = s_xOver = 0;
  if (tid < 3) s_xCnt[tid] = 0u;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  // early warning of the closed-loop re-partition: this bin holds particles and its block has a missing block within {-1..2}^3 -- a
  // particle that crosses one more block could find no slot (or its stencil no node) there
  if (tid == 0 && A.blockEdge && A.blockEdge[geo.block]) A.status[3] = 1;
  if (tid >= 64 && tid < 64 + 27) {
    const int code = tid - 64;
    s_nbrBlk[code] = A.nbr27[(size_t)geo.block * 27 + code];
    s_nbrBin[code] = code == 13 ? bin : neighbour_bin<SIDE>(A.nbr27, geo.block, bin, code);
  }
  __syncthreads();  // the table is complete
  if (w == 0) SLP_ADD(1, tStart);
  if (w == 0) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 0>(mp, ps, geo, bin, total, lane, nchunks, sh, A);
  else if (w == 1) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 1>(mp, ps, geo, bin, total, lane, nchunks, sh, A);
  else if (w == 2) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 2>(mp, ps, geo, bin, total, lane, nchunks, sh, A);
  else if (w == 3) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 3>(mp, ps, geo, bin, total, lane, nchunks, sh, A);
  else {
    if (w == 4) g2p2g_slot_consumer<SIDE, 0>(mp, geo, mask, total, lane, nchunks, sh, A);
    else if (w == 5) g2p2g_slot_consumer<SIDE, 1>(mp, geo, mask, total, lane, nchunks, sh, A);
    else if (w == 6) g2p2g_slot_consumer<SIDE, 2>(mp, geo, mask, total, lane, nchunks, sh, A);
    else g2p2g_slot_consumer<SIDE, 3>(mp, geo, mask, total, lane, nchunks, sh, A);
  }
  SLP_T0(tTail);
  __syncthreads();  // all channel sets are in the arena
  if (tid < 64) {  // this step's departures and in-bin arrivals of the bin's cells, for slot_rehome_kernel / slot_commit_kernel
    const unsigned c = s_clr[tid], nl = s_arrLocal[tid];
    if (c) A.claim[((size_t)A.nbinsAll + (size_t)bin) * 64 + tid] = c;
    if (nl) A.claim[(size_t)bin * 64 + tid] = nl << 16;  // (the low half -- arrivals from other bins -- is counted after this kernel)
  }
  if (tid == 0) {
    // movers sent / re-homed: running sums spread over SL_NCTR words -- one device-wide word serves ~90 atomics per microsecond,
    // i.e. 1.5 ms for one add per bin of the 64 M-particle column
    A.moverCount[bin] = s_outCount < A.cap ? s_outCount : A.cap;  // records in the outbox
    if (s_sent) atomicAdd(&A.status[SL_SENT + (bin & (SL_NCTR - 1))], s_sent);
    if (s_homed) atomicAdd(&A.status[SL_DELIVERED + (bin & (SL_NCTR - 1))], s_homed);
  }
  if (tid < 216) {
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = A.nbr[(size_t)geo.block * 8 + slot];
    const float *a = s_parena + AL::at(x, y, z);
    if (bn >= 0) {
      float *g = A.gridB + (size_t)bn * 7 * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        const float v = a[ch * AL::CH];
        if (v != 0.f) unsafeAtomicAdd(g + ch * NC, v);
      }
    } else if (a[0] != 0.f) {
      A.status[2] = 1;  // mass for a node whose block is not in the partition
    }
  }
  if (s_xOver > 0) {  // (rare) a chunk had more than SL_XQ movers for the consumers' list: their full records -> grid, all eight waves
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores have reached L2 ...
    __syncthreads();                                    // ... and so have everybody else's (and the arena, which shares the ring, has been flushed)
    const int oc = s_outCount < A.cap ? s_outCount : A.cap;
    constexpr int RB = SL_NG * G2P2G_QF * 64 / SL_REC;  // records per batch: the whole staging ring is free now
    for (int j0 = 0; j0 < oc; j0 += RB) {
      const int nb = oc - j0 < RB ? oc - j0 : RB;
      const float *src = A.moverRec + ((size_t)bin * A.cap + (size_t)j0) * SL_REC;
      // agent-scope loads: served by L2, where the stores are (never by an L1 line of this CU)
      for (int k = tid; k < nb * SL_REC; k += 512) s_stage[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      outbox_scatter_global<SIDE>(mp, geo, s_stage, nb, w, lane, s_nbrBlk, A.gridB, A.status);
      __syncthreads();
    }
  }
  if (w == 0) {
    SLP_ADD(10, tTail);
    SLP_ADD(0, tStart);
    SLP_PUT(11, 1);
  }
}

// after the main kernel: one wave per bin, one lane per outbox record; records that left their bin (word SLR_DCELL = destination cell)
// get their slot here.  Ticket of the destination cell's counter (low half of the claim word; the high half holds the cell's in-bin
// arrivals, final by now) -> the free rounds above the in-bin arrivals, from the bottom; the particle's state is copied from the
// record's first line into that slot.
template <bool FLUID, bool DP, bool WRITE_ALL>
static __global__ __launch_bounds__(256) void slot_rehome_kernel(ParticlesDev ps, const unsigned *cellMask, unsigned *claim, const int *moverCount,
                                                                 const float *moverRec, int cap, size_t nbins, int K, int *status) {
  constexpr int LW = 64;
  // a wave takes EIGHT bins, eight records of each per pass (a bin sends ~7 records per step): the kernel is a chain of dependent round
  // trips (count -> record -> ticket -> stores), so what matters is how many waves stand in line, not how busy their lanes are --
  // one bin per wave: 0.43 ms per step of the 64 Mi column
  const size_t bin = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + ((threadIdx.x & 63) >> 3);
  const int n = bin < nbins ? moverCount[bin] : 0;
  if (__ballot(n > 0) == 0ull) return;
  const unsigned kmask = K >= 32 ? 0xffffffffu : ((1u << K) - 1u);
  int nhomed = 0;
  for (int k = (int)(threadIdx.x & 7); k < n; k += 8) {
    const float *rc = moverRec + (bin * (size_t)cap + (size_t)k) * SL_REC;
    const float4 r0 = reinterpret_cast<const float4 *>(rc)[0], r1 = reinterpret_cast<const float4 *>(rc)[1],
                 r2 = reinterpret_cast<const float4 *>(rc)[2], r3 = reinterpret_cast<const float4 *>(rc)[3];
    const unsigned dcell = __float_as_uint(r3.z);
    if (dcell == 0xffffffffu) continue;  // it stayed inside its bin (arrival queue full) or kept its slot: its workgroup stored it
    const unsigned occ = cellMask[dcell];
    const unsigned old = atomicAdd(&claim[dcell], 1u);
    const int rr = nth_low_bit(~occ & kmask, (old >> 16) + (old & 0xffffu));
    size_t i;
    if (rr >= 0) {
      i = ((size_t)(dcell >> 6) * (size_t)K + (size_t)rr) * 64 + (size_t)(dcell & 63u);
    } else {
      // destination cell full: the particle goes back into the slot it left (still free: this step's arrivals only take rounds that
      // were free when the step began) with its new state, and the departure is withdrawn before slot_commit_kernel folds it in.  It
      // is then stored under the wrong cell -- reported ([1]); the caller re-slots the storage -- but it is not lost.
      status[1] = 1;
      const unsigned src = __float_as_uint(r3.w) >> 1;  // round * 64 + cell inside this bin
      i = (bin * (size_t)K + (size_t)(src >> 6)) * 64 + (size_t)(src & 63u);
      atomicAnd(&claim[(nbins + bin) * 64 + (size_t)(src & 63u)], ~(1u << (src >> 6)));
    }
    ++nhomed;
    const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
    const float x[3] = {r0.y, r0.z, r0.w};
    const float F[9] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x};
    pstore1<LW>(ps.mass, o, r0.x);
    pstore<LW, 3>(ps.pos, o, x);
    pstore_state<LW, FLUID>(ps.F, o, F);
    if constexpr (DP) pstore1<LW>(ps.logJp, o, r3.y);
    if constexpr (WRITE_ALL) {
      float v[3], C[9], PF[9];
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] = rc[SLR_V + d];
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        C[d] = rc[SLR_C + d];
        PF[d] = rc[SLR_PF + d];
      }
      pstore<LW, 3>(ps.vel, o, v);
      pstore<LW, 9>(ps.C, o, C);
      {
        float S[STRESS_N];
        stress_pack(PF, S);
        pstore<LW, STRESS_N>(ps.stress, o, S);
      }
    }
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) nhomed += __shfl_xor(nhomed, sft, 64);
  if ((threadIdx.x & 63) == 0 && nhomed) atomicAdd(&status[SL_DELIVERED + (int)((bin >> 3) & (SL_NCTR - 1))], nhomed);
}

// after the step: departures leave the occupancy words, this step's arrivals enter them (the lowest free rounds: in-bin arrivals first,
// then the arrivals from other bins), the counters are zero again
static __global__ __launch_bounds__(256) void slot_commit_kernel(unsigned *cellMask, unsigned *claim, size_t ncells, int K) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  const unsigned tk = claim[c], clr = claim[ncells + c];
  if (!tk && !clr) return;
  const unsigned m0 = cellMask[c];
  const unsigned kmask = K >= 32 ? 0xffffffffu : ((1u << K) - 1u);
  const unsigned n = (tk >> 16) + (tk & 0xffffu);
  unsigned rest = ~m0 & kmask, bits = 0u;
  for (unsigned k = 0; k < n && rest; ++k) {
    const unsigned b = rest & (0u - rest);
    bits |= b;
    rest ^= b;
  }
  cellMask[c] = (m0 & ~clr) | bits;
  claim[c] = 0u;
  claim[ncells + c] = 0u;
}

// ------------------------------------------------------------------------------------------------------------------ re-partition in place
// ComputeSparsity (simulation/sparsity/SparsityOp.hpp:65-86) from the occupancy words: a particle is stored under the cell c of its base node, the
// reference inserts the block of (base node + 1 - 2) = c - 1 per axis -- no particle is read.  One thread per cell of the old partition.
template <int SIDE>
static __global__ __launch_bounds__(256) void slot_sparsity_kernel(BhtDev oldT, const unsigned *cellMask, size_t ncells, int kscale, BhtDev newT) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = c < ncells && cellMask[c] != 0u;
  int b[3] = {0, 0, 0};
  if (valid) {
    const size_t bin = c >> 6;
    const int lane = (int)(c & 63), blk = (int)(bin / bins_per_block<SIDE>()), sub = (int)(bin % bins_per_block<SIDE>());
    const int o[3] = {SIDE == 4 ? 0 : ((sub >> 2) & 1) * 4, SIDE == 4 ? 0 : ((sub >> 1) & 1) * 4, SIDE == 4 ? 0 : (sub & 1) * 4};
    const int l[3] = {lane >> 4, (lane >> 2) & 3, lane & 3};
#pragma unroll
    for (int d = 0; d < 3; ++d) b[d] = floordiv(oldT.activeKeys[3 * (size_t)blk + d] * (SIDE / kscale) + o[d] + l[d] - 1, SIDE) * kscale;
  }
  const int px = shfl_up(b[0], 1), py = shfl_up(b[1], 1), pz = shfl_up(b[2], 1);
  const bool pvalid = shfl_up((int)valid, 1) != 0;
  const bool dup = lane_id() != 0 && pvalid && px == b[0] && py == b[1] && pz == b[2];
  if (valid && !dup) bht_insert<3>(newT, b);
}
// map[i] = number of old block i in the new partition (-1: not there)
static __global__ __launch_bounds__(256) void reslot_map_kernel(BhtDev oldT, int nOld, BhtDev newT, int *map) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= nOld) return;
  const int k[3] = {oldT.activeKeys[3 * (size_t)i], oldT.activeKeys[3 * (size_t)i + 1], oldT.activeKeys[3 * (size_t)i + 2]};
  map[i] = bht_query<3>(newT, k);
}
// one workgroup per old bin: its occupancy words and its occupied rounds (whole tile rows: rounds x C channels x 64 lanes are contiguous)
// move to the bin's place in the new partition
static __global__ __launch_bounds__(256) void reslot_move_kernel(const int *map, int bpb, int K, int C, const float4 *oldBuf, float4 *newBuf,
                                                                 const unsigned *oldMask, unsigned *newMask, int *status) {
  const size_t bin = blockIdx.x;
  __shared__ unsigned sAny;
  if (threadIdx.x == 0) sAny = 0u;
  __syncthreads();
  const unsigned m = threadIdx.x < 64 ? oldMask[bin * 64 + threadIdx.x] : 0u;
  if (m) atomicOr(&sAny, m);
  __syncthreads();
  const unsigned any = sAny;
  if (!any) return;
  const int nb = map[bin / bpb];
  if (nb < 0) {
    if (threadIdx.x == 0) status[2] = 1;  // a block that holds particles is not in the new partition (cannot happen with slot_sparsity + enlarge)
    return;
  }
  const size_t nbin = (size_t)nb * bpb + bin % bpb;
  if (threadIdx.x < 64) newMask[nbin * 64 + threadIdx.x] = m;
  const int rounds = 32 - __clz((int)any);
  const size_t per = (size_t)K * C * 16, cnt = (size_t)rounds * C * 16;  // float4 per bin / to copy
  const float4 *src = oldBuf + bin * per;
  float4 *dst = newBuf + nbin * per;
  for (size_t i = threadIdx.x; i < cnt; i += 256) dst[i] = src[i];
}
// grid blocks (7 channels x side^3 cells) of the old partition -> their place in the new one
static __global__ __launch_bounds__(256) void reslot_grid_kernel(const int *map, int nOld, int blockFloats4, const float4 *oldGrid, float4 *newGrid) {
  const int b = blockIdx.x;
  const int nb = map[b];
  if (nb < 0) return;
  for (int i = threadIdx.x; i < blockFloats4; i += 256) newGrid[(size_t)nb * blockFloats4 + i] = oldGrid[(size_t)b * blockFloats4 + i];
}

}  // namespace zsr

using namespace zsr;

extern "C" {

#if defined(ZS_SLOT_PROBE) && !defined(ZS_SLOT_PROBE_BLK)  // measurement-only build: read (and clear) the phase stamps of g2p2g_slot_kernel
void zs_rocm_slot_probe(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zsr::g_slot_probe), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(zsr::g_slot_probe), z, sizeof(z));
  }
}
#endif

size_t zs_rocm_mpm_slot_outbox_bytes(size_t nbins, int cap, int which) {
  if (which == 0) return nbins * sizeof(int);                              // moverCount
  if (which == 1) return 2 * nbins * 64 * sizeof(unsigned);                // ticket + departure words (two per cell; zeroed by the caller once)
  return nbins * (size_t)cap * (size_t)SL_REC * sizeof(float);             // moverRec
}

void zs_rocm_mpm_build_neighbors27(zs_rocm_policy *pol, const zs_rocm_bht_3 *tab, int *nbr27, int keyStride) {
  Launch L(pol, "build_neighbors27");
  const int nb = bht_size(tab->t, L.stream);
  if (!nb) return;
  hipLaunchKernelGGL(build_neighbors27_kernel, dim3(ceil_div((size_t)nb * 27, 256)), dim3(256), 0, L.stream, tab->t.dev(), nb, nbr27,
                     keyStride > 0 ? keyStride : 1);
}

// particles (any order, TileVector<f32,64> with C channels, n elements in `src`) -> slotted storage `dst` (nbins * K tiles of 64):
// cellMask[nbins*64] is written; status[1] is set if a cell holds more than K particles (those are NOT stored), status[2] if a
// particle lies outside the partition.  Returns 0, or -1 on bad arguments.
int zs_rocm_mpm_slot_particles(zs_rocm_policy *pol, const zs_rocm_bht_3 *tab, zs_rocm_attr pos, size_t n, float dx, int side, int keyIsOrigin,
                               int K, const float *src, float *dst, int C, unsigned *cellMask, int *status) {
  if (K < 1 || K > 32 || (side != 4 && side != 8) || pos.tileMask != 63u) return -1;
  Launch L(pol, "slot_particles");
  const int nb = bht_size(tab->t, L.stream);
  if (nb == 0) return 0;
  const size_t nbins = (size_t)nb * (side == 4 ? 1 : 8), ncells = nbins * 64, nslots = nbins * (size_t)K * 64;
  int *srcOf = (int *)L.temp(sizeof(int) * nslots);
  ZSR_CHECK(hipMemsetAsync(srcOf, 0xff, sizeof(int) * nslots, L.stream));
  ZSR_CHECK(hipMemsetAsync(cellMask, 0, sizeof(unsigned) * ncells, L.stream));
  BhtDev t = tab->t.dev();
  Port<float> pp = make_port<float>(pos);
  if (n) {
    if (side == 4)
      hipLaunchKernelGGL((slot_assign_kernel<4>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t, pp, n, dx, cellMask, K, srcOf, status, keyIsOrigin ? side : 1);
    else
      hipLaunchKernelGGL((slot_assign_kernel<8>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t, pp, n, dx, cellMask, K, srcOf, status, keyIsOrigin ? side : 1);
  }
  hipLaunchKernelGGL(slot_mask_kernel, dim3(ceil_div(ncells, 256)), dim3(256), 0, L.stream, cellMask, ncells, K);
  hipLaunchKernelGGL(slot_gather_kernel, dim3(ceil_div(nslots, 256)), dim3(256), 0, L.stream, src, dst, nslots, C, (const int *)srcOf);
  return 0;
}

// occupied slots in slot order -> slots[0 .. count); returns count (synchronises the stream)
size_t zs_rocm_mpm_slot_list(zs_rocm_policy *pol, const unsigned *cellMask, size_t nbins, int K, int *slots) {
  Launch L(pol, "slot_list");
  const size_t ncells = nbins * 64;
  if (!ncells) return 0;
  unsigned *cnt = (unsigned *)L.temp(sizeof(unsigned) * (ncells + 1)), *start = (unsigned *)L.temp(sizeof(unsigned) * (ncells + 1));
  ZSR_CHECK(hipMemsetAsync(cnt + ncells, 0, sizeof(unsigned), L.stream));
  hipLaunchKernelGGL(slot_popc_kernel, dim3(ceil_div(ncells, 256)), dim3(256), 0, L.stream, cellMask, ncells, cnt);
  exclusive_scan_u32(L, cnt, ncells + 1, start);
  unsigned total = 0;
  ZSR_CHECK(hipMemcpyAsync(&total, start + ncells, sizeof(unsigned), hipMemcpyDeviceToHost, L.stream));
  if (slots) hipLaunchKernelGGL(slot_emit_kernel, dim3(ceil_div(ncells, 256)), dim3(256), 0, L.stream, cellMask, (const unsigned *)start, ncells, K, slots);
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  return (size_t)total;
}

// The fused step on slotted storage.  particles: attributes of ONE TileVector<f32, 64> with nbins*K*64 elements (particles.n);
// gridB zeroed by the caller; mover buffers sized by zs_rocm_mpm_slot_outbox_bytes (the claim words zeroed by the caller ONCE: every
// step leaves them zero); status: int[ZS_ROCM_SLOT_STATUS_WORDS], zeroed by the caller when it wants to (they latch).
// blocks [blockBegin, blockEnd) only; finish != 0: also give the step's outbox records their new slots and fold departures / arrivals
// into the occupancy words -- ONCE per step, after the ranges that cover all blocks.  Multi-GPU step (bench.py): the partition is
// numbered with the blocks near a rank boundary first; their range is launched first and their ghost-block sums travel on a second
// stream while the interior range computes (a bin writes grid nodes at most one block away from its own).
// Returns 0, -1 on bad arguments.
int zs_rocm_mpm_g2p2g_slots(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab,
                            const float *gridA, float *gridB, size_t nblocks, const zs_rocm_slot_storage *st, int writeAll,
                            size_t blockBegin, size_t blockEnd, int finish) {
  return zsr::mpm_g2p2g_slots_signal(pol, p, ps, tab, gridA, gridB, nblocks, st, writeAll, blockBegin, blockEnd, finish, nullptr, 0);
}
}  // extern "C"
namespace zsr {
// ... and with `signal`: every workgroup of the blocks [0, signalBlocks) of the range adds 1 to *signal when its sums have reached grid B
// (8^3 blocks only; zs_rocm_mpm_step_slotted's one-launch schedule waits for the count on its exchange stream)
int mpm_g2p2g_slots_signal(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, const float *gridA,
                           float *gridB, size_t nblocks, const zs_rocm_slot_storage *st, int writeAll, size_t blockBegin, size_t blockEnd, int finish,
                           unsigned long long *signal, size_t signalBlocks) {
  if (!st) return -1;
  if (signal && p->side != 8) return -1;
  unsigned *const cellMask = st->cellMask;
  const int K = st->K, outboxCap = st->outboxCap;
  const int *const nbr = st->nbr, *const nbr27 = st->nbr27;
  int *const moverCount = st->moverCount, *const status = st->status;
  unsigned *const claim = st->claim;
  float *const moverRec = st->moverRec;
  if (!nblocks) return 0;
  if (!cellMask || !nbr || !nbr27 || !moverCount || !claim || !moverRec || !status || K < 1 || K > 32 || outboxCap < 1 || outboxCap > (1 << 20)) return -1;
  if (p->model < ZS_MPM_FIXED_COROTATED || p->model > ZS_MPM_EQUATION_OF_STATE) return -1;
  if (uniform_lane_width(ps, model_uses_logjp(p->model), writeAll != 0) != 64 || (writeAll && (!ps.vel.base || !ps.C.base))) {
    fprintf(stderr, "[zs_rocm] g2p2g_slotted needs all particle attributes in one TileVector<f32, 64>\n");
    return -1;
  }
  if (blockEnd > nblocks) blockEnd = nblocks;
  Launch L(pol, "G2P2GTransfer(slotted)");
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  const unsigned bpb = p->side == 4 ? 1u : 8u;
  const unsigned nbinsAll = (unsigned)(nblocks * bpb);
  const unsigned nbins = blockBegin < blockEnd ? (unsigned)((blockEnd - blockBegin) * bpb) : 0u;
  const SlotArgs A{gridA, gridB, cellMask, K, nbr, nbr27, moverCount, claim, moverRec, status, (int)(blockBegin * bpb), (int)nbins, (int)nbinsAll, outboxCap,
                   st->blockEdge, signal, (int)signalBlocks};
#define CALL_REHOME3(M, WA)                                                                                                              \
  hipLaunchKernelGGL((slot_rehome_kernel<model_is_fluid(M), model_uses_logjp(M), WA>), dim3(ceil_div((size_t)nbinsAll, 32)), dim3(256), 0,  \
                     L.stream, pd, (const unsigned *)cellMask, claim, (const int *)moverCount, (const float *)moverRec, outboxCap,        \
                     (size_t)nbinsAll, K, status)
#define CALL_REHOME(SS, M)                      \
  do {                                          \
    if (writeAll) { CALL_REHOME3(M, true); }    \
    else { CALL_REHOME3(M, false); }            \
  } while (0)
  // 8^3 blocks: one workgroup per block (mpm_slotblk.hip); 4^3 blocks (bin == block): one workgroup per bin (g2p2g_slot_kernel<4, ...>)
  if (p->side == 8) {
    if (nbins) launch_g2p2g_slotblk(L.stream, p->model, writeAll != 0, mp, pd, t, A);
  } else if (nbins) {
#define CALL_SLOT4(SS, M)                                                                                                  \
  do {                                                                                                                     \
    if (writeAll) hipLaunchKernelGGL((g2p2g_slot_kernel<4, M, true>), dim3(nbins), dim3(512), 0, L.stream, mp, pd, t, A);  \
    else hipLaunchKernelGGL((g2p2g_slot_kernel<4, M, false>), dim3(nbins), dim3(512), 0, L.stream, mp, pd, t, A);          \
  } while (0)
    ZSR_DISPATCH_PURE_(4, p->model, CALL_SLOT4)
  }
  if (finish) ZSR_DISPATCH_PURE_(0, p->model, CALL_REHOME)
  if (finish) {
    const size_t ncells = (size_t)nbinsAll * 64;
    hipLaunchKernelGGL(slot_commit_kernel, dim3(ceil_div(ncells, 256)), dim3(256), 0, L.stream, cellMask, claim, ncells, K);
  }
  return 0;
}
}  // namespace zsr
extern "C" {

int zs_rocm_mpm_g2p2g_slotted_range(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab,
                                    const float *gridA, float *gridB, size_t nblocks, unsigned *cellMask, int K, const int *nbr, const int *nbr27,
                                    int *moverCount, unsigned *claim, float *moverRec, int outboxCap, int writeAll, int *status,
                                    size_t blockBegin, size_t blockEnd, int finish) {
  const zs_rocm_slot_storage st{cellMask, K, nbr, nbr27, moverCount, claim, moverRec, outboxCap, status, nullptr};
  return zs_rocm_mpm_g2p2g_slots(pol, p, ps, tab, gridA, gridB, nblocks, &st, writeAll, blockBegin, blockEnd, finish);
}

int zs_rocm_mpm_g2p2g_slotted(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, const float *gridA,
                              float *gridB, size_t nblocks, unsigned *cellMask, int K, const int *nbr, const int *nbr27, int *moverCount,
                              unsigned *claim, float *moverRec, int outboxCap, int writeAll, int *status) {
  return zs_rocm_mpm_g2p2g_slotted_range(pol, p, ps, tab, gridA, gridB, nblocks, cellMask, K, nbr, nbr27, moverCount, claim, moverRec, outboxCap,
                                         writeAll, status, 0, nblocks, 1);
}

// edge[i] = 1 if one of the blocks at offsets [lo, hi]^3 (in blocks) of block i is missing from the partition.  With lo = -1, hi = 2 a block
// with edge == 0 can lose a particle to any of its 26 neighbours and that particle's stencil (+{0,1}^3) still finds its nodes.
void zs_rocm_mpm_partition_edge(zs_rocm_policy *pol, const zs_rocm_bht_3 *tab, unsigned char *edge, int keyStride, int lo, int hi) {
  Launch L(pol, "partition_edge");
  const int nb = bht_size(tab->t, L.stream);
  if (!nb) return;
  hipLaunchKernelGGL(partition_edge_kernel, dim3(ceil_div((size_t)nb, 256)), dim3(256), 0, L.stream, tab->t.dev(), nb, edge,
                     keyStride > 0 ? keyStride : 1, lo, hi);
}

// ---- re-partition of slotted storage without touching a particle.  (1) zs_rocm_mpm_slot_compute_sparsity: the reference's ComputeSparsity
// over the cells that hold particles (from the occupancy words) into `newTab` (reset by the caller); the caller then enlarges it
// (zs_rocm_mpm_enlarge_sparsity) exactly as for a fresh partition.  (2) zs_rocm_mpm_reslot: every bin that holds particles moves, as
// whole tile rows, to the number its block has in `newTab`; newMask / newGrid (optional: the grid of node velocities the next fused step
// gathers from) are written for the whole new partition (zero where nothing was).  Returns 0, -1 on bad arguments; status[2] is set if a
// populated block is missing from the new partition.
void zs_rocm_mpm_slot_compute_sparsity(zs_rocm_policy *pol, const zs_rocm_bht_3 *oldTab, const unsigned *cellMask, size_t nblocksOld, int side,
                                       int keyIsOrigin, zs_rocm_bht_3 *newTab) {
  Launch L(pol, "slot_compute_sparsity");
  if (!nblocksOld || (side != 4 && side != 8)) return;
  const size_t ncells = nblocksOld * (side == 4 ? 1 : 8) * 64;
  if (side == 4)
    hipLaunchKernelGGL((slot_sparsity_kernel<4>), dim3(ceil_div(ncells, 256)), dim3(256), 0, L.stream, oldTab->t.dev(), cellMask, ncells, keyIsOrigin ? side : 1, newTab->t.dev());
  else
    hipLaunchKernelGGL((slot_sparsity_kernel<8>), dim3(ceil_div(ncells, 256)), dim3(256), 0, L.stream, oldTab->t.dev(), cellMask, ncells, keyIsOrigin ? side : 1, newTab->t.dev());
}
int zs_rocm_mpm_reslot(zs_rocm_policy *pol, const zs_rocm_bht_3 *oldTab, const zs_rocm_bht_3 *newTab, int side, int K, int C, const float *oldBuf,
                       float *newBuf, const unsigned *oldMask, unsigned *newMask, const float *oldGrid, float *newGrid, int *status) {
  if ((side != 4 && side != 8) || K < 1 || K > 32 || C < 1 || !oldBuf || !newBuf || !oldMask || !newMask || !status) return -1;
  Launch L(pol, "reslot");
  const int nOld = bht_size(oldTab->t, L.stream), nNew = bht_size(newTab->t, L.stream);
  if (!nOld || !nNew) return 0;
  const int bpb = side == 4 ? 1 : 8;
  int *map = (int *)L.temp(sizeof(int) * (size_t)nOld);
  hipLaunchKernelGGL(reslot_map_kernel, dim3(ceil_div((size_t)nOld, 256)), dim3(256), 0, L.stream, oldTab->t.dev(), nOld, newTab->t.dev(), map);
  ZSR_CHECK(hipMemsetAsync(newMask, 0, sizeof(unsigned) * (size_t)nNew * bpb * 64, L.stream));
  hipLaunchKernelGGL(reslot_move_kernel, dim3((unsigned)((size_t)nOld * bpb)), dim3(256), 0, L.stream, (const int *)map, bpb, K, C,
                     reinterpret_cast<const float4 *>(oldBuf), reinterpret_cast<float4 *>(newBuf), oldMask, newMask, status);
  if (oldGrid && newGrid) {
    const size_t bf = (size_t)7 * side * side * side;
    ZSR_CHECK(hipMemsetAsync(newGrid, 0, sizeof(float) * bf * (size_t)nNew, L.stream));
    hipLaunchKernelGGL(reslot_grid_kernel, dim3((unsigned)nOld), dim3(256), 0, L.stream, (const int *)map, nOld, (int)(bf / 4),
                       reinterpret_cast<const float4 *>(oldGrid), reinterpret_cast<float4 *>(newGrid));
  }
  return 0;
}

}  // extern "C"
