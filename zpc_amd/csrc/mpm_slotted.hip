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
//     the ones that stay in their cell from registers.  A particle whose base node changes ("mover": same bin or not) is not
//     scattered: its state {m, x', F', logJp', v', C', P F^T} goes to the OUTBOX of its source bin (a fixed region per bin: no global
//     counter), with its destination cell; its slot becomes a hole (mask bit cleared);
//   * the mover kernel (one workgroup per bin) PULLS: it scans the outboxes of its 27 neighbour bins for records addressed to one
//     of its cells (a particle moves less than a cell per step), hands them to the lane of that cell, scatters them with the same
//     register-stencil code as the main kernel, and stores their state into free rounds of the cell.  No sort, no global atomics
//     except the arena flush, and nobody but the movers is ever relocated.
//
// The result of a step is the same sum of the same per-particle terms as zs_rocm_mpm_g2p2g (different summation order); bins,
// re-bins and exact-path queues disappear from the time loop.  Capacity limits (K rounds per cell, `cap` records per outbox,
// SL_MAXIN arrivals per cell and step) are reported in the status words, never dropped silently.
#include "mpm_device.hpp"

namespace zsr {

constexpr int SL_REC = 36;    // floats per record: m, x(3), F(9), logJp, v(3), C(9), P F^T(9), pad
constexpr int SL_MAXIN = 16;  // arrivals one cell can take per step

struct SlotArgs {
  const float *gridA;
  float *gridB;
  unsigned *cellMask;   // [nbins][64] occupancy of the K rounds of every cell
  int K;
  const int *nbr;       // [nblocks][8]  blocks at offsets {0,1}^3 (arena -> grid)
  const int *nbr27;     // [nblocks][27] blocks at offsets {-1,0,1}^3 (mover pull)
  int *moverCount;      // [nbins]
  long long *moverDest; // [nbins][cap] destination cell, packed 21 bits per axis (biased)
  float *moverRec;      // [nbins][cap][SL_REC]
  int *status;          // [0] outbox full, [1] cell full (K), [2] mass for a block outside the partition, [3] more than SL_MAXIN
                        // arrivals in one cell, [4] a particle was not stored under its cell; [8 .. 8 + 256) records sent,
                        // [264 .. 264 + 256) records delivered (running sums spread over 256 words each: unequal totals after a step = a
                        // mover's destination block is not in the partition)
  int binBase, nbins;
  int cap;              // outbox records per bin and step (caller's choice: a bin holds 512 particles at 8 per cell)
};

constexpr int SL_NCTR = 256, SL_SENT = 8, SL_DELIVERED = 8 + SL_NCTR;  // layout of the status words (zs_rocm.h: ZS_ROCM_SLOT_STATUS_WORDS)
__device__ __forceinline__ long long pack_cell(int x, int y, int z) {
  return ((long long)(x + (1 << 20)) << 42) | ((long long)(y + (1 << 20)) << 21) | (long long)(z + (1 << 20));
}
__device__ __forceinline__ void unpack_cell(long long p, int &x, int &y, int &z) {
  x = (int)((p >> 42) & 0x1fffff) - (1 << 20);
  y = (int)((p >> 21) & 0x1fffff) - (1 << 20);
  z = (int)(p & 0x1fffff) - (1 << 20);
}

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
constexpr int SL_NG = 9;       // staged entry groups of 64: a chunk being produced (4) + the chunk being consumed (4) + a straddling round
constexpr int SL_KMAX = 32;    // rounds per bin the 32-bit occupancy words allow

// producer wave W (0..3): entries [64 (4c + W), +64) of every chunk c
template <int SIDE, int SMODEL, bool WRITE_ALL, int W>
__device__ __forceinline__ void g2p2g_slot_producer(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, int bin, int total,
                                                    int lane, int nchunks, float *varena, float *stage, unsigned long long *smask,
                                                    const unsigned short *tab, int *outCount, unsigned *clrAll, const SlotArgs &A) {
  using AL = ArenaLds;
  constexpr int LW = 64;
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool FLUID = model_is_fluid(SMODEL);
  constexpr int NC = SIDE * SIDE * SIDE;
  const float dxi = 1.0f / mp.dx;
  const float D_inv = 4.f * dxi * dxi;
  const size_t rowBase = (size_t)bin * (size_t)A.K;
  RecG<LW, DP, FLUID> cur, nxt;
  bool has0 = false, has1 = false;
  size_t i0 = 0, i1 = 0;
  unsigned code0 = 0, code1 = 0;
  if (nchunks > 0) {
    const int j = 64 * W + lane;
    has0 = j < total;
    if (has0) {
      code0 = tab[j];
      i0 = (rowBase + (size_t)(code0 >> 6)) * 64 + (size_t)(code0 & 63u);
      cur.load(ps, i0);
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
  __syncthreads();
  for (int it = 0; it <= nchunks; ++it) {
    if (it < nchunks) {
      const int grp = 4 * it + W;
      float *myStage = stage + (size_t)(grp % SL_NG) * (G2P2G_NF * 64);
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
          const POff<LW> o = particle_offset<LW>(ps.pos.chns, i0);
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
#ifdef ZS_SLOT_NOMOVER  // measurement build: what does the presence of the mover path cost a step without movers?
          const bool moved = false;
#else
          const bool moved = nc[0] != cx || nc[1] != cy || nc[2] != cz;
#endif
          float *rec = nullptr;
          if (moved) {
            const int k = atomicAdd(outCount, 1);
            if (k < A.cap) {
              rec = A.moverRec + ((size_t)bin * A.cap + (size_t)k) * SL_REC;
              A.moverDest[(size_t)bin * A.cap + (size_t)k] = pack_cell(nc[0] + geo.org[0], nc[1] + geo.org[1], nc[2] + geo.org[2]);
#pragma unroll
              for (int d = 0; d < 3; ++d) rec[1 + d] = pos[d];
#pragma unroll
              for (int d = 0; d < 9; ++d) rec[4 + d] = F[d];
#pragma unroll
              for (int d = 0; d < 3; ++d) rec[14 + d] = vel[d];
#pragma unroll
              for (int d = 0; d < 9; ++d) rec[17 + d] = C[d];
            } else {
              A.status[0] = 1;  // outbox full: the particle (and its contribution) would be lost -- reported, the caller must react
            }
            atomicOr(&clrAll[cell], 1u << r);  // its slot becomes a hole
          } else {
            pstore_state<LW, FLUID>(ps.F, o, F);
            pstore<LW, 3>(ps.pos, o, pos);
            if (WRITE_ALL) {
              pstore<LW, 3>(ps.vel, o, vel);
              pstore<LW, 9>(ps.C, o, C);
            }
          }
          {  // the plastic models may project the local copy of F (the stored / recorded F is the unprojected one, P2G.hpp:101)
            float lj = 0.f;
            if constexpr (DP) lj = cur.logJp;
            model_stress<SMODEL>(mp.mat, lj, F, PF, C);
            if (moved) {
              if (rec) {
                rec[0] = cur.m;  // (the mass is the last value of the record load: stored here, its wait does not hold up the stores above)
                rec[13] = lj;
#pragma unroll
                for (int d = 0; d < 9; ++d) rec[26 + d] = PF[d];
              }
            } else {
              if constexpr (DP) pstore1<LW>(ps.logJp, o, lj);
              if (WRITE_ALL) pstore<LW, 9>(ps.stress, o, PF);
            }
          }
          if (!moved) {
            // staged AFTER the constitutive update, as in g2p2g_rs_producer: with m, x', v', C' dead before it the compiler
            // reuses their registers for the SVD at once and waits for the particle stores just issued (s_waitcnt vmcnt(1)
            // in front of the SVD: 2 ms per 64 Mi particles)
            valid = true;
            myStage[0 * 64 + lane] = cur.m;
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
      cur = nxt;
      has0 = has1;
      i0 = i1;
      code0 = code1;
    }
    __syncthreads();
  }
}
// consumer wave of channel set CS: lane = cell; after chunk c has been produced every round whose last entry lies below 256 (c + 1)
// is complete and is consumed while the producers work on chunk c + 1
template <int CS>
__device__ __forceinline__ void g2p2g_slot_consumer(const MpmDev &mp, unsigned mask, int total, int lane, int nchunks, const float *stage,
                                                    const unsigned long long *smask, float *parena) {
  using S = ConsumerSet<CS>;
  using AL = ArenaLds;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = 1.0f / mp.dx;
  const float kscale = -mp.dt * (4.f * dxi * dxi);
  const unsigned long long lt = lanemask_lt();
  float acc[27][S::NA];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int q = 0; q < S::NA; ++q) acc[k][q] = 0.f;
  for (int k = (int)threadIdx.x - 256; k < 7 * AL::CH; k += 256) parena[k] = 0.f;  // the four consumer waves clear the bin's arena
  __syncthreads();  // (the producers fill the velocity arena meanwhile)
  int r = 0, off = 0;  // next round to consume, entry number of its first particle
  for (int it = 0; it <= nchunks; ++it) {
    if (it > 0) {
      const int produced = 256 * it < total ? 256 * it : total;
#pragma unroll 1
      while (off < total) {
        const bool has = (mask >> r) & 1u;
        const unsigned long long occ = __ballot(has);
        const int cnt = __popcll(occ);
        if (off + cnt > produced) break;  // the round's last entries belong to the chunk in production
        if (has) {
          const int e = off + __popcll(occ & lt);
          const int grp = (e >> 6) % SL_NG, pos = e & 63;
          if ((smask[grp] >> pos) & 1ull) g2p2g_consume_set<CS>(mp, stage, grp * (G2P2G_NF * 64) + pos, kscale, acc);
        }
        off += cnt;
        ++r;
      }
    }
    __syncthreads();
  }
  // the set's channels of the bin's arena belong to this wave alone; phases ordered inside the wave (see g2p2g_body)
  float *a0 = parena + (size_t)S::CH0 * AL::CH + AL::at(cx, cy, cz);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
    for (int q = 0; q < S::NA; ++q) g[q * AL::CH] += acc[k][q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

template <int SIDE, int SMODEL, bool WRITE_ALL>
static __global__ __launch_bounds__(512, 4) void g2p2g_slot_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, SlotArgs A) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float varena[3 * AL::CH];
  __shared__ float parena[7 * AL::CH];
  __shared__ float stage[SL_NG * G2P2G_NF * 64];
  __shared__ unsigned long long smask[SL_NG];
  __shared__ unsigned short tab[SL_KMAX * 64];  // entry -> round * 64 + cell
  __shared__ unsigned clrAll[64];
  __shared__ int outCount;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bin = blockIdx.x + A.binBase;
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
      if ((r & 7) == w && has) tab[total + __popcll(occ & lt)] = (unsigned short)(r * 64 + lane);
      total += __popcll(occ);
    }
  }
  const int nchunks = (total + 255) >> 8;
  if (tid < 64) clrAll[tid] = 0u;
  if (tid == 0) outCount = 0;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  __syncthreads();  // the table is complete
  if (w == 0) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 0>(mp, ps, geo, bin, total, lane, nchunks, varena, stage, smask, tab, &outCount, clrAll, A);
  else if (w == 1) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 1>(mp, ps, geo, bin, total, lane, nchunks, varena, stage, smask, tab, &outCount, clrAll, A);
  else if (w == 2) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 2>(mp, ps, geo, bin, total, lane, nchunks, varena, stage, smask, tab, &outCount, clrAll, A);
  else if (w == 3) g2p2g_slot_producer<SIDE, SMODEL, WRITE_ALL, 3>(mp, ps, geo, bin, total, lane, nchunks, varena, stage, smask, tab, &outCount, clrAll, A);
  else if (w == 4) g2p2g_slot_consumer<0>(mp, mask, total, lane, nchunks, stage, smask, parena);
  else if (w == 5) g2p2g_slot_consumer<1>(mp, mask, total, lane, nchunks, stage, smask, parena);
  else if (w == 6) g2p2g_slot_consumer<2>(mp, mask, total, lane, nchunks, stage, smask, parena);
  else g2p2g_slot_consumer<3>(mp, mask, total, lane, nchunks, stage, smask, parena);
  __syncthreads();  // all channel sets are in the arena, every mover is in the outbox
  if (tid < 64) {
    const unsigned c = clrAll[tid];
    if (c) A.cellMask[(size_t)bin * 64 + tid] = mask & ~c;  // (tid < 64: lane == tid, `mask` is this cell's)
  }
  if (tid == 0) {
    // movers sent (the mover kernel counts the deliveries): running sums spread over SL_NCTR words -- one device-wide word serves ~90
    // atomics per microsecond, i.e. 1.5 ms for one add per bin of the 64 M-particle column
    const int oc = outCount < A.cap ? outCount : A.cap;
    A.moverCount[bin] = oc;
    if (oc) atomicAdd(&A.status[SL_SENT + (bin & (SL_NCTR - 1))], oc);
  }
  if (tid < 216) {
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = A.nbr[(size_t)geo.block * 8 + slot];
    const float *a = parena + AL::at(x, y, z);
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
}

// ------------------------------------------------------------------------------------------------------------------ mover kernel
// One workgroup (4 waves = the 4 channel sets of ConsumerSet) per destination bin.
// what channel set CS needs of a record: position, mass / v_d / C column entries (or P F^T entries)
template <int CS> struct MoverFields {
  using S = ConsumerSet<CS>;
  float pos[3], m, c0[S::NV], c1[S::NV], c2[S::NV], v[S::NV];
  __device__ __forceinline__ void load(const float *rec) {
    constexpr int cb = S::STRESS ? 26 : 17;
#pragma unroll
    for (int d = 0; d < 3; ++d) pos[d] = rec[1 + d];
    m = S::STRESS ? 0.f : rec[0];
#pragma unroll
    for (int j = 0; j < S::NV; ++j) {
      const int d = S::D0 + j;
      c0[j] = rec[cb + d];
      c1[j] = rec[cb + 3 + d];
      c2[j] = rec[cb + 6 + d];
      v[j] = S::STRESS ? 0.f : rec[14 + d];
    }
  }
};
template <int CS>
__device__ __forceinline__ void mover_accumulate(const MpmDev &mp, const MoverFields<CS> &f, const int (&org)[3], int cx, int cy, int cz,
                                                 float kscale, float (&acc)[27][ConsumerSet<CS>::NA]) {
  using S = ConsumerSet<CS>;
  const float dxi = 1.0f / mp.dx;
  Arena ar;
  const int cc[3] = {cx, cy, cz};
#pragma unroll
  for (int d = 0; d < 3; ++d) {  // the record's position lies in this lane's cell: same arithmetic as the main kernel's staging
    const float X = f.pos[d] * dxi;
    const float fl = (float)(org[d] + cc[d]);
    const float d0 = X - fl;
    ar.w[d][0] = 0.5f * (1.5f - d0) * (1.5f - d0);
    const float d1 = d0 - 1.0f;
    ar.w[d][1] = 0.75f - d1 * d1;
    const float zz = 0.5f + d1;
    ar.w[d][2] = 0.5f * zz * zz;
    ar.lp[d] = d0 * mp.dx;
  }
  const float scale = S::STRESS ? kscale : f.m;
  float wzs[3], Pz[S::NV][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) wzs[k] = ar.w[2][k] * scale;
#pragma unroll
  for (int j = 0; j < S::NV; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) Pz[j][k] = fmaf(f.c2[j], (float)k * mp.dx - ar.lp[2], f.v[j]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float x0 = (float)a * mp.dx - ar.lp[0];
    float Pxa[S::NV];
#pragma unroll
    for (int j = 0; j < S::NV; ++j) Pxa[j] = f.c0[j] * x0;
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float x1 = (float)bb * mp.dx - ar.lp[1];
      const float wxy = ar.w[0][a] * ar.w[1][bb];
      float q[S::NV];
#pragma unroll
      for (int j = 0; j < S::NV; ++j) q[j] = fmaf(f.c1[j], x1, Pxa[j]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float Ws = wxy * wzs[c];
        auto &Ac = acc[(a * 3 + bb) * 3 + c];
        if constexpr (S::MASS) Ac[0] += Ws;
#pragma unroll
        for (int j = 0; j < S::NV; ++j) Ac[(S::MASS ? 1 : 0) + j] = fmaf(Ws, q[j] + Pz[j][c], Ac[(S::MASS ? 1 : 0) + j]);
      }
    }
  }
}

// wave CS of a destination bin: its channel set of every arrival; the lightest set (CS == 3: one channel) also gives the arrivals
// their new home (lowest free round of the cell that was free BEFORE this step's departures).  Records are fetched one round ahead.
template <int SIDE, int CS, bool FLUID, bool DP, bool WRITE_ALL>
__device__ __forceinline__ void mover_role(const MpmDev &mp, const ParticlesDev &ps, const SlotArgs &A, int bin, const int (&org)[3], int lane,
                                           int rounds, const int *inboxCount, const int (*inbox)[SL_MAXIN], float *parena) {
  using S = ConsumerSet<CS>;
  using AL = ArenaLds;
  constexpr int LW = 64;
  constexpr bool REHOME = CS == 3;
  constexpr int NH = WRITE_ALL ? 35 : 14;  // floats of the record the new home needs: m, x, F, logJp (, v, C, P F^T)
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = 1.0f / mp.dx;
  const float kscale = -mp.dt * (4.f * dxi * dxi);
  float acc[27][S::NA];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int q = 0; q < S::NA; ++q) acc[k][q] = 0.f;
  const int mine = inboxCount[lane] < SL_MAXIN ? inboxCount[lane] : SL_MAXIN;
  unsigned m = 0u;
  if constexpr (REHOME) m = A.cellMask[(size_t)bin * 64 + lane];
  MoverFields<CS> cur, nxt;
  float hcur[REHOME ? NH : 1], hnxt[REHOME ? NH : 1];
  auto fetch = [&](int r, MoverFields<CS> &f, float (&h)[REHOME ? NH : 1]) {
    const float *rec = A.moverRec + (size_t)inbox[lane][r] * SL_REC;
    f.load(rec);
    if constexpr (REHOME) {
#pragma unroll
      for (int d = 0; d < NH; ++d) h[d] = rec[d];
    }
  };
  if (mine > 0) fetch(0, cur, hcur);
  for (int r = 0; r < rounds; ++r) {
    if (r + 1 < mine) fetch(r + 1, nxt, hnxt);
    if (r < mine) {
      mover_accumulate<CS>(mp, cur, org, cx, cy, cz, kscale, acc);
      if constexpr (REHOME) {
        const unsigned freeBits = ~m & (A.K >= 32 ? 0xffffffffu : ((1u << A.K) - 1u));
        if (freeBits == 0u) {
          A.status[1] = 1;  // cell full
        } else {
          const int rr = __ffs((int)freeBits) - 1;
          m |= 1u << rr;
          const size_t i = ((size_t)bin * (size_t)A.K + (size_t)rr) * 64 + (size_t)lane;
          const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
          float x[3], F[9];
#pragma unroll
          for (int d = 0; d < 3; ++d) x[d] = hcur[1 + d];
#pragma unroll
          for (int d = 0; d < 9; ++d) F[d] = hcur[4 + d];
          pstore1<LW>(ps.mass, o, hcur[0]);
          pstore<LW, 3>(ps.pos, o, x);
          pstore_state<LW, FLUID>(ps.F, o, F);
          if constexpr (DP) pstore1<LW>(ps.logJp, o, hcur[13]);
          if constexpr (WRITE_ALL) {
            float v[3], C[9], PF[9];
#pragma unroll
            for (int d = 0; d < 3; ++d) v[d] = hcur[14 + d];
#pragma unroll
            for (int d = 0; d < 9; ++d) { C[d] = hcur[17 + d]; PF[d] = hcur[26 + d]; }
            pstore<LW, 3>(ps.vel, o, v);
            pstore<LW, 9>(ps.C, o, C);
            pstore<LW, 9>(ps.stress, o, PF);
          }
        }
      }
    }
    cur = nxt;
    if constexpr (REHOME) {
#pragma unroll
      for (int d = 0; d < NH; ++d) hcur[d] = hnxt[d];
    }
  }
  if constexpr (REHOME) {
    if (mine > 0) A.cellMask[(size_t)bin * 64 + lane] = m;
  }
  float *a0 = parena + (size_t)S::CH0 * AL::CH + AL::at(cx, cy, cz);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
    for (int q = 0; q < S::NA; ++q) g[q * AL::CH] += acc[k][q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

template <int SIDE, bool FLUID, bool DP, bool WRITE_ALL>
static __global__ __launch_bounds__(256) void mover_pull_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, SlotArgs A) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  constexpr int BPB = bins_per_block<SIDE>();
  __shared__ float parena[7 * AL::CH];
  __shared__ int inboxCount[64];
  __shared__ int inbox[64][SL_MAXIN];
  __shared__ int srcBin[27];
  __shared__ int srcCnt[27];
  __shared__ int srcOff[28];
  __shared__ int total;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bin = blockIdx.x + A.binBase;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  if (tid < 64) inboxCount[tid] = 0;
  if (tid == 0) total = 0;
  if (tid < 27) {
    const int sb = neighbour_bin<SIDE>(A.nbr27, geo.block, bin, tid);
    int c = 0;
    if (sb >= 0) c = A.moverCount[sb];
    srcBin[tid] = sb;
    srcCnt[tid] = c;
    if (c) atomicAdd(&total, c);
  }
  __syncthreads();
  if (total == 0) return;  // nothing addressed to anybody around here (uniform)
  // every record of the 27 outboxes is looked at by one thread; all destination loads of the workgroup are in flight together
  // (a loop over the outboxes pays 27 memory latencies in a row: 2.8 -> 2.0 ms; fetching the first 32 destinations of every
  // outbox together with its count, to save the second latency, costs more in wasted reads than it gains: 2.25 ms)
  if (tid == 0) {
    int o = 0;
    for (int k = 0; k < 27; ++k) {
      srcOff[k] = o;
      o += srcCnt[k];
    }
    srcOff[27] = o;
  }
  __syncthreads();
  const int tot = srcOff[27];
  for (int tt = tid; tt < tot; tt += 256) {
    int sidx = 0;
    while (tt >= srcOff[sidx + 1]) ++sidx;
    const int k = tt - srcOff[sidx], sb = srcBin[sidx];
    int x, y, z;
    unpack_cell(A.moverDest[(size_t)sb * A.cap + (size_t)k], x, y, z);
    const int rx = x - geo.org[0], ry = y - geo.org[1], rz = z - geo.org[2];
    if ((unsigned)rx < 4u && (unsigned)ry < 4u && (unsigned)rz < 4u) {
      const int l2 = (rx * 4 + ry) * 4 + rz;
      const int slot = atomicAdd(&inboxCount[l2], 1);
      if (slot < SL_MAXIN) inbox[l2][slot] = sb * A.cap + k;
      else A.status[3] = 1;
    }
  }
  __syncthreads();
  int rounds = inboxCount[lane];
  int got = rounds < SL_MAXIN ? rounds : SL_MAXIN;
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const int o2 = __shfl_xor(rounds, sft, 64);
    rounds = o2 > rounds ? o2 : rounds;
    got += __shfl_xor(got, sft, 64);
  }
  if (rounds > SL_MAXIN) rounds = SL_MAXIN;
  if (rounds == 0) return;  // nothing addressed to this bin (uniform)
  if (tid == 0) atomicAdd(&A.status[SL_DELIVERED + (bin & (SL_NCTR - 1))], got);
  for (int k = tid; k < 7 * AL::CH; k += 256) parena[k] = 0.f;
  __syncthreads();
  if (w == 0) mover_role<SIDE, 0, FLUID, DP, WRITE_ALL>(mp, ps, A, bin, geo.org, lane, rounds, inboxCount, inbox, parena);
  else if (w == 1) mover_role<SIDE, 1, FLUID, DP, WRITE_ALL>(mp, ps, A, bin, geo.org, lane, rounds, inboxCount, inbox, parena);
  else if (w == 2) mover_role<SIDE, 2, FLUID, DP, WRITE_ALL>(mp, ps, A, bin, geo.org, lane, rounds, inboxCount, inbox, parena);
  else mover_role<SIDE, 3, FLUID, DP, WRITE_ALL>(mp, ps, A, bin, geo.org, lane, rounds, inboxCount, inbox, parena);
  __syncthreads();
  if (tid < 216) {
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = A.nbr[(size_t)geo.block * 8 + slot];
    const float *a = parena + AL::at(x, y, z);
    if (bn >= 0) {
      float *g = A.gridB + (size_t)bn * 7 * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        const float v = a[ch * AL::CH];
        if (v != 0.f) unsafeAtomicAdd(g + ch * NC, v);
      }
    } else if (a[0] != 0.f) {
      A.status[2] = 1;
    }
  }
}

}  // namespace zsr

using namespace zsr;

extern "C" {

size_t zs_rocm_mpm_slot_outbox_bytes(size_t nbins, int cap, int which) {
  if (which == 0) return nbins * sizeof(int);                              // moverCount
  if (which == 1) return nbins * (size_t)cap * sizeof(long long);          // moverDest
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
// gridB zeroed by the caller; outbox buffers sized by zs_rocm_mpm_slot_outbox_bytes; status: 5 ints, zeroed by the caller when it
// wants to (they latch).  Returns 0, -1 on bad arguments.
int zs_rocm_mpm_g2p2g_slotted(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, const float *gridA,
                              float *gridB, size_t nblocks, unsigned *cellMask, int K, const int *nbr, const int *nbr27, int *moverCount,
                              long long *moverDest, float *moverRec, int outboxCap, int writeAll, int *status) {
  if (!nblocks) return 0;
  if (!cellMask || !nbr || !nbr27 || !moverCount || !moverDest || !moverRec || !status || K < 1 || K > 32 || outboxCap < 1) return -1;
  if (p->model < ZS_MPM_FIXED_COROTATED || p->model > ZS_MPM_EQUATION_OF_STATE) return -1;
  if (uniform_lane_width(ps, model_uses_logjp(p->model), writeAll != 0) != 64 || (writeAll && (!ps.vel.base || !ps.C.base))) {
    fprintf(stderr, "[zs_rocm] g2p2g_slotted needs all particle attributes in one TileVector<f32, 64>\n");
    return -1;
  }
  Launch L(pol, "G2P2GTransfer(slotted)");
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  const unsigned bpb = p->side == 4 ? 1u : 8u;
  const unsigned nbins = (unsigned)(nblocks * bpb);
  const SlotArgs A{gridA, gridB, cellMask, K, nbr, nbr27, moverCount, moverDest, moverRec, status, 0, (int)nbins, outboxCap};
#define CALL_SLOT3(SS, M, WA)                                                                                          \
  hipLaunchKernelGGL((g2p2g_slot_kernel<SS, M, WA>), dim3(nbins), dim3(512), 0, L.stream, mp, pd, t, A);                \
  hipLaunchKernelGGL((mover_pull_kernel<SS, model_is_fluid(M), model_uses_logjp(M), WA>), dim3(nbins), dim3(256), 0, L.stream, mp, pd, t, A)
#define CALL_SLOT(SS, M)                       \
  do {                                         \
    if (writeAll) { CALL_SLOT3(SS, M, true); } \
    else { CALL_SLOT3(SS, M, false); }         \
  } while (0)
  ZSR_DISPATCH_SIDE_PURE(p->side, p->model, CALL_SLOT);
  return 0;
}

}  // extern "C"
