// mpm_c2.hip -- the gather-style transfers: P2C2GTransfer / P2C2GTransferMomentum / P2C2GTransferForce
// (simulation/transfer/P2C2G.hpp:53-189, :346-439, :547-679) and PreG2C2P / G2C2P / PostG2C2P (simulation/transfer/G2C2P.hpp:59-135,
// :208-275).  Linear particle <-> cell-centre weights, 1/8 cell <-> node weights.
//
// The reference runs one functor over Collapse{nblocks, side^3}: per cell it walks the 27 IndexBuckets around the cell, evaluates the
// constitutive model of every particle in range (again for each of the up to 8 cells that see the particle) and then does float
// atomics into the 8 nodes (P2C2G) or into the particles (G2C2P).  Here every stage is a gather, so no float atomic is issued and a run
// is reproducible bit for bit:
//   P2C2G  0. per bucket: order the bucket's particles by octant                                     (c2_octant_kernel)
//          1. per particle: constitutive update ONCE, 64-byte record {pos, mass, Q, mass*vel} in bucket order (c2_particle_kernel)
//          2. per 2x2x2 cells: walk the 4x4x4 buckets around them, sum the 16 moments of each cell     (p2c2g_cell8_kernel)
//          3. per node: sum the 8 cells around the node, add to the grid                               (p2c2g_node_kernel)
//   G2C2P  1. per cell: v_c and v_c (x) x_i from the 8 nodes                                            (g2c2p_cell_kernel)
//          2. per particle: sum over the cells in range, add to v_p / B_p                              (g2c2p_particle_kernel)
// Evaluating the plastic models once per particle also removes the reference's order dependence (its functor stores logJp up to 8
// times per particle and later cells read the updated value).
#include "mpm_device.hpp"
#include "hashtable.hpp"

using namespace zsr;

namespace {

constexpr int C2_TRANSFER = 0, C2_MOMENTUM = 1, C2_FORCE = 2;
#ifndef ZS_C2_CXN
#  define ZS_C2_CXN 2  // 2 x 2 x 2 cells per lane (1: 1 x 2 x 2 -- measured 5 % slower)
#endif

__device__ __forceinline__ float c2_dinv(float x, float dx, float dxi) {
  const float r = x - (float)(int)floorf(x * dxi + 0.5f) * dx;  // P2C2G.hpp:88
  return 2.f / (dx * dx - 2 * r * r);                            // :89
}

// ---- P2C2G stage 0: per bucket, order the bucket's particles by the octant of the cell they sit in (stable: ascending id inside an
// octant).  A cell at offset -1 / +1 from a bucket along an axis can only be reached by the particles in the upper / lower half of the
// bucket along that axis (|x_p - x_c| <= dx), so the cell kernel walks 64 instead of 216 candidates per cell.  sub[b]: byte k = number of
// particles of the bucket in octants < k (code = 4 hx + 2 hy + hz); ~0 = not ordered (more than 255 particles in the bucket, or buckets that are
// not the cells of this grid: then every bucket is walked whole and only the range check decides).
__device__ __forceinline__ int c2_octant(const Port<float> &pos, size_t i, float dxi) {
  float p[3];
  load_attr<3>(pos, i, p);
  int code = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float X = p[d] * dxi;
    code = code * 2 + (X - floorf(X) >= 0.5f ? 1 : 0);
  }
  return code;
}
__global__ __launch_bounds__(256) void c2_octant_kernel(Port<float> pos, float dxi, const int *offsets, const int *indices, int nbuckets,
                                                        int canOrder, int *slotOf, unsigned long long *sub) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  const int st = offsets[b], ed = offsets[b + 1];
  if (!canOrder || ed - st > 255) {
    for (int k = st; k < ed; ++k) slotOf[indices[k]] = k;
    sub[b] = ~0ull;
    return;
  }
  unsigned long long counts = 0, codes = 0;  // the octants of the first 21 particles stay in a register: one scattered read each
  for (int k = st; k < ed; ++k) {
    const int code = c2_octant(pos, (size_t)indices[k], dxi);
    counts += 1ull << (8 * code);
    if (k - st < 21) codes |= (unsigned long long)code << (3 * (k - st));
  }
  const unsigned long long excl = (counts * 0x0101010101010101ull) << 8;  // bytewise exclusive prefix sum (total <= 255: no carry)
  unsigned long long cur = excl;
  for (int k = st; k < ed; ++k) {
    const int id = indices[k];
    const int sh = 8 * (k - st < 21 ? (int)((codes >> (3 * (k - st))) & 7) : c2_octant(pos, (size_t)id, dxi));
    slotOf[id] = st + (int)((cur >> sh) & 255);
    cur += 1ull << sh;
  }
  sub[b] = excl;
}

// ---- P2C2G stage 1: per particle, in bucket order (slot s of IndexBuckets::indices)
template <int MODEL, int KIND>
__global__ __launch_bounds__(256) void c2_particle_kernel(MpmDev mp, ParticlesDev ps, const int *slotOf, float4 *rec) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ps.n) return;
  // records are laid out in bucket order (a bucket's particles are one contiguous run); the particle attributes are read in storage
  // order (coalesced AoSoA rows) and the 64-byte record is the scattered access
  const size_t slot = (size_t)slotOf[i];
  const float dx = mp.dx, dxi = mp.dxi;
  float pos[3], vel[3] = {0.f, 0.f, 0.f}, C[9], Q[9], Dinv[3];
  load_attr<3>(ps.pos, i, pos);
  load_attr<9>(ps.C, i, C);
  const float mass = ps.mass.base[ps.mass.off(i)];
  load_attr<3>(ps.vel, i, vel);
#pragma unroll
  for (int d = 0; d < 3; ++d) Dinv[d] = c2_dinv(pos[d], dx, dxi);
#pragma unroll
  for (int d = 0; d < 9; ++d) C[d] *= Dinv[d / 3];
  if constexpr (KIND == C2_MOMENTUM) {
#pragma unroll
    for (int d = 0; d < 9; ++d) Q[d] = C[d] * mass;  // P2C2G.hpp:391
  } else {
    float F[9];
    load_state<model_is_fluid(MODEL)>(ps.F, i, F);
    float lj = 0.f;
    if constexpr (model_uses_logjp(MODEL)) lj = ps.logJp.base[ps.logJp.off(i)];
    model_stress<MODEL>(mp.mat, lj, F, Q, C);
    if constexpr (model_uses_logjp(MODEL)) ps.logJp.base[ps.logJp.off(i)] = lj;  // P2C2G.hpp:140
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      Q[d] *= Dinv[d / 3] * -mp.dt;                            // :144
      if constexpr (KIND == C2_TRANSFER) Q[d] += C[d] * mass;  // :146
    }
  }
  float4 *r = rec + 4 * slot;
  r[0] = make_float4(pos[0], pos[1], pos[2], mass);
  r[1] = make_float4(Q[0], Q[1], Q[2], Q[3]);
  r[2] = make_float4(Q[4], Q[5], Q[6], Q[7]);
  r[3] = make_float4(Q[8], mass * vel[0], mass * vel[1], mass * vel[2]);
}

// block key (in the table's convention) and cell coordinate of cell `cell` of block b
template <int SIDE> __device__ __forceinline__ void c2_cell_coord(const BhtDev &t, int b, int cell, int kscale, int (&coord)[3]) {
  const int cs = SIDE / kscale;  // keys are block coordinates (kscale 1) or block origins in cells (kscale SIDE)
  coord[0] = t.activeKeys[3 * (size_t)b] * cs + cell / (SIDE * SIDE);
  coord[1] = t.activeKeys[3 * (size_t)b + 1] * cs + (cell / SIDE) % SIDE;
  coord[2] = t.activeKeys[3 * (size_t)b + 2] * cs + cell % SIDE;
}

// bucket number of a cell: a probe of the IndexBuckets' hash table, or -- buckets built over the partition itself
// (zs_rocm_index_buckets_for_partition) -- the cell's position in the grid: block * side^3 + cell id
struct C2Buckets {
  HtDev ht;
  int dense;
};
template <int SIDE> __device__ __forceinline__ int c2_bucket_no(const C2Buckets &bk, const BhtDev &t, int kscale, const int (&bc)[3]) {
  if (!bk.dense) return ht_query<3>(bk.ht, bc);
  int loc[3], key[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    loc[d] = bc[d] & (SIDE - 1);
    key[d] = (bc[d] - loc[d]) / SIDE * kscale;
  }
  const int b = bht_query<3>(t, key);
  return b < 0 ? -1 : b * (SIDE * SIDE * SIDE) + (loc[0] * SIDE + loc[1]) * SIDE + loc[2];
}

// ---- P2C2G stage 2: per cell, the 16 moments m_c, mv_c, Q_c, (Q x_p)_c  (P2C2G.hpp:66-163); sums[b][16][NC].  One workgroup per
// block: the bucket ranges of the (SIDE+2)^3 cells around the block are looked up once (one hash probe per halo cell instead of 27 per
// cell) and kept in LDS together with the octant offsets of stage 0.
template <int SIDE, int KIND>
__global__ __launch_bounds__(256) void p2c2g_cell_kernel(MpmDev mp, BhtDev t, C2Buckets buckets, const int *offsets,
                                                         const unsigned long long *sub, const float4 *rec, float *sums) {
  constexpr int NC = SIDE * SIDE * SIDE, H = SIDE + 2, NH = H * H * H;
  __shared__ int2 range[NH];
  __shared__ unsigned long long octs[NH];
  const int b = blockIdx.x;
  const float dx = mp.dx, dxi = mp.dxi;
  int org[3];
  c2_cell_coord<SIDE>(t, b, 0, mp.kscale, org);
  for (int h = threadIdx.x; h < NH; h += blockDim.x) {
    const int bc[3] = {org[0] - 1 + h / (H * H), org[1] - 1 + (h / H) % H, org[2] - 1 + h % H};
    const int bno = c2_bucket_no<SIDE>(buckets, t, mp.kscale, bc);
    range[h] = bno < 0 ? make_int2(0, 0) : make_int2(offsets[bno], offsets[bno + 1] - offsets[bno]);  // {start, count}
    octs[h] = bno < 0 ? 0ull : sub[bno];
  }
  __syncthreads();
  for (int cell = threadIdx.x; cell < NC; cell += blockDim.x) {
    const int lx = cell / (SIDE * SIDE), ly = (cell / SIDE) % SIDE, lz = cell % SIDE;
    const float pc0 = ((float)(org[0] + lx) + 0.5f) * dx, pc1 = ((float)(org[1] + ly) + 0.5f) * dx, pc2 = ((float)(org[2] + lz) + 0.5f) * dx;
    float m_c = 0.f, mv[3] = {0.f, 0.f, 0.f}, Qc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, QX[3] = {0.f, 0.f, 0.f};
    for (int o = 0; o < 27; ++o) {  // ndrange<3>(3) order
      const int ox = o / 9, oy = (o / 3) % 3, oz = o % 3;
      const int h = ((lx + ox) * H + ly + oy) * H + lz + oz;
      const int2 sc = range[h];
      if (!sc.y) continue;
      const unsigned long long oc = octs[h];
      const bool ordered = oc != ~0ull;
      // half-cells of the bucket that can reach this cell: offset -1 -> upper half only, +1 -> lower half only
      const int xlo = ox == 0, xhi = ox != 2, ylo = oy == 0, yhi = oy != 2, zlo = oz == 0, zhi = oz != 2;
      for (int hx = xlo; hx <= xhi; ++hx)
        for (int hy = ylo; hy <= yhi; ++hy) {
          int a = 0, e = sc.y;
          if (ordered) {
            const int clo = hx * 4 + hy * 2 + zlo, chi = hx * 4 + hy * 2 + zhi;
            a = (int)((oc >> (8 * clo)) & 255);
            if (chi != 7) e = (int)((oc >> (8 * chi + 8)) & 255);
          } else if (hx != xlo || hy != ylo) {
            continue;  // an unordered bucket is walked once, whole
          }
          for (int st = sc.x + a, ed = sc.x + e; st < ed; ++st) {
            const float4 *r = rec + 4 * (size_t)st;
            const float4 r0 = r[0];
            const float d0 = pc0 - r0.x, d1 = pc1 - r0.y, d2 = pc2 - r0.z;
            if (fabsf(d0) > dx || fabsf(d1) > dx || fabsf(d2) > dx) continue;  // checkInKernelRange, :72-76
            const float4 r1 = r[1], r2 = r[2], r3 = r[3];
            const float a0 = fabsf(d0 * dxi), a1 = fabsf(d1 * dxi), a2 = fabsf(d2 * dxi);
            float W = 1.f;
            if constexpr (KIND == C2_TRANSFER) {  // :149-151
              W *= 1.f - a0; W *= 1.f - a1; W *= 1.f - a2;
            } else {  // :396-402, :649-655
              W *= a0 <= 1 ? 1.f - a0 : 0.f; W *= a1 <= 1 ? 1.f - a1 : 0.f; W *= a2 <= 1 ? 1.f - a2 : 0.f;
            }
            const float Q[9] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x};
            // all 16 moments for every kind (the Force variant's node stage ignores m_c, mv_c): when that variant read only part of
            // the record the compiler re-sliced the four float4 loads into 16-byte loads at offsets 12 / 28 plus a late dependent
            // dword, and the kernel ran 2x slower than the variants that use everything
            m_c += r0.w * W;
            mv[0] += r3.y * W; mv[1] += r3.z * W; mv[2] += r3.w * W;
#pragma unroll
            for (int d = 0; d < 3; ++d) QX[d] += (Q[d] * r0.x + Q[3 + d] * r0.y + Q[6 + d] * r0.z) * W;
#pragma unroll
            for (int d = 0; d < 9; ++d) Qc[d] += Q[d] * W;
          }
        }
    }
    float *s = sums + (size_t)b * 16 * NC + cell;
    s[0] = m_c;
#pragma unroll
    for (int d = 0; d < 3; ++d) s[(1 + d) * NC] = mv[d];
#pragma unroll
    for (int d = 0; d < 9; ++d) s[(4 + d) * NC] = Qc[d];
#pragma unroll
    for (int d = 0; d < 3; ++d) s[(13 + d) * NC] = QX[d];
  }
}

// ---- P2C2G stage 2, 2x2x2 cells per lane: the 8 cells of a lane need the particles of a 3x3x3-cell volume (27 x 8 records instead of
// 8 x 64: the cell-per-lane kernel above leaves L2 with 8-16 GB per launch because the 8 cells that need a record run at different
// times in different waves).  The record is loaded once and applied to whichever of the lane's 8 cells it is in range of.  One wave per
// 8^3 block (8 blocks of 4^3 cells per wave).
template <int SIDE, int KIND, int CXN>
__global__ __launch_bounds__(128) void p2c2g_cell8_kernel(MpmDev mp, BhtDev t, C2Buckets buckets, const int *offsets,
                                                         const unsigned long long *sub, const float4 *rec, float *sums, int nblocks) {
  constexpr int NC = SIDE * SIDE * SIDE, H = SIDE + 2, NH = H * H * H, GS = SIDE / 2, G = (SIDE / CXN) * GS * GS;
  constexpr int TPB = SIDE == 8 ? G : 64, NB = TPB / G;  // CXN x 2 x 2 cells per lane; threads per workgroup, blocks per workgroup
  __shared__ int2 range[NB * NH];
  __shared__ unsigned long long octs[NB * NH];
  const int b0 = blockIdx.x * NB;
  const float dx = mp.dx, dxi = mp.dxi;
  for (int h = threadIdx.x; h < NB * NH; h += TPB) {
    const int b = b0 + h / NH, hh = h % NH;
    int2 r = make_int2(0, 0);
    unsigned long long oc = 0ull;
    if (b < nblocks) {
      int o3[3];
      c2_cell_coord<SIDE>(t, b, 0, mp.kscale, o3);
      const int bc[3] = {o3[0] - 1 + hh / (H * H), o3[1] - 1 + (hh / H) % H, o3[2] - 1 + hh % H};
      const int bno = c2_bucket_no<SIDE>(buckets, t, mp.kscale, bc);
      if (bno >= 0) {
        r = make_int2(offsets[bno], offsets[bno + 1] - offsets[bno]);
        oc = sub[bno];
      }
    }
    range[h] = r;
    octs[h] = oc;
  }
  __syncthreads();
  const int bi = threadIdx.x / G, grp = threadIdx.x % G, b = b0 + bi;
  if (b >= nblocks) return;
  const int g3[3] = {CXN * (grp / (GS * GS)), 2 * ((grp / GS) % GS), 2 * (grp % GS)};  // local coordinate of the lane's first cell
  int org[3];
  c2_cell_coord<SIDE>(t, b, 0, mp.kscale, org);
  float pc[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j) pc[d][j] = ((float)(org[d] + g3[d] + j) + 0.5f) * dx;
  float acc[4 * CXN][16];
#pragma unroll
  for (int c = 0; c < 4 * CXN; ++c)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[c][k] = 0.f;
  for (int bb = 0; bb < 16 * (CXN + 2); ++bb) {  // buckets (first cell - 1) + {0..CXN+1} x {0..3}^2
    const int bx = bb >> 4, by = (bb >> 2) & 3, bz = bb & 3;
    const int h = bi * NH + ((g3[0] + bx) * H + g3[1] + by) * H + g3[2] + bz;
    const int2 sc = range[h];
    if (!sc.y) continue;
    const unsigned long long oc = octs[h];
    const bool ordered = oc != ~0ull;
    // the lane's cells reach the upper half of bucket 0 and the lower half of bucket 3 only
    const int xlo = bx == 0, xhi = bx != CXN + 1, ylo = by == 0, yhi = by != 3, zlo = bz == 0, zhi = bz != 3;
    for (int hx = xlo; hx <= xhi; ++hx)
      for (int hy = ylo; hy <= yhi; ++hy) {
        int a = 0, e = sc.y;
        if (ordered) {
          const int clo = hx * 4 + hy * 2 + zlo, chi = hx * 4 + hy * 2 + zhi;
          a = (int)((oc >> (8 * clo)) & 255);
          if (chi != 7) e = (int)((oc >> (8 * chi + 8)) & 255);
        } else if (hx != xlo || hy != ylo) {
          continue;
        }
        for (int st = sc.x + a, ed = sc.x + e; st < ed; ++st) {
          const float4 *r = rec + 4 * (size_t)st;
          const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
          const float x3[3] = {r0.x, r0.y, r0.z};
          float w[3][2];
          bool in[3][2];
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float dd = pc[d][j] - x3[d];
              in[d][j] = !(fabsf(dd) > dx);  // checkInKernelRange, :72-76
              const float aa = fabsf(dd * dxi);
              w[d][j] = KIND == C2_TRANSFER ? 1.f - aa : (aa <= 1 ? 1.f - aa : 0.f);  // :149-151 / :396-402, :649-655
            }
          const float Q[9] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x};
          float qx[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) qx[d] = Q[d] * r0.x + Q[3 + d] * r0.y + Q[6 + d] * r0.z;
#pragma unroll
          for (int c = 0; c < 4 * CXN; ++c) {
            const int i = c >> 2, j = (c >> 1) & 1, k = c & 1;
            if (in[0][i] && in[1][j] && in[2][k]) {
              float W = 1.f;
              W *= w[0][i]; W *= w[1][j]; W *= w[2][k];
              acc[c][0] += r0.w * W;
              acc[c][1] += r3.y * W; acc[c][2] += r3.z * W; acc[c][3] += r3.w * W;
#pragma unroll
              for (int d = 0; d < 9; ++d) acc[c][4 + d] += Q[d] * W;
#pragma unroll
              for (int d = 0; d < 3; ++d) acc[c][13 + d] += qx[d] * W;
            }
          }
        }
      }
  }
#pragma unroll
  for (int c = 0; c < 4 * CXN; ++c) {
    const int lx = g3[0] + (c >> 2), ly = g3[1] + ((c >> 1) & 1), lz = g3[2] + (c & 1);
    float *sp = sums + (size_t)b * 16 * NC + ((lx * SIDE + ly) * SIDE + lz);
#pragma unroll
    for (int k = 0; k < 16; ++k) sp[k * NC] = acc[c][k];
  }
}

// ---- P2C2G stage 3: per node, the 8 cells node - {0,1}^3  (the gather form of :166-187).  One workgroup per block.
template <int SIDE, int KIND>
__global__ __launch_bounds__(256) void p2c2g_node_kernel(MpmDev mp, BhtDev t, const float *sums, float *grid) {
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ int nb[8];  // block number of key - {0,1}^3, x-major
  const int b = blockIdx.x;
  if (threadIdx.x < 8) {
    const int o = threadIdx.x;
    int k[3] = {t.activeKeys[3 * (size_t)b] - (o >> 2) * mp.kscale, t.activeKeys[3 * (size_t)b + 1] - ((o >> 1) & 1) * mp.kscale,
                t.activeKeys[3 * (size_t)b + 2] - (o & 1) * mp.kscale};
    nb[o] = o ? bht_query<3>(t, k) : b;
  }
  __syncthreads();
  const float dx = mp.dx;
  for (int cell = threadIdx.x; cell < NC; cell += blockDim.x) {
    int ci[3];
    c2_cell_coord<SIDE>(t, b, cell, mp.kscale, ci);
    const float p0 = (float)ci[0] * dx, p1 = (float)ci[1] * dx, p2 = (float)ci[2] * dx;  // posi, :171
    const int lx = cell / (SIDE * SIDE), ly = (cell / SIDE) % SIDE, lz = cell % SIDE;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 7; o >= 0; --o) {  // cell = node - (o bits): ascending cell coordinate
      const int x = lx - (o >> 2), y = ly - ((o >> 1) & 1), z = lz - (o & 1);
      const int w = ((x < 0) << 2) | ((y < 0) << 1) | (z < 0);
      const int bn = nb[w];
      if (bn < 0) continue;  // that cell's block is not in the partition: the reference's launch range has no such cell
      const float *s = sums + (size_t)bn * 16 * NC + (((x & (SIDE - 1)) * SIDE + (y & (SIDE - 1))) * SIDE + (z & (SIDE - 1)));
      constexpr float Wci = 1.f / 8;
      if constexpr (KIND != C2_FORCE) acc[0] += s[0] * Wci;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float mvd = KIND != C2_FORCE ? s[(1 + d) * NC] : 0.f;
        const float lin = (s[(4 + d) * NC] * p0 + s[(7 + d) * NC] * p1 + s[(10 + d) * NC] * p2) - s[(13 + d) * NC];
        acc[1 + d] += (KIND != C2_FORCE ? mvd + lin : lin) * Wci;
      }
    }
    float *gp = grid + (size_t)b * 7 * NC + cell;
    if constexpr (KIND != C2_FORCE) gp[0] += acc[0];
#pragma unroll
    for (int d = 0; d < 3; ++d) gp[(1 + d) * NC] += acc[1 + d];
  }
}

// ---- G2C2P stage 1: per cell v_c, v_c (x) x_i from the nodes cell + {0,1}^3  (G2C2P.hpp:69-90); cv[b][NC][12]
template <int SIDE> __global__ __launch_bounds__(256) void g2c2p_cell_kernel(MpmDev mp, BhtDev t, const float *grid, float *cv) {
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ int nb[8];  // block number of key + {0,1}^3
  const int b = blockIdx.x;
  if (threadIdx.x < 8) {
    const int o = threadIdx.x;
    int k[3] = {t.activeKeys[3 * (size_t)b] + (o >> 2) * mp.kscale, t.activeKeys[3 * (size_t)b + 1] + ((o >> 1) & 1) * mp.kscale,
                t.activeKeys[3 * (size_t)b + 2] + (o & 1) * mp.kscale};
    nb[o] = o ? bht_query<3>(t, k) : b;
  }
  __syncthreads();
  const float dx = mp.dx;
  for (int cell = threadIdx.x; cell < NC; cell += blockDim.x) {
    int coord[3];
    c2_cell_coord<SIDE>(t, b, cell, mp.kscale, coord);
    const int lx = cell / (SIDE * SIDE), ly = (cell / SIDE) % SIDE, lz = cell % SIDE;
    float v[3] = {0.f, 0.f, 0.f}, vx[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int x = lx + (o >> 2), y = ly + ((o >> 1) & 1), z = lz + (o & 1);
      const int w = ((x >= SIDE) << 2) | ((y >= SIDE) << 1) | (z >= SIDE);
      const int bn = nb[w];
      if (bn < 0) continue;  // :82
      const float *gp = grid + (size_t)bn * 7 * NC + (((x & (SIDE - 1)) * SIDE + (y & (SIDE - 1))) * SIDE + (z & (SIDE - 1)));
      const float posi[3] = {(float)(coord[0] + (o >> 2)) * dx, (float)(coord[1] + ((o >> 1) & 1)) * dx, (float)(coord[2] + (o & 1)) * dx};
      const float vi[3] = {gp[1 * NC], gp[2 * NC], gp[3 * NC]};
      constexpr float W = 1.f / 8;
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] += vi[d] * W;
#pragma unroll
      for (int d = 0; d < 9; ++d) vx[d] += W * vi[d % 3] * posi[d / 3];
    }
    float4 *c = (float4 *)cv + 3 * ((size_t)b * NC + cell);  // 48 contiguous bytes per cell: a particle reads 8 cells
    c[0] = make_float4(v[0], v[1], v[2], vx[0]);
    c[1] = make_float4(vx[1], vx[2], vx[3], vx[4]);
    c[2] = make_float4(vx[5], vx[6], vx[7], vx[8]);
  }
}

// ---- G2C2P stage 2: per particle, the cells whose centre is within dx  (the gather form of :92-131).  Adds to v_p and B_p like the
// reference's atomics (PreG2C2PTransfer zeroes them).  Cells floor(x/dx - 0.5) + {0,1}^3: a third cell per axis can only pass the range
// check with a weight that rounds to 0.
// STEP: PreG2C2P + G2C2P + PostG2C2P in one pass (v, B start from 0 instead of being zeroed, re-read and re-written; then C = B Dinv,
// F <- (I + dt C) F or J <- (1 + tr C dt) J, x += v dt): same bits as the three calls.
template <int SIDE, bool STEP = false, bool FLUID = false>
__global__ __launch_bounds__(256) void g2c2p_particle_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *cv) {
  constexpr int NC = SIDE * SIDE * SIDE;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ps.n) return;
  const float dx = mp.dx, dxi = mp.dxi;
  float pos[3], v[3] = {0.f, 0.f, 0.f}, B[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  load_attr<3>(ps.pos, i, pos);
  if constexpr (!STEP) {
    load_attr<3>(ps.vel, i, v);
    load_attr<9>(ps.C, i, B);
  }
  int c0[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) c0[d] = (int)floorf(pos[d] * dxi - 0.5f);
  int lastKey[3] = {0, 0, 0}, lastBlk = -2;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const int c[3] = {c0[0] + (o >> 2), c0[1] + ((o >> 1) & 1), c0[2] + (o & 1)};
    const float d0 = ((float)c[0] + 0.5f) * dx - pos[0], d1 = ((float)c[1] + 0.5f) * dx - pos[1], d2 = ((float)c[2] + 0.5f) * dx - pos[2];
    if (fabsf(d0) > dx || fabsf(d1) > dx || fabsf(d2) > dx) continue;
    int key[3], loc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      loc[d] = c[d] & (SIDE - 1);
      key[d] = (c[d] - loc[d]) / SIDE * mp.kscale;
    }
    if (lastBlk == -2 || key[0] != lastKey[0] || key[1] != lastKey[1] || key[2] != lastKey[2]) {
      lastBlk = bht_query<3>(t, key);
      lastKey[0] = key[0]; lastKey[1] = key[1]; lastKey[2] = key[2];
    }
    if (lastBlk < 0) continue;
    const float a0 = fabsf(d0 * dxi), a1 = fabsf(d1 * dxi), a2 = fabsf(d2 * dxi);
    float W = 1.f;
    W *= a0 <= 1 ? 1.f - a0 : 0.f; W *= a1 <= 1 ? 1.f - a1 : 0.f; W *= a2 <= 1 ? 1.f - a2 : 0.f;  // :109-115
    const float4 *cp = (const float4 *)cv + 3 * ((size_t)lastBlk * NC + ((loc[0] * SIDE + loc[1]) * SIDE + loc[2]));
    const float4 q0 = cp[0], q1 = cp[1], q2 = cp[2];
    const float vc[3] = {q0.x, q0.y, q0.z}, vx[9] = {q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
    for (int d = 0; d < 3; ++d) v[d] += vc[d] * W;  // :120
#pragma unroll
    for (int d = 0; d < 9; ++d) B[d] += W * (vx[d] - vc[d % 3] * pos[d / 3]);  // :122-124
  }
  store_attr<3>(ps.vel, i, v);
  store_attr<9>(ps.C, i, B);
  if constexpr (STEP) {  // PostG2C2PTransfer, G2C2P.hpp:235-270
    float C[9], oldF[9], F[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) C[d] = B[d] * c2_dinv(pos[d / 3], dx, dxi);
    load_state<FLUID>(ps.F, i, oldF);
    advance_state<FLUID>(oldF, C, mp.dt, F);
    if constexpr (FLUID) ps.F.base[ps.F.off(i)] = F[0];
    else store_attr<9>(ps.F, i, F);
#pragma unroll
    for (int d = 0; d < 3; ++d) pos[d] += v[d] * mp.dt;
    store_attr<3>(ps.pos, i, pos);
  }
}

__global__ __launch_bounds__(256) void pre_g2c2p_kernel(ParticlesDev ps) {  // G2C2P.hpp:215-218
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ps.n) return;
  const float z3[3] = {0.f, 0.f, 0.f}, z9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  store_attr<3>(ps.vel, i, z3);
  store_attr<9>(ps.C, i, z9);
}

template <bool FLUID> __global__ __launch_bounds__(256) void post_g2c2p_kernel(MpmDev mp, ParticlesDev ps) {  // G2C2P.hpp:235-270
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ps.n) return;
  const float dx = mp.dx, dxi = mp.dxi;
  float pos[3], vel[3], C[9], oldF[9], F[9];
  load_attr<3>(ps.pos, i, pos);
  load_attr<3>(ps.vel, i, vel);
  load_attr<9>(ps.C, i, C);
#pragma unroll
  for (int d = 0; d < 9; ++d) C[d] *= c2_dinv(pos[d / 3], dx, dxi);
  load_state<FLUID>(ps.F, i, oldF);
  advance_state<FLUID>(oldF, C, mp.dt, F);
  if constexpr (FLUID) ps.F.base[ps.F.off(i)] = F[0];
  else store_attr<9>(ps.F, i, F);
#pragma unroll
  for (int d = 0; d < 3; ++d) pos[d] += vel[d] * mp.dt;
  store_attr<3>(ps.pos, i, pos);
}

}  // namespace

extern "C" {

int zs_rocm_mpm_p2c2g(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_index_buckets *buckets,
                      const zs_rocm_bht_3 *tab, float *grid, size_t nblocks, int kind) {
  if (kind < C2_TRANSFER || kind > C2_FORCE || (p->side != 4 && p->side != 8)) return -1;
  if (kind != C2_MOMENTUM && (p->model < ZS_MPM_FIXED_COROTATED || p->model > ZS_MPM_EQUATION_OF_STATE)) return -1;
  if (!ps.n || !nblocks) return 0;
  if (!buckets || (!buckets->table && !buckets->dense) || (buckets->dense && buckets->denseSide != p->side) || !buckets->offsets ||
      !buckets->indices || (size_t)buckets->numEntries != ps.n) {
    fprintf(stderr, "[zs_rocm] p2c2g needs the IndexBuckets of these particles (index_buckets_for_particles, cell size dx, displacement 0)\n");
    return -1;
  }
  Launch L(pol, "P2C2GTransfer");
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  const size_t nc = (size_t)p->side * p->side * p->side;
  float4 *rec = (float4 *)L.temp(sizeof(float4) * 4 * ps.n);
  int *slotOf = (int *)L.temp(sizeof(int) * ps.n);
  // buckets over the partition: the extra bucket of the unlisted particles is ordered too (every particle needs a record slot)
  const int nbk = buckets->numBuckets + (buckets->dense ? 1 : 0);
  unsigned long long *sub = (unsigned long long *)L.temp(sizeof(unsigned long long) * (nbk + 1));
  hipLaunchKernelGGL(c2_octant_kernel, dim3(ceil_div((size_t)nbk, 256)), dim3(256), 0, L.stream, pd.pos, 1.0f / p->dx,
                     (const int *)buckets->offsets, (const int *)buckets->indices, nbk,
                     (int)(buckets->dx == p->dx && buckets->displacement == 0.f), slotOf, sub);  // other buckets: walked whole
  float *sums = (float *)L.temp(sizeof(float) * 16 * nc * nblocks);
  const dim3 pg(ceil_div(ps.n, 256)), blk(256);
#define CALL_C2_PARTICLE(S, M) \
  if (kind == C2_TRANSFER) hipLaunchKernelGGL((c2_particle_kernel<M, C2_TRANSFER>), pg, blk, 0, L.stream, mp, pd, (const int *)slotOf, rec); \
  else hipLaunchKernelGGL((c2_particle_kernel<M, C2_FORCE>), pg, blk, 0, L.stream, mp, pd, (const int *)slotOf, rec)
  if (kind == C2_MOMENTUM) hipLaunchKernelGGL((c2_particle_kernel<ZS_MPM_FIXED_COROTATED, C2_MOMENTUM>), pg, blk, 0, L.stream, mp, pd,
                                                (const int *)slotOf, rec);
  else { ZSR_DISPATCH_PURE_(0, p->model, CALL_C2_PARTICLE) }
  const C2Buckets bk{buckets->dense ? HtDev{} : buckets->table->dev(), buckets->dense};
  static const bool cell1 = [] { const char *e = getenv("ZS_ROCM_C2_CELL1"); return e && e[0] == '1'; }();  // the cell-per-lane kernel
  constexpr int CXN = ZS_C2_CXN;  // cells per lane = CXN x 2 x 2
#define CALL_C2_CELLS(S, K)                                                                                                               \
  if (cell1)                                                                                                                              \
    hipLaunchKernelGGL((p2c2g_cell_kernel<S, K>), dim3((unsigned)nblocks), dim3(S == 4 ? 64 : 256), 0, L.stream, mp, t, bk,                \
                       (const int *)buckets->offsets, (const unsigned long long *)sub, (const float4 *)rec, sums);                        \
  else                                                                                                                                    \
    hipLaunchKernelGGL((p2c2g_cell8_kernel<S, K, CXN>), dim3((unsigned)ceil_div(nblocks, (size_t)(S == 4 ? 4 * CXN : 1))),                \
                       dim3(S == 4 ? 64 : 128 / CXN), 0, L.stream, mp, t, bk, (const int *)buckets->offsets,                              \
                       (const unsigned long long *)sub, (const float4 *)rec, sums, (int)nblocks);                                         \
  hipLaunchKernelGGL((p2c2g_node_kernel<S, K>), dim3((unsigned)nblocks), dim3(S == 4 ? 64 : 256), 0, L.stream, mp, t, (const float *)sums, grid)
#define CALL_C2_KIND(S)                                    \
  if (kind == C2_TRANSFER) { CALL_C2_CELLS(S, C2_TRANSFER); } \
  else if (kind == C2_MOMENTUM) { CALL_C2_CELLS(S, C2_MOMENTUM); } \
  else { CALL_C2_CELLS(S, C2_FORCE); }
  if (p->side == 4) { CALL_C2_KIND(4) } else { CALL_C2_KIND(8) }
  return 0;
}

void zs_rocm_mpm_pre_g2c2p(zs_rocm_policy *pol, zs_rocm_particles ps) {
  Launch L(pol, "PreG2C2PTransfer");
  if (!ps.n) return;
  hipLaunchKernelGGL(pre_g2c2p_kernel, dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, make_particles(ps));
}

int zs_rocm_mpm_g2c2p(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_index_buckets *buckets,
                      const zs_rocm_bht_3 *tab, const float *grid, size_t nblocks) {
  (void)buckets;  // the particle side is a gather too: the buckets of the reference functor are not needed
  if (p->side != 4 && p->side != 8) return -1;
  if (!ps.n || !nblocks) return 0;
  Launch L(pol, "G2C2PTransfer");
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  const size_t nc = (size_t)p->side * p->side * p->side;
  float *cv = (float *)L.temp(sizeof(float) * 12 * nc * nblocks);
  if (p->side == 4) {
    hipLaunchKernelGGL((g2c2p_cell_kernel<4>), dim3((unsigned)nblocks), dim3(64), 0, L.stream, mp, t, grid, cv);
    hipLaunchKernelGGL((g2c2p_particle_kernel<4>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd, t, (const float *)cv);
  } else {
    hipLaunchKernelGGL((g2c2p_cell_kernel<8>), dim3((unsigned)nblocks), dim3(256), 0, L.stream, mp, t, grid, cv);
    hipLaunchKernelGGL((g2c2p_particle_kernel<8>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd, t, (const float *)cv);
  }
  return 0;
}

int zs_rocm_mpm_g2c2p_step(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab,
                           const float *grid, size_t nblocks) {
  if (p->side != 4 && p->side != 8) return -1;
  if (!ps.n || !nblocks) return 0;
  Launch L(pol, "G2C2PTransfer");
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  BhtDev t = tab->t.dev();
  const size_t nc = (size_t)p->side * p->side * p->side;
  float *cv = (float *)L.temp(sizeof(float) * 12 * nc * nblocks);
  const dim3 pg(ceil_div(ps.n, 256)), blk(256);
  const bool fluid = p->model == ZS_MPM_EQUATION_OF_STATE;
  if (p->side == 4) {
    hipLaunchKernelGGL((g2c2p_cell_kernel<4>), dim3((unsigned)nblocks), dim3(64), 0, L.stream, mp, t, grid, cv);
    if (fluid) hipLaunchKernelGGL((g2c2p_particle_kernel<4, true, true>), pg, blk, 0, L.stream, mp, pd, t, (const float *)cv);
    else hipLaunchKernelGGL((g2c2p_particle_kernel<4, true, false>), pg, blk, 0, L.stream, mp, pd, t, (const float *)cv);
  } else {
    hipLaunchKernelGGL((g2c2p_cell_kernel<8>), dim3((unsigned)nblocks), dim3(256), 0, L.stream, mp, t, grid, cv);
    if (fluid) hipLaunchKernelGGL((g2c2p_particle_kernel<8, true, true>), pg, blk, 0, L.stream, mp, pd, t, (const float *)cv);
    else hipLaunchKernelGGL((g2c2p_particle_kernel<8, true, false>), pg, blk, 0, L.stream, mp, pd, t, (const float *)cv);
  }
  return 0;
}

void zs_rocm_mpm_post_g2c2p(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps) {
  Launch L(pol, "PostG2C2PTransfer");
  if (!ps.n) return;
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
  if (p->model == ZS_MPM_EQUATION_OF_STATE)
    hipLaunchKernelGGL((post_g2c2p_kernel<true>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd);
  else hipLaunchKernelGGL((post_g2c2p_kernel<false>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd);
}

}  // extern "C"
