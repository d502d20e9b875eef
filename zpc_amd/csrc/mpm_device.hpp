#pragma once
// mpm_device.hpp -- device code of the MLS-MPM particle<->grid transfers for gfx950, shared by the translation units
// mpm.hip (partition, binning, grid update, test entries, halo), mpm_p2g.hip, mpm_g2p.hip and mpm_fused.hip: the kernels are
// templates over (block side, constitutive model, lane width); every TU instantiates only what its entry points launch, so the
// four files compile in parallel (one TU took 3.5 minutes).  Kernels have internal linkage (`static __global__`).
//
// Replaces, behind include/zs_rocm.h:
//   ComputeSparsity / EnlargeSparsity        simulation/sparsity/SparsityOp.hpp:59-115
//   P2GTransfer::operator()                  simulation/transfer/P2G.hpp:51-125 (+ cuda/simulation/transfer/P2G.hpp:38-116)
//   compute_stress_fixedcorotated / _sand    cuda/physics/ConstitutiveModel.hpp:10-326, math::svd cuda/math/matrix/svd.cuh
//   ComputeGridBlockVelocity                 simulation/grid/GridOp.hpp:71-108
//   G2PTransfer::operator()                  simulation/transfer/G2P.hpp:44-83
//
// The reference's CUDA P2G issues 27 hash queries + 189 global float atomics per particle.  Here:
//   * particles are binned by the 4x4x4 cell group ("bin") of their base node (count -> scan -> distribute,
//     the IndexBuckets idea of simulation/particle/Query.tpp:9-58) and stored round-robin over the 64 cells
//     of a bin: round r holds the r-th particle of every cell that has one;
//   * one wavefront owns one bin, lane c owns cell c.  All particles of a cell share the same 27 stencil
//     nodes, so the lane accumulates its 27 x 7 node contributions IN REGISTERS across its particles, then
//     adds them into a per-wave LDS arena of 6^3 nodes x 7 channels in 27 conflict-free phases (plain
//     ds_read/ds_write: for a fixed stencil offset the 64 cells map to 64 distinct nodes on 32 distinct
//     banks per half-wave), and the arena is flushed ONCE to the grid with global_atomic_add_f32.
//     LDS float atomics are deliberately NOT used: ds_add_f32 measures 193 cycles per wave-instruction on
//     gfx950 (3 cycles per lane, serialised) against 4.2 for ds_add_u32 and 11 for a read-add-write pair
//     (tools/lds_bench.hip, profiles/); the first version of this kernel spent 94 % of its time in them;
//   * particles that left their cell since the last re-binning are queued and handled by the exact
//     particle-order kernel afterwards, so results never depend on how fresh the bins are;
//   * the 3x3 SVD is per-lane scalar VALU (quaternion Jacobi, v_rsq_f32): it is not a dense
//     contraction, so no MFMA;
//   * G2P: the lane loads the 27 x 3 node velocities of its cell from the LDS arena once and keeps them in
//     registers for all its particles.
// Algorithmic HBM bytes per particle: P2G 100 B read (+ 7 B grid), G2P 48 B read + 96 B write (+1.5 B grid).
#include "bht.hpp"
#include "hashtable.hpp"

namespace zsr {

void exclusive_scan_u32(Launch &L, const unsigned *in, size_t n, unsigned *out);
void radix_sort_pair_u32(Launch &L, const unsigned *kin, const int *vin, unsigned *kout, int *vout, size_t n, int sbit, int ebit);

// ======================================================================================= small math
__device__ __forceinline__ float rsq(float x) { return __frsqrt_rn(x); }

#define SVD_GAMMA 5.8284273147583007813f
#define SVD_CSTAR 0.9238795325112867f
#define SVD_SSTAR 0.3826834323650898f

// one Jacobi conjugation in the (X,Y) plane of the symmetric matrix S, accumulated into quaternion q=(w,v)
template <int X, int Y, int Z> __device__ __forceinline__ void jacobi_conj(float (&S)[3][3], float (&q)[4]) {
  float sh = S[X][Y] * 0.5f;
  float ch = S[X][X] - S[Y][Y];
  const bool ok = sh * sh >= 1.e-20f;
  sh = ok ? sh : 0.f;
  ch = ok ? ch : 1.f;
  float sh2 = sh * sh, ch2 = ch * ch;
  const float w = rsq(sh2 + ch2);
  sh *= w;
  ch *= w;
  const bool fix = ch2 <= SVD_GAMMA * sh2;  // angle too large for the approximation: use pi/8
  sh = fix ? SVD_SSTAR : sh;
  ch = fix ? SVD_CSTAR : ch;
  sh2 = sh * sh;
  ch2 = ch * ch;
  const float c = ch2 - sh2, s = 2.f * sh * ch;
  const float sxx = S[X][X], sxy = S[X][Y], syy = S[Y][Y], sxz = S[X][Z], syz = S[Y][Z];
  const float t1 = c * sxx + s * sxy, t2 = c * sxy + s * syy;
  const float t3 = -s * sxx + c * sxy, t4 = -s * sxy + c * syy;
  S[X][X] = c * t1 + s * t2;
  S[X][Y] = S[Y][X] = c * t3 + s * t4;
  S[Y][Y] = -s * t3 + c * t4;
  S[X][Z] = S[Z][X] = c * sxz + s * syz;
  S[Y][Z] = S[Z][Y] = -s * sxz + c * syz;
  const float qw = q[0], qx = q[1 + X], qy = q[1 + Y], qz = q[1 + Z];
  q[0] = qw * ch - qz * sh;
  q[1 + X] = qx * ch + qy * sh;
  q[1 + Y] = qy * ch - qx * sh;
  q[1 + Z] = qz * ch + qw * sh;
}

template <int A, int B, bool SWAPV> __device__ __forceinline__ void cond_swap_cols(float (&rho)[3], float (&Bm)[3][3], float (&Vm)[3][3]) {
  const bool sw = rho[A] < rho[B];
  const float ra = rho[A], rb = rho[B];
  rho[A] = sw ? rb : ra;
  rho[B] = sw ? ra : rb;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float ba = Bm[r][A], bb = Bm[r][B];
    Bm[r][A] = sw ? bb : ba;
    Bm[r][B] = sw ? -ba : bb;
    if constexpr (SWAPV) {
      const float va = Vm[r][A], vb = Vm[r][B];
      Vm[r][A] = sw ? vb : va;
      Vm[r][B] = sw ? -va : vb;
    }
  }
}

template <int P, int R> __device__ __forceinline__ void qr_step(float (&Bm)[3][3], float (&Um)[3][3]) {
  const float a1 = Bm[P][P], a2 = Bm[R][P];
  const float rho2 = a1 * a1 + a2 * a2;
  const bool ok = rho2 > 1.e-24f;
  const float ir = rsq(ok ? rho2 : 1.f);
  const float c = ok ? a1 * ir : 1.f, s = ok ? a2 * ir : 0.f;
#pragma unroll
  for (int col = 0; col < 3; ++col) {
    const float bp = Bm[P][col], br = Bm[R][col];
    Bm[P][col] = c * bp + s * br;
    Bm[R][col] = -s * bp + c * br;
  }
#pragma unroll
  for (int row = 0; row < 3; ++row) {
    const float up = Um[row][P], ur = Um[row][R];
    Um[row][P] = c * up + s * ur;
    Um[row][R] = -s * up + c * ur;
  }
}

// A = U diag(S) V^T; U, V rotations, |S0| >= |S1| >= |S2| (math::svd convention).  Outputs as [row][col] arrays:
// Um, Sg, and -- only when asked for -- Vm (sorted) and Bs = A V (sorted, before the QR), which lets the caller form
// P F^T = U diag(Phat) (F V)^T without ever building P or re-multiplying by F.
template <bool NEED_V, bool NEED_B>
__device__ __forceinline__ void svd3_core(const float (&A)[9], float (&Um)[3][3], float (&Sg)[3], float (&Vm)[3][3], float (&Bs)[3][3]) {
  float S[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) S[i][j] = A[3 * i] * A[3 * j] + A[1 + 3 * i] * A[1 + 3 * j] + A[2 + 3 * i] * A[2 + 3 * j];
  float q[4] = {1.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sweep = 0; sweep < 4; ++sweep) {
    jacobi_conj<0, 1, 2>(S, q);
    jacobi_conj<1, 2, 0>(S, q);
    jacobi_conj<2, 0, 1>(S, q);
  }
  const float n = rsq(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float w = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
  Vm[0][0] = 1 - 2 * (y * y + z * z); Vm[0][1] = 2 * (x * y - w * z);     Vm[0][2] = 2 * (x * z + w * y);
  Vm[1][0] = 2 * (x * y + w * z);     Vm[1][1] = 1 - 2 * (x * x + z * z); Vm[1][2] = 2 * (y * z - w * x);
  Vm[2][0] = 2 * (x * z - w * y);     Vm[2][1] = 2 * (y * z + w * x);     Vm[2][2] = 1 - 2 * (x * x + y * y);
  float Bm[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Bm[r][c] = A[r] * Vm[0][c] + A[r + 3] * Vm[1][c] + A[r + 6] * Vm[2][c];
  float rho[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) rho[c] = Bm[0][c] * Bm[0][c] + Bm[1][c] * Bm[1][c] + Bm[2][c] * Bm[2][c];
  cond_swap_cols<0, 1, NEED_V>(rho, Bm, Vm);
  cond_swap_cols<0, 2, NEED_V>(rho, Bm, Vm);
  cond_swap_cols<1, 2, NEED_V>(rho, Bm, Vm);
  if constexpr (NEED_B) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Bs[r][c] = Bm[r][c];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Um[r][c] = r == c ? 1.f : 0.f;
  qr_step<0, 1>(Bm, Um);
  qr_step<0, 2>(Bm, Um);
  qr_step<1, 2>(Bm, Um);
  Sg[0] = Bm[0][0]; Sg[1] = Bm[1][1]; Sg[2] = Bm[2][2];
}

// column-major 9-vector interface (diagnostic entry point zs_rocm_svd3)
__device__ __forceinline__ void svd3(const float (&A)[9], float (&U)[9], float (&Sg)[3], float (&V)[9]) {
  float Um[3][3], Vm[3][3], Bs[3][3];
  svd3_core<true, false>(A, Um, Sg, Vm, Bs);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      U[r + 3 * c] = Um[r][c];
      V[r + 3 * c] = Vm[r][c];
    }
}

// out = M1 diag(d) M2^T (math/matrix/MatrixUtils.h:26-47)
__device__ __forceinline__ void mat_diag_matT(float (&out)[9], const float (&m1)[9], const float (&d)[3], const float (&m2)[9]) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r + 3 * c] = m1[r] * d[0] * m2[c] + m1[r + 3] * d[1] * m2[c + 3] + m1[r + 6] * d[2] * m2[c + 6];
}
__device__ __forceinline__ void pft_vol(const float (&P)[9], const float (&F)[9], float volume, float (&PF)[9]) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) PF[r + 3 * c] = (P[r] * F[c] + P[r + 3] * F[c + 3] + P[r + 6] * F[c + 6]) * volume;
}

// The cached stress attribute (`particles.stress`, written by the tail of G2P / update_stress, read by P2G): P F^T vol is the Kirchhoff
// stress times the volume, symmetric for every isotropic model of P2G.hpp:82-101 (and for the fluid: -p I + viscosity (C + C^T)), so it
// is stored as its 6 distinct components {xx, xy, xz, yy, yz, zz} -- 24 instead of 36 bytes per particle on both the G2P write and the
// P2G read (88 instead of 100 B of particle state per P2G particle).  The symmetric part is taken: the off-diagonal pairs of the
// computed product differ by rounding only.
constexpr int STRESS_N = 6;
__device__ __forceinline__ void stress_pack(const float (&PF)[9], float (&S)[STRESS_N]) {
  S[0] = PF[0];
  S[1] = 0.5f * (PF[1] + PF[3]);
  S[2] = 0.5f * (PF[2] + PF[6]);
  S[3] = PF[4];
  S[4] = 0.5f * (PF[5] + PF[7]);
  S[5] = PF[8];
}
__device__ __forceinline__ void stress_unpack(const float (&S)[STRESS_N], float (&PF)[9]) {
  PF[0] = S[0]; PF[1] = S[1]; PF[2] = S[2];
  PF[3] = S[1]; PF[4] = S[3]; PF[5] = S[4];
  PF[6] = S[2]; PF[7] = S[4]; PF[8] = S[5];
}

struct Material {
  float volume, mu, lam, cohesion, beta, yieldSurface;
  int volCorrection;
  float yieldStress;           // von Mises
  float bm, xi, Msqr;          // NACC: bulk modulus NACCConfig::bulk(), hardening factor, M^2
  int hardeningOn;
  float bulk, viscosity;       // EquationOfState
  // derived on the host once (make_dev) so that the kernels find them in SGPRs: computed per wave they are loop invariants the compiler
  // hoists into VGPRs, and in the 128-register fused kernels every such register is a spill (r05: scratch reloads behind the record
  // prefetch = a full memory latency per chunk)
  float smu;                   // 2 mu
  float dpCoef;                // DruckerPrager: (3 lam + 2 mu) / (2 mu)
  float expCohesion;           // DruckerPrager: exp(cohesion)
};

// compute_stress_fixedcorotated (cuda/physics/ConstitutiveModel.hpp:10-47).  The reference forms P = U diag(Phat) V^T and
// then P F^T; since (F V) is already available from the SVD, P F^T = U diag(Phat) (F V)^T is formed directly.
__device__ __forceinline__ void stress_fixedcorotated(const Material &m, const float (&F)[9], float (&PF)[9]) {
  float U[3][3], S[3], V[3][3], B[3][3];
  svd3_core<false, true>(F, U, S, V, B);
  const float J = S[0] * S[1] * S[2];
  const float smu = 2.f * m.mu, slam = m.lam * (J - 1.f);
  float Ph[3];
  Ph[0] = (smu * (S[0] - 1.f) + slam * (S[1] * S[2])) * m.volume;
  Ph[1] = (smu * (S[1] - 1.f) + slam * (S[0] * S[2])) * m.volume;
  Ph[2] = (smu * (S[2] - 1.f) + slam * (S[0] * S[1])) * m.volume;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float u0 = U[r][0] * Ph[0], u1 = U[r][1] * Ph[1], u2 = U[r][2] * Ph[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) PF[r + 3 * c] = u0 * B[c][0] + u1 * B[c][1] + u2 * B[c][2];
  }
}

// compute_stress_sand (cuda/physics/ConstitutiveModel.hpp:246-326): Drucker-Prager return mapping in log-strain.
// logJp is updated.  The reference overwrites F with the projected F_e = U diag(New_S) V^T and then forms
// P F_e^T * vol with P = U diag(Phat) V^T; with V^T V = I that product is U diag(Phat_i New_S_i) U^T * vol -- the
// Kirchhoff stress -- so neither P nor V is needed for the force.  WRITE_F: also return the projected F (test entry).
template <bool WRITE_F>
__device__ __forceinline__ void stress_sand(const Material &m, float &logJp, float (&F)[9], float (&PF)[9]) {
  float U[3][3], S[3], V[3][3], B[3][3];
  svd3_core<WRITE_F, false>(F, U, S, V, B);
  const float smu = m.smu;
  float eps[3], NS[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float a = fabsf(S[i]);
    a = a > 1e-4f ? a : 1e-4f;
    eps[i] = logf(a) - m.cohesion;
  }
  const float sum_eps = eps[0] + eps[1] + eps[2];
  const float tr = sum_eps + logJp;
  float eh[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) eh[i] = eps[i] - (tr * (1.f / 3.f));
  const float ehn = sqrtf(eh[0] * eh[0] + eh[1] * eh[1] + eh[2] * eh[2]);
  bool newF = false;
  float Hs[3] = {0.f, 0.f, 0.f};  // log of the projected singular values
  if (tr >= 0.f) {  // case II: cone tip
    NS[0] = NS[1] = NS[2] = m.expCohesion;
    Hs[0] = Hs[1] = Hs[2] = m.cohesion;
    newF = true;
    if (m.volCorrection) logJp = m.beta * sum_eps + logJp;
  } else if (m.mu != 0.f) {
    logJp = 0.f;
    const float dg = ehn + m.dpCoef * tr * m.yieldSurface;
    float H[3];
    if (dg <= 0.f) {  // case I: inside the cone
#pragma unroll
      for (int i = 0; i < 3; ++i) H[i] = eps[i] + m.cohesion;
    } else {  // case III: onto the cone surface
      const float sc = dg * __frcp_rn(ehn);
#pragma unroll
      for (int i = 0; i < 3; ++i) H[i] = eps[i] - sc * eh[i] + m.cohesion;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if constexpr (WRITE_F) NS[i] = expf(H[i]);
      else NS[i] = 1.f;  // only its positivity matters below
      Hs[i] = H[i];
    }
    newF = true;
  }
  // New_S_log = log(New_S) (ConstitutiveModel.hpp:309): New_S = exp(H), so log(New_S) == H up to one rounding; the
  // mu == 0 && trace < 0 corner keeps the reference's log(0) = -inf
  float tau[3];
  {
    float lg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) lg[i] = NS[i] > 0.f ? Hs[i] : -INFINITY;
    const float trl = lg[0] + lg[1] + lg[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) tau[i] = (smu * lg[i] + m.lam * trl) * m.volume;  // Phat_i * New_S_i * vol
  }
  if (newF) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float u0 = U[r][0] * tau[0], u1 = U[r][1] * tau[1], u2 = U[r][2] * tau[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) PF[r + 3 * c] = u0 * U[c][0] + u1 * U[c][1] + u2 * U[c][2];
    }
    if constexpr (WRITE_F) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float u0 = U[r][0] * NS[0], u1 = U[r][1] * NS[1], u2 = U[r][2] * NS[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) F[r + 3 * c] = u0 * V[c][0] + u1 * V[c][1] + u2 * V[c][2];
      }
    }
  } else {
    // mu == 0 && trace < 0: F is not projected and New_S = 0 (reference corner case): P = U diag(-inf/0) V^T -> NaN/inf;
    // reproduce "non-finite" without caring about the exact pattern
#pragma unroll
    for (int d = 0; d < 9; ++d) PF[d] = tau[0];
  }
}

// compute_stress_vonmisesfixedcorotated (cuda/physics/ConstitutiveModel.hpp:47-116): von Mises return mapping of the
// Kirchhoff stress in principal space, F projected in place (the caller decides whether it is stored), then the
// fixed-corotated P F^T vol of the projected state.
__device__ __forceinline__ void stress_vonmises(const Material &m, float (&F)[9], float (&PF)[9]) {
  float U[9], S[3], V[9];
  svd3(F, U, S, V);
  float Sc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) Sc[d] = 1e-4f > S[d] ? 1e-4f : S[d];
  float J = Sc[0] * Sc[1] * Sc[2];
  float tau[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) tau[d] = 2 * m.mu * (Sc[d] - 1) * Sc[d] + m.lam * (J - 1) * J;
  const float tr = tau[0] + tau[1] + tau[2];
  float st[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) st[d] = tau[d] - (tr / 3.f);
  const float s_norm = sqrtf(st[0] * st[0] + st[1] * st[1] + st[2] * st[2]);
  const float scaled_tauy = sqrtf(2.f / (6.f - 3.f)) * m.yieldStress;
  if (s_norm - scaled_tauy > 0) {
    const float alpha = scaled_tauy / s_norm;
    J = 1.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float tau_new = alpha * st[d] + (tr / 3.f);
      const float b2m4ac = m.mu * m.mu - 2 * m.mu * (m.lam * (J - 1) * J - tau_new);
      S[d] = (m.mu + sqrtf(b2m4ac)) / (2 * m.mu);
    }
    mat_diag_matT(F, U, S, V);
  }
  J = S[0] * S[1] * S[2];
  const float smu = 2.f * m.mu, slam = m.lam * (J - 1.f);
  float Ph[3], P[9];
  Ph[0] = smu * (S[0] - 1.f) + slam * (S[1] * S[2]);
  Ph[1] = smu * (S[1] - 1.f) + slam * (S[0] * S[2]);
  Ph[2] = smu * (S[2] - 1.f) + slam * (S[0] * S[1]);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) P[r + 3 * c] = Ph[0] * U[r] * V[c] + Ph[1] * U[r + 3] * V[c + 3] + Ph[2] * U[r + 6] * V[c + 6];
  pft_vol(P, F, m.volume, PF);
}

// compute_stress_nacc (cuda/physics/ConstitutiveModel.hpp:118-243): non-associated Cam-Clay, three projection cases +
// hardening through logJp; F projected in place; neo-Hookean-type P F^T vol of the projected state.
__device__ __forceinline__ void stress_nacc(const Material &m, float &logJp, float (&F)[9], float (&PF)[9]) {
  float U[9], S[3], V[9];
  svd3(F, U, S, V);
  const float bm = m.bm, beta = m.beta, Msqr = m.Msqr, mu = m.mu;
  const float p0 = bm * (0.00001f + sinhf(m.xi * (-logJp > 0 ? -logJp : 0)));
  const float p_min = -beta * p0;
  const float Je_trial = S[0] * S[1] * S[2];
  const float Bh[3] = {S[0] * S[0], S[1] * S[1], S[2] * S[2]};
  const float trB = (Bh[0] + Bh[1] + Bh[2]) / 3.f;
  const float Jm = mu * powf(Je_trial, -2.f / 3.f);
  const float sh[3] = {Jm * (Bh[0] - trB), Jm * (Bh[1] - trB), Jm * (Bh[2] - trB)};
  const float psi = bm * 0.5f * (Je_trial - 1.f / Je_trial);
  const float p_trial = -psi * Je_trial;
  const float ys = 3.f / 2.f * (1 + 2.f * beta);
  const float yp = (Msqr * (p_trial - p_min) * (p_trial - p0));
  const float sn = sh[0] * sh[0] + sh[1] * sh[1] + sh[2] * sh[2];
  const float y = (ys * sn) + yp;
  if (p_trial > p0) {  // case 1: max tip
    const float Je_new = sqrtf(-2.f * p0 / bm + 1.f);
    S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
    mat_diag_matT(F, U, S, V);
    if (m.hardeningOn) logJp += logf(Je_trial / Je_new);
  } else if (p_trial < p_min) {  // case 2: min tip
    const float Je_new = sqrtf(-2.f * p_min / bm + 1.f);
    S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
    mat_diag_matT(F, U, S, V);
    if (m.hardeningOn) logJp += logf(Je_trial / Je_new);
  } else if (y >= 1e-4) {  // case 3: onto the yield surface + hardening
    const float Bs = powf(Je_trial, 2.f / 3.f) / mu * sqrtf(-yp / ys) / sqrtf(sn);
#pragma unroll
    for (int i = 0; i < 3; ++i) S[i] = sqrtf(sh[i] * Bs + trB);
    mat_diag_matT(F, U, S, V);
    if (m.hardeningOn && p0 > 1e-4 && p_trial < p0 - 1e-4 && p_trial > 1e-4 + p_min) {
      const float pc = (1.f - beta) * p0 / 2;
      const float q_trial = sqrtf(3.f / 2.f * sn);
      float dir[2] = {pc - p_trial, -q_trial};
      const float dn = sqrtf(dir[0] * dir[0] + dir[1] * dir[1]);
      dir[0] /= dn;
      dir[1] /= dn;
      const float Cq = Msqr * (pc - p_min) * (pc - p0);
      const float Bq = Msqr * dir[0] * (2 * pc - p0 - p_min);
      const float Aq = Msqr * dir[0] * dir[0] + (1 + 2 * beta) * dir[1] * dir[1];
      const float l1 = (-Bq + sqrtf(Bq * Bq - 4 * Aq * Cq)) / (2 * Aq);
      const float l2 = (-Bq - sqrtf(Bq * Bq - 4 * Aq * Cq)) / (2 * Aq);
      const float p1 = pc + l1 * dir[0], p2 = pc + l2 * dir[0];
      const float pf = (p_trial - pc) * (p1 - pc) > 0 ? p1 : p2;
      const float tJ = (-2 * pf / bm + 1);
      const float Jf = sqrtf(tJ > 0 ? tJ : -tJ);
      if (Jf > 1e-4) logJp += logf(Je_trial / Jf);
    }
  }
  const float J = S[0] * S[1] * S[2];
  float b[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) b[r + 3 * c] = F[r] * F[c] + F[r + 3] * F[c + 3] + F[r + 6] * F[c + 6];  // F F^T
  const float trb = (b[0] + b[4] + b[8]) / 3.f;
  b[0] -= trb; b[4] -= trb; b[8] -= trb;
  const float dc = mu * powf(J, -2.f / 3.f), ic = bm * .5f * (J * J - 1.f);
#pragma unroll
  for (int d = 0; d < 9; ++d) PF[d] = (dc * b[d] + ((d & 3) ? 0.f : ic)) * m.volume;
}

// EquationOfState branch of P2GTransfer (simulation/transfer/P2G.hpp:60-81): J = particles.J, C = particles.C
__device__ __forceinline__ void stress_eos(const Material &m, float J, const float (&C)[9], float (&PF)[9]) {
  const float vol = m.volume * J;
  float pressure = m.bulk;
  {
    const float J2 = J * J, J4 = J2 * J2;
    pressure = pressure * (1 / (J * J2 * J4) - 1);
  }
  PF[0] = ((C[0] + C[0]) * m.viscosity - pressure) * vol;
  PF[1] = (C[1] + C[3]) * m.viscosity * vol;
  PF[2] = (C[2] + C[6]) * m.viscosity * vol;
  PF[3] = (C[3] + C[1]) * m.viscosity * vol;
  PF[4] = ((C[4] + C[4]) * m.viscosity - pressure) * vol;
  PF[5] = (C[5] + C[7]) * m.viscosity * vol;
  PF[6] = (C[6] + C[2]) * m.viscosity * vol;
  PF[7] = (C[7] + C[5]) * m.viscosity * vol;
  PF[8] = ((C[8] + C[8]) * m.viscosity - pressure) * vol;
}
// the fluid model keeps J where the solids keep F (component 0 of the `F` attribute); -2 = fluid without a constitutive
// update in G2P (the G2P kernels' "no model" value for solids is -1)
constexpr int MPM_FLUID_NO_STRESS = -2;
__host__ __device__ constexpr bool model_is_fluid(int model) { return model == ZS_MPM_EQUATION_OF_STATE || model == MPM_FLUID_NO_STRESS; }
// which models carry the scalar plastic state logJp (P2G.hpp:88-101)
__host__ __device__ constexpr bool model_uses_logjp(int model) { return model == ZS_MPM_DRUCKER_PRAGER || model == ZS_MPM_NACC; }
// one entry point for the four constitutive models of P2G.hpp:82-101.  F is the local copy: the plastic models project it
// in place, P2G / G2P never store it back (only logJp), the test entry zs_rocm_mpm_stress does (WRITE_F).
template <int MODEL, bool WRITE_F = false>
__device__ __forceinline__ void model_stress(const Material &m, float &logJp, float (&F)[9], float (&PF)[9], const float (&C)[9]) {
  if constexpr (MODEL == ZS_MPM_EQUATION_OF_STATE) stress_eos(m, F[0], C, PF);
  else if constexpr (MODEL == ZS_MPM_FIXED_COROTATED) stress_fixedcorotated(m, F, PF);
  else if constexpr (MODEL == ZS_MPM_DRUCKER_PRAGER) stress_sand<WRITE_F>(m, logJp, F, PF);
  else if constexpr (MODEL == ZS_MPM_VONMISES_FIXED_COROTATED) stress_vonmises(m, F, PF);
  else stress_nacc(m, logJp, F, PF);
}

// ======================================================================================= arena
// node k (0, 1, 2) of the stencil minus the local position, k dx - lp -- written without the product (k dx is a loop invariant the compiler
// would keep in a VGPR; 2 dx is exact, so the fma returns the same bits, and 0 dx - lp = -lp up to the sign of a zero)
__device__ __forceinline__ float node_off(float dx, int k, float lp) { return k == 0 ? -lp : (k == 1 ? dx - lp : fmaf(2.f, dx, -lp)); }
// LocalArena<collocated, quadratic> (simulation/Utils.hpp:47-75, InterpolationKernel.hpp:47-55,93-130)
struct Arena {
  int corner[3];
  float lp[3];    // local position * dx
  float w[3][3];  // w[axis][k]
};
// X = pos * (1/dx): the reference divides (simulation/Utils.hpp:52-55); the product differs by <= 1 ulp, which moves
// a weight by O(1e-7) and never changes which bin a particle is stored in because the binning kernel uses this
// same expression.
__device__ __forceinline__ void make_arena(float dx, float dxinv, const float (&pos)[3], Arena &a) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float X = pos[d] * dxinv;
    const float fl = floorf(X - 0.5f);
    a.corner[d] = (int)fl;
    const float lpn = X - fl;
    const float d0 = lpn - floorf(lpn - 0.5f);
    a.w[d][0] = 0.5f * (1.5f - d0) * (1.5f - d0);
    const float d1 = d0 - 1.0f;
    a.w[d][1] = 0.75f - d1 * d1;
    const float zz = 0.5f + d1;
    a.w[d][2] = 0.5f * zz * zz;
    a.lp[d] = lpn * dx;
  }
}
__device__ __forceinline__ void make_arena(float dx, const float (&pos)[3], Arena &a) { make_arena(dx, 1.0f / dx, pos, a); }

__device__ __forceinline__ int floordiv(int a, int b) { return (a + (a < 0 ? -b + 1 : 0)) / b; }

template <int N> __device__ __forceinline__ void load_attr(const Port<float> &p, size_t i, float (&out)[N]) {
  const float *b = p.base + p.off(i);
  const size_t cs = p.cstride();
#pragma unroll
  for (int d = 0; d < N; ++d) out[d] = b[d * cs];
}
template <int N> __device__ __forceinline__ void store_attr(const Port<float> &p, size_t i, const float (&v)[N]) {
  float *b = p.base + p.off(i);
  const size_t cs = p.cstride();
#pragma unroll
  for (int d = 0; d < N; ++d) b[d * cs] = v[d];
}

// Fast particle addressing for the binned kernels.  When every attribute lives in ONE TileVector<f32, LW> (same tile
// width / channel count, iterator index 0 -- the host checks this) the element offset of particle i,
// ((i / LW) * chns) * LW + i % LW, is computed once per particle and every component load/store becomes
// base(SGPR) + offset(VGPR) + d * LW * 4 (immediate): ~1 VALU per attribute instead of ~3 per component.
// LW == 0: generic iterator ports (AoS vectors, mixed layouts).
template <int LW> struct POff { size_t o; };
template <int LW> __device__ __forceinline__ POff<LW> particle_offset(unsigned chns, size_t i) {
  POff<LW> r;
  if constexpr (LW != 0) r.o = ((i / LW) * (size_t)chns) * LW + (i % LW);
  else r.o = i;
  return r;
}
template <int LW, int N> __device__ __forceinline__ void pload(const Port<float> &p, POff<LW> o, float (&out)[N]) {
  if constexpr (LW != 0) {
    const float *b = p.base + o.o;
#pragma unroll
    for (int d = 0; d < N; ++d) {
      out[d] = b[d * LW];
    }
  } else
    load_attr<N>(p, o.o, out);
}
template <int LW> __device__ __forceinline__ float pload1(const Port<float> &p, POff<LW> o, int comp = 0) {
  if constexpr (LW != 0) return p.base[o.o + comp * LW];
  else return p.base[p.off(o.o) + comp * p.cstride()];
}
template <int LW, int N> __device__ __forceinline__ void pstore(const Port<float> &p, POff<LW> o, const float (&v)[N]) {
  if constexpr (LW != 0) {
    float *b = p.base + o.o;
#pragma unroll
    for (int d = 0; d < N; ++d) {
#ifdef ZS_PSTORE_NT  // measurement builds: particle state written with non-temporal stores
      __builtin_nontemporal_store(v[d], b + d * LW);
#else
      b[d * LW] = v[d];
#endif
    }
  } else
    store_attr<N>(p, o.o, v);
}
template <int LW> __device__ __forceinline__ void pstore1(const Port<float> &p, POff<LW> o, float v) {
#ifdef ZS_PSTORE_NT
  if constexpr (LW != 0) __builtin_nontemporal_store(v, p.base + o.o);
#else
  if constexpr (LW != 0) p.base[o.o] = v;
#endif
  else p.base[p.off(o.o)] = v;
}
// deformation state of a particle: F (9 components) for the solids, the volume ratio J = component 0 of the same attribute
// for the EquationOfState fluid (Structurefree.hpp: particles.F / particles.J)
template <int LW, bool FLUID> __device__ __forceinline__ void pload_state(const Port<float> &p, POff<LW> o, float (&F)[9]) {
  if constexpr (FLUID) {
#pragma unroll
    for (int d = 1; d < 9; ++d) F[d] = 0.f;
    F[0] = pload1<LW>(p, o);
  } else
    pload<LW, 9>(p, o, F);
}
template <int LW, bool FLUID> __device__ __forceinline__ void pstore_state(const Port<float> &p, POff<LW> o, const float (&F)[9]) {
  if constexpr (FLUID) pstore1<LW>(p, o, F[0]);
  else pstore<LW, 9>(p, o, F);
}
template <bool FLUID> __device__ __forceinline__ void load_state(const Port<float> &p, size_t i, float (&F)[9]) {
  if constexpr (FLUID) {
#pragma unroll
    for (int d = 1; d < 9; ++d) F[d] = 0.f;
    F[0] = p.base[p.off(i)];
  } else
    load_attr<9>(p, i, F);
}
// G2P: F <- (I + dt C) F (G2P.hpp:75-78, MatrixUtils.h:136-146) or J <- (1 + tr(C) dt) J (:70-74)
template <bool FLUID> __device__ __forceinline__ void advance_state(const float (&oldF)[9], const float (&C)[9], float dt, float (&F)[9]) {
  if constexpr (FLUID) {
#pragma unroll
    for (int d = 1; d < 9; ++d) F[d] = 0.f;
    F[0] = (1 + (C[0] + C[4] + C[8]) * dt) * oldF[0];
  } else {
    float tmp[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) tmp[d] = C[d] * dt + ((d & 0x3) ? 0.f : 1.f);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) F[r + 3 * c] = tmp[r] * oldF[3 * c] + tmp[r + 3] * oldF[3 * c + 1] + tmp[r + 6] * oldF[3 * c + 2];
  }
}

struct ParticlesDev {
  Port<float> mass, pos, vel, C, F, logJp, stress;
  size_t n;
};
// third "model" of the P2G kernels: P F^T * vol is read from the particles' `stress` attribute (written by the G2P of
// the previous step, or by zs_rocm_mpm_update_stress) instead of being recomputed
constexpr int MPM_CACHED_STRESS = 100;

template <int N> __device__ __forceinline__ void load_attr(const Port<float> &p, size_t i, float (&out)[N]);
template <int N> __device__ __forceinline__ void store_attr(const Port<float> &p, size_t i, const float (&v)[N]);
struct MpmDev {
  Material mat;
  int model;
  float dx, dt;
  float dxi, D_inv;  // 1 / dx, 4 / dx^2 (host-derived: see Material)
  float fscale, fscaleDx;  // -dt D_inv (contrib = -dt D_inv P F^T vol, P2G.hpp:105), and that times dx
  int kscale;  // partition keys are block coordinates (1: Grids + HashTable/bht convention) or block ORIGINS in cells
               // (SIDE: SparseGrid convention, geometry/SparseGrid.hpp:305-309)
};

// per-particle constitutive update -> contrib = -dt * D_inv * (P F^T vol)   (P2G.hpp:60-105)
template <int MODEL>
__device__ __forceinline__ void particle_contrib(const MpmDev &mp, const ParticlesDev &ps, size_t i, float D_inv, float (&contrib)[9]) {
  float F[9];
  if constexpr (MODEL == MPM_CACHED_STRESS) {
    float S[STRESS_N];
    load_attr<STRESS_N>(ps.stress, i, S);
    stress_unpack(S, contrib);
  } else {
    load_state<model_is_fluid(MODEL)>(ps.F, i, F);
    float Cp[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (model_is_fluid(MODEL)) load_attr<9>(ps.C, i, Cp);
    float lj = 0.f;
    if constexpr (model_uses_logjp(MODEL)) lj = ps.logJp.base[ps.logJp.off(i)];
    model_stress<MODEL>(mp.mat, lj, F, contrib, Cp);
    if constexpr (model_uses_logjp(MODEL)) ps.logJp.base[ps.logJp.off(i)] = lj;  // P2G.hpp:101; the projected F is not written back
  }
#pragma unroll
  for (int d = 0; d < 9; ++d) contrib[d] = contrib[d] * -mp.dt * D_inv;
}

// ======================================================================================= sparsity
static __global__ __launch_bounds__(256) void compute_sparsity_kernel(BhtDev t, Port<float> pos, size_t n, float dxinv, int side, int kscale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int b[3] = {0, 0, 0};
  if (valid) {
    float p[3];
    load_attr<3>(pos, i, p);
#pragma unroll
    for (int d = 0; d < 3; ++d) b[d] = floordiv((int)floorf(p[d] * dxinv + 0.5f) + (-2), side) * kscale;
  }
  // neighbouring lanes usually carry the same block: let only the first lane of a run insert (the others
  // would get sentinel_v back from insert anyway)
  const int px = shfl_up(b[0], 1), py = shfl_up(b[1], 1), pz = shfl_up(b[2], 1);
  const bool pvalid = shfl_up((int)valid, 1) != 0;
  const bool dup = lane_id() != 0 && pvalid && px == b[0] && py == b[1] && pz == b[2];
  if (valid && !dup) bht_insert<3>(t, b);
}
static __global__ __launch_bounds__(256) void enlarge_sparsity_kernel(BhtDev t, int nblocks, int lo0, int lo1, int lo2, int e0, int e1, int e2, int kscale) {
  // thread per (block, offset)
  const int per = e0 * e1 * e2;
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nblocks * per) return;
  const int i = (int)(g / per), o = (int)(g % per);
  const int dx = lo0 + o / (e1 * e2), dy = lo1 + (o / e2) % e1, dz = lo2 + o % e2;
  int k[3] = {t.activeKeys[3 * (size_t)i] + dx * kscale, t.activeKeys[3 * (size_t)i + 1] + dy * kscale, t.activeKeys[3 * (size_t)i + 2] + dz * kscale};
  bht_insert<3>(t, k);
}
// the same functors on a zs::HashTable<i32,3,int> (simulation/sparsity/SparsityOp.hpp:59-115 are written against HashTableView)
static __global__ __launch_bounds__(256) void compute_sparsity_ht_kernel(HtDev t, Port<float> pos, size_t n, float dxinv, int side) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int b[3] = {0, 0, 0};
  if (valid) {
    float p[3];
    load_attr<3>(pos, i, p);
#pragma unroll
    for (int d = 0; d < 3; ++d) b[d] = floordiv((int)floorf(p[d] * dxinv + 0.5f) + (-2), side);
  }
  const int px = shfl_up(b[0], 1), py = shfl_up(b[1], 1), pz = shfl_up(b[2], 1);
  const bool pvalid = shfl_up((int)valid, 1) != 0;
  const bool dup = lane_id() != 0 && pvalid && px == b[0] && py == b[1] && pz == b[2];
  if (valid && !dup) ht_insert<3>(t, b);
}
static __global__ __launch_bounds__(256) void enlarge_sparsity_ht_kernel(HtDev t, int nblocks, int lo0, int lo1, int lo2, int e0, int e1, int e2) {
  const int per = e0 * e1 * e2;
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nblocks * per) return;
  const int i = (int)(g / per), o = (int)(g % per);
  int k[3] = {t.activeKeys[3 * (size_t)i] + lo0 + o / (e1 * e2), t.activeKeys[3 * (size_t)i + 1] + lo1 + (o / e2) % e1,
              t.activeKeys[3 * (size_t)i + 2] + lo2 + o % e2};
  ht_insert<3>(t, k);
}
// index_buckets_for_particles (simulation/particle/Query.tpp:9-58): ComputeSparsity with blockLen 1 / offset 0, then
// SpatiallyCount (sparsity/SparsityOp.hpp:117-152)
static __global__ __launch_bounds__(256) void ib_cells_kernel(HtDev t, Port<float> pos, size_t n, float dxinv, float displacement, int *full) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int b[3] = {0, 0, 0};
  if (valid) {
    float p[3];
    load_attr<3>(pos, i, p);
#pragma unroll
    for (int d = 0; d < 3; ++d) b[d] = (int)floorf(p[d] * dxinv + displacement);
  }
  const int px = shfl_up(b[0], 1), py = shfl_up(b[1], 1), pz = shfl_up(b[2], 1);
  const bool pvalid = shfl_up((int)valid, 1) != 0;
  const bool dup = lane_id() != 0 && pvalid && px == b[0] && py == b[1] && pz == b[2];
  if (valid && !dup && ht_insert<3>(t, b) == HT_FAIL) *full = 1;  // table too small for the occupied cells: the host grows it and retries
}
static __global__ __launch_bounds__(256) void ib_count_kernel(HtDev t, Port<float> pos, size_t n, float dxinv, float displacement, unsigned *counts,
                                                       unsigned *cellOf, int *ids) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[3];
  load_attr<3>(pos, i, p);
  int b[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) b[d] = (int)floorf(p[d] * dxinv + displacement);
  int c = ht_query<3>(t, b);
  if (c < 0) c = *t.cnt;  // not in the table (cannot happen after a successful cell pass): the spare last bucket, never out of bounds
  cellOf[i] = (unsigned)c;
  ids[i] = (int)i;
  atomicAdd(&counts[c], 1u);
}
// buckets over the cells of a block partition (zs_rocm_index_buckets_for_partition): bucket = block * side^3 + cell id of the cell
// that contains the particle; particles whose cell is not in the partition go to the extra bucket `nbuckets`
static __global__ __launch_bounds__(256) void ib_dense_count_kernel(BhtDev t, Port<float> pos, size_t n, float dxinv, int side, int kscale,
                                                             int nbuckets, unsigned *counts, unsigned *cellOf, int *ids) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[3];
  load_attr<3>(pos, i, p);
  int key[3], loc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int c = (int)floorf(p[d] * dxinv);
    loc[d] = c & (side - 1);
    key[d] = (c - loc[d]) / side * kscale;
  }
  const int b = bht_query<3>(t, key);
  const int bucket = b < 0 ? nbuckets : b * side * side * side + (loc[0] * side + loc[1]) * side + loc[2];
  cellOf[i] = (unsigned)bucket;
  ids[i] = (int)i;
  atomicAdd(&counts[bucket], 1u);
}
static __global__ __launch_bounds__(256) void build_neighbors_kernel(BhtDev t, int nblocks, int *nbr, int kscale) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nblocks * 8) return;
  const int i = (int)(g >> 3), o = (int)(g & 7);
  int k[3] = {t.activeKeys[3 * (size_t)i] + (o >> 2) * kscale, t.activeKeys[3 * (size_t)i + 1] + ((o >> 1) & 1) * kscale,
              t.activeKeys[3 * (size_t)i + 2] + (o & 1) * kscale};
  nbr[g] = bht_query<3>(t, k);
}

// ======================================================================================= binning
// A "bin" is a 4x4x4 group of cells = 64 cells = one wavefront.  SIDE 4: bin == grid block.  SIDE 8: a grid
// block holds 2x2x2 bins, bin = block * 8 + sub, sub = ((lx>>2)*2 + (ly>>2))*2 + (lz>>2).
// Launch order of the per-bin kernels: plain blockIdx.  (r04, measured: giving XCD k the k-th contiguous eighth of the bins -- the
// dispatcher deals workgroups round-robin over the 8 XCDs -- changes neither the atomics' write traffic, which is write-through per
// touched 32-byte sector whatever the order, nor the time for the better: the empty apron bins end up on a few XCDs and the
// stand-alone P2G runs 1.83 -> 2.09 ms, the slotted step 7.9 -> 11.3 ms; numbering the blocks lexicographically or along the Morton
// curve instead of in insertion order: 1.86 / 1.91 ms and 8.5 / 8.2 ms.  profiles/r04_launch_order.md.  -DZS_ROCM_XCD_CHUNKS keeps
// the mapping for measurement builds.)
__device__ __forceinline__ unsigned xcd_chunked(unsigned i, unsigned n) {
#ifdef ZS_ROCM_XCD_CHUNKS
  const unsigned q = n >> 3, rem = n & 7u, k = i & 7u, j = i >> 3;
  return k * q + (k < rem ? k : rem) + j;
#else
  (void)n;
  return i;
#endif
}

template <int SIDE> constexpr int bins_per_block() { return (SIDE / 4) * (SIDE / 4) * (SIDE / 4); }

template <int SIDE>
static __global__ __launch_bounds__(256) void bin_count_kernel(BhtDev t, Port<float> pos, size_t n, float dx, unsigned *cellCount,
                                                        unsigned *cellOf, unsigned *rankOf, int *err, int kscale) {
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
    *err = 1;
    cellOf[i] = 0xffffffffu;
    return;
  }
  const int sub = SIDE == 4 ? 0 : (((loc[0] >> 2) * 2 + (loc[1] >> 2)) * 2 + (loc[2] >> 2));
  const unsigned cell = ((unsigned)b * bins_per_block<SIDE>() + sub) * 64u +
                        (unsigned)(((loc[0] & 3) * 4 + (loc[1] & 3)) * 4 + (loc[2] & 3));
  cellOf[i] = cell;
  rankOf[i] = atomicAdd(&cellCount[cell], 1u);
}
static __global__ __launch_bounds__(256) void bin_place_kernel(size_t n, const unsigned *cellStart, const unsigned *cellOf,
                                                        const unsigned *rankOf, int *byCell) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned c = cellOf[i];
  if (c != 0xffffffffu) byCell[cellStart[c] + rankOf[i]] = (int)i;
}
// one wave per bin, lane = cell: (cell, rank) order -> (rank, cell) order.  The ranks handed out by the counting pass are in
// arrival order of its atomics; the lane first sorts its cell's particle ids (odd-even transposition network in registers,
// K = 8 / 16 / 32 chosen per bin), so that within a cell the particles keep their previous relative order: a particle that
// did not change cell stays in "its" round, and the permutation of a re-ordering fused step is the identity except around the
// movers (coalesced reads through `order`).
template <int K>
__device__ __forceinline__ void bin_rr_emit(unsigned cnt, unsigned st, const int *byCell, int *order, unsigned &base, unsigned long long lt) {
  int ids[K];
#pragma unroll
  for (int k = 0; k < K; ++k) ids[k] = (unsigned)k < cnt ? byCell[st + k] : 0x7fffffff;
#pragma unroll
  for (int pass = 0; pass < K; ++pass)
#pragma unroll
    for (int k = pass & 1; k + 1 < K; k += 2) {
      const int a = ids[k], b = ids[k + 1];
      ids[k] = a < b ? a : b;
      ids[k + 1] = a < b ? b : a;
    }
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const bool has = cnt > (unsigned)r;
    const unsigned long long m = __ballot(has);
    if (!m) return;
    if (has) order[base + (unsigned)__popcll(m & lt)] = ids[r];
    base += (unsigned)__popcll(m);
  }
}
static __global__ __launch_bounds__(64) void bin_roundrobin_kernel(int nbins, const unsigned *cellStart, const unsigned *cellCount,
                                                            const int *byCell, int *order, int *binStart, unsigned total) {
  const int bin = blockIdx.x, c = threadIdx.x;
  const unsigned cnt = cellCount[(size_t)bin * 64 + c], st = cellStart[(size_t)bin * 64 + c];
  unsigned base = shfl(st, 0);
  if (c == 0) {
    binStart[bin] = (int)base;
    if (bin == nbins - 1) binStart[nbins] = (int)total;
  }
  const unsigned long long lt = lanemask_lt();
  unsigned mx = cnt;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned o = shfl_down(mx, d);
    mx = o > mx ? o : mx;
  }
  mx = shfl(mx, 0);
  if (mx <= 8u) bin_rr_emit<8>(cnt, st, byCell, order, base, lt);
  else if (mx <= 16u) bin_rr_emit<16>(cnt, st, byCell, order, base, lt);
  else bin_rr_emit<32>(cnt, st, byCell, order, base, lt);
  for (unsigned r = 32;; ++r) {  // cells with more than 32 particles: the rest in arrival order
    const bool has = cnt > r;
    const unsigned long long m = __ballot(has);
    if (!m) break;
    if (has) order[base + (unsigned)__popcll(m & lt)] = byCell[st + r];
    base += (unsigned)__popcll(m);
  }
}

// ======================================================================================= P2G
// ---- particle-order path: the reference's algorithm (hash query + global float atomics per node), with the
//      27 queries folded into the <= 8 distinct blocks a stencil can touch.
template <int SIDE, int MODEL>
__device__ __forceinline__ void p2g_scatter_global(const MpmDev &mp, const ParticlesDev &ps, size_t i, const BhtDev &t, float *grid,
                                                   float D_inv) {
  constexpr int NC = SIDE * SIDE * SIDE;
  float pos[3], vel[3], C[9], contrib[9];
  load_attr<3>(ps.pos, i, pos);
  load_attr<3>(ps.vel, i, vel);
  load_attr<9>(ps.C, i, C);
  const float mass = ps.mass.base[ps.mass.off(i)];
  particle_contrib<MODEL>(mp, ps, i, D_inv, contrib);
  Arena ar;
  make_arena(mp.dx, mp.dxi, pos, ar);
  int loc[3], key[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    loc[d] = ar.corner[d] & (SIDE - 1);
    key[d] = (ar.corner[d] - loc[d]) / SIDE * mp.kscale;
  }
  int blk[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const bool need = (!(o & 4) || loc[0] + 2 >= SIDE) && (!(o & 2) || loc[1] + 2 >= SIDE) && (!(o & 1) || loc[2] + 2 >= SIDE);
    int k[3] = {key[0] + (o >> 2) * mp.kscale, key[1] + ((o >> 1) & 1) * mp.kscale, key[2] + (o & 1) * mp.kscale};
    blk[o] = need ? bht_query<3>(t, k) : -1;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int x = loc[0] + a, y = loc[1] + b, z = loc[2] + c;
        const int o = ((x >= SIDE) << 2) | ((y >= SIDE) << 1) | (z >= SIDE);
        int bn = blk[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) bn = (o == q) ? blk[q] : bn;
        if (bn < 0) continue;  // the reference does not check (P2G.hpp:109-110); a valid partition never gets here
        const int cell = ((x & (SIDE - 1)) * SIDE + (y & (SIDE - 1))) * SIDE + (z & (SIDE - 1));
        float *g = grid + (size_t)bn * 7 * NC + cell;
        const float xi0 = (float)a * mp.dx - ar.lp[0], xi1 = (float)b * mp.dx - ar.lp[1], xi2 = (float)c * mp.dx - ar.lp[2];
        float W = ar.w[0][a];
        W *= ar.w[1][b];
        W *= ar.w[2][c];
        unsafeAtomicAdd(g, mass * W);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          unsafeAtomicAdd(g + (1 + d) * NC, W * mass * (vel[d] + (C[d] * xi0 + C[3 + d] * xi1 + C[6 + d] * xi2)));
          unsafeAtomicAdd(g + (4 + d) * NC, (contrib[d] * xi0 + contrib[3 + d] * xi1 + contrib[6 + d] * xi2) * W);
        }
      }
}

template <int SIDE, int MODEL>
static __global__ __launch_bounds__(256) void p2g_global_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ps.n) return;
  const float dxi = mp.dxi;
  p2g_scatter_global<SIDE, MODEL>(mp, ps, i, t, grid, 4.f * dxi * dxi);
}

// ---- binned path
// LDS arena of one bin: 6^3 nodes, strides (floats) z + 8 y + 52 x: for a fixed stencil offset the 64 cells of
// a bin land on 32 distinct banks per 32-lane half.
struct ArenaLds {
  static constexpr int W = 6;
  static constexpr int SY = 8, SX = 52, CH = W * SX;
  __device__ static constexpr int at(int x, int y, int z) { return x * SX + y * SY + z; }
};

// LDS arena of one 8^3 block: 10^3 nodes (the block's cells + the two node layers of the quadratic stencil), dense
struct ArenaBlk {
  static constexpr int W = 10;
  static constexpr int SY = 10, SX = 100, CH = 1000;
  __device__ static constexpr int at(int x, int y, int z) { return x * SX + y * SY + z; }
};

// geometry of bin `bin`: grid block, origin of the bin inside the block (cells), origin in world cells
template <int SIDE> struct BinGeom {
  int block, o[3], org[3];
  __device__ __forceinline__ explicit BinGeom(int bin) {  // block and origin inside it; the caller fills org
    constexpr int BPB = bins_per_block<SIDE>();
    block = bin / BPB;
    const int sub = bin % BPB;
    o[0] = SIDE == 4 ? 0 : ((sub >> 2) & 1) * 4;
    o[1] = SIDE == 4 ? 0 : ((sub >> 1) & 1) * 4;
    o[2] = SIDE == 4 ? 0 : (sub & 1) * 4;
  }
  __device__ __forceinline__ BinGeom(const BhtDev &t, int bin, int kscale) : BinGeom(bin) {
#pragma unroll
    for (int d = 0; d < 3; ++d) org[d] = t.activeKeys[3 * (size_t)block + d] * (SIDE / kscale) + o[d];
  }
};

// arena node (x,y,z) of a bin -> (neighbour slot 0..7, cell id) in the grid block layout
template <int SIDE> __device__ __forceinline__ void arena_to_grid(const int (&o)[3], int x, int y, int z, int &slot, int &cell) {
  const int gx = o[0] + x, gy = o[1] + y, gz = o[2] + z;
  slot = ((gx >= SIDE) << 2) | ((gy >= SIDE) << 1) | (gz >= SIDE);
  cell = ((gx & (SIDE - 1)) * SIDE + (gy & (SIDE - 1))) * SIDE + (gz & (SIDE - 1));
}

// round-robin walk of one bin: round r visits the r-th particle of every cell (lane) that has one; the lanes
// that take part in a round read consecutive particles (coalesced), the index needs only ballots on the counts,
// so the loads of round r+1 can be issued before round r is computed (software pipelining: with ~220 VGPRs only
// two waves share a SIMD and memory latency must be hidden inside the wave).
struct RoundWalk {
  unsigned cnt, r = 0;
  int base;
  unsigned long long lt;
  __device__ __forceinline__ RoundWalk(unsigned cnt_, int start) : cnt(cnt_), base(start), lt(lanemask_lt()) {}
  // returns whether this lane has a particle in the next round; any = some lane has
  __device__ __forceinline__ bool next(int &i, bool &any) {
    const bool has = cnt > r;
    const unsigned long long m = __ballot(has);
    any = m != 0ull;
    i = base + __popcll(m & lt);
    base += __popcll(m);
    ++r;
    return has;
  }
};
template <int LW> struct RecA {  // sweep A inputs: x, v, C, m (16 floats)
  float pos[3], vel[3], C[9], mass;
  __device__ __forceinline__ void load(const ParticlesDev &ps, size_t i) {
    const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
    pload<LW, 3>(ps.pos, o, pos);
    pload<LW, 3>(ps.vel, o, vel);
    pload<LW, 9>(ps.C, o, C);
    mass = pload1<LW>(ps.mass, o);
  }
};
template <int MODEL, int LW> struct RecB {  // sweep B inputs: x, F (, logJp) -- or x, cached P F^T vol; fluid: x, J, C
  float pos[3], F[9], logJp;
  float C[model_is_fluid(MODEL) ? 9 : 1];
  __device__ __forceinline__ void load(const ParticlesDev &ps, size_t i) {
    const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
    pload<LW, 3>(ps.pos, o, pos);
    if constexpr (MODEL == MPM_CACHED_STRESS) {
      float S[STRESS_N];
      pload<LW, STRESS_N>(ps.stress, o, S);
      stress_unpack(S, F);
    } else pload_state<LW, model_is_fluid(MODEL)>(ps.F, o, F);
    if constexpr (model_uses_logjp(MODEL)) logJp = pload1<LW>(ps.logJp, o);
    if constexpr (MODEL == ZS_MPM_EQUATION_OF_STATE) pload<LW, 9>(ps.C, o, C);  // P2G sweep only (G2P recomputes C)
  }
};

template <int SIDE, int MODEL, int LW>
static __global__ __launch_bounds__(64, 2) void p2g_binned_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *binStart,
                                                        const unsigned *cellCount, const int *nbr, int *stale, int *staleCount) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float arena[7 * AL::CH];
  const int bin = (int)xcd_chunked(blockIdx.x, gridDim.x);
  const int start = binStart[bin], end = binStart[bin + 1];
  if (start == end) return;  // empty bin (ghost block): uniform exit
  const int lane = threadIdx.x;
  for (int k = lane; k < 7 * AL::CH; k += 64) arena[k] = 0.f;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const unsigned cnt = cellCount[(size_t)bin * 64 + lane];
  const float dxi = mp.dxi;
  const float D_inv = mp.D_inv;

  float *a0 = arena + AL::at(cx, cy, cz);
  __syncthreads();
  // Two sweeps over the bin's particles keep the register-resident stencil at 27 x 4 (mass, momentum) and
  // 27 x 3 (stress) accumulators instead of 27 x 7 = 189, which would cap occupancy at one wave per SIMD;
  // the price is reading x twice (+12 B/particle).
  {  // ---- sweep A: m, m v + m C (xi - xp)
    float acc[27][4];
#pragma unroll
    for (int k = 0; k < 27; ++k)
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) acc[k][ch] = 0.f;
    RoundWalk walk(cnt, start);
    int i0, i1;
    bool any, any1;
    bool has0 = walk.next(i0, any);
    RecA<LW> cur, nxt;
    if (has0) cur.load(ps, (size_t)i0);
    while (any) {
      const bool has1 = walk.next(i1, any1);
      if (has1) nxt.load(ps, (size_t)i1);  // in flight while the current round is computed
      if (has0) {
        Arena ar;
        make_arena(mp.dx, mp.dxi, cur.pos, ar);
        if (ar.corner[0] - geo.org[0] != cx || ar.corner[1] - geo.org[1] != cy || ar.corner[2] - geo.org[2] != cz) {
          stale[atomicAdd(staleCount, 1)] = i0;  // left its cell since the last re-binning: exact path afterwards
        } else {
          // W m (v + C (xi - xp)) is affine in the node offset: evaluate it as (Px[a] + Py[b]) + Pz[c] with the
          // per-axis products hoisted -> 8 VALU ops per node instead of ~20 (rounding differs from the
          // reference's association by O(1 ulp), inside the stated tolerance)
          float Px[3][3], Py[3][3], Pz[3][3], wzm[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float x0 = (float)k * mp.dx - ar.lp[0], x1 = (float)k * mp.dx - ar.lp[1], x2 = (float)k * mp.dx - ar.lp[2];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              Px[k][d] = cur.C[d] * x0;
              Py[k][d] = cur.C[3 + d] * x1;
              Pz[k][d] = fmaf(cur.C[6 + d], x2, cur.vel[d]);
            }
            wzm[k] = ar.w[2][k] * cur.mass;
          }
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
              const float wxy = ar.w[0][a] * ar.w[1][bb];
              const float q0 = Px[a][0] + Py[bb][0], q1 = Px[a][1] + Py[bb][1], q2 = Px[a][2] + Py[bb][2];
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float Wm = wxy * wzm[c];
                float(&A)[4] = acc[(a * 3 + bb) * 3 + c];
                A[0] += Wm;
                A[1] = fmaf(Wm, q0 + Pz[c][0], A[1]);
                A[2] = fmaf(Wm, q1 + Pz[c][1], A[2]);
                A[3] = fmaf(Wm, q2 + Pz[c][2], A[3]);
              }
            }
        }
      }
      cur = nxt;
      has0 = has1;
      i0 = i1;
      any = any1;
    }
    // 27 phases: in phase (a,b,c) lane (cx,cy,cz) owns node (cx+a, cy+b, cz+c) -- all 64 nodes distinct
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) g[ch * AL::CH] += acc[k][ch];
      __syncthreads();
    }
  }
  {  // ---- sweep B: rhs = -dt D_inv (P F^T vol) (xi - xp) W
    float acc[27][3];
#pragma unroll
    for (int k = 0; k < 27; ++k)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) acc[k][ch] = 0.f;
    RoundWalk walk(cnt, start);
    int i0, i1;
    bool any, any1;
    bool has0 = walk.next(i0, any);
    RecB<MODEL, LW> cur, nxt;
    if (has0) cur.load(ps, (size_t)i0);
    while (any) {
      const bool has1 = walk.next(i1, any1);
      if (has1) nxt.load(ps, (size_t)i1);
      if (has0) {
        Arena ar;
        make_arena(mp.dx, mp.dxi, cur.pos, ar);
        if (ar.corner[0] - geo.org[0] == cx && ar.corner[1] - geo.org[1] == cy && ar.corner[2] - geo.org[2] == cz) {
          float contrib[9];
          if constexpr (MODEL == MPM_CACHED_STRESS) {
#pragma unroll
            for (int d = 0; d < 9; ++d) contrib[d] = cur.F[d];
          } else {
            float lj = model_uses_logjp(MODEL) ? cur.logJp : 0.f;
            float Cp[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (model_is_fluid(MODEL)) {
#pragma unroll
              for (int d = 0; d < 9; ++d) Cp[d] = cur.C[d];
            }
            model_stress<MODEL>(mp.mat, lj, cur.F, contrib, Cp);
            if constexpr (model_uses_logjp(MODEL))
              pstore1<LW>(ps.logJp, particle_offset<LW>(ps.pos.chns, (size_t)i0), lj);  // P2G.hpp:101 (projected F not written back)
          }
#pragma unroll
          for (int d = 0; d < 9; ++d) contrib[d] = contrib[d] * -mp.dt * D_inv;
          float Qx[3][3], Qy[3][3], Qz[3][3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float x0 = (float)k * mp.dx - ar.lp[0], x1 = (float)k * mp.dx - ar.lp[1], x2 = (float)k * mp.dx - ar.lp[2];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              Qx[k][d] = contrib[d] * x0;
              Qy[k][d] = contrib[3 + d] * x1;
              Qz[k][d] = contrib[6 + d] * x2;
            }
          }
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
              const float wxy = ar.w[0][a] * ar.w[1][bb];
              const float q0 = Qx[a][0] + Qy[bb][0], q1 = Qx[a][1] + Qy[bb][1], q2 = Qx[a][2] + Qy[bb][2];
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float Wt = wxy * ar.w[2][c];
                float(&A)[3] = acc[(a * 3 + bb) * 3 + c];
                A[0] = fmaf(Wt, q0 + Qz[c][0], A[0]);
                A[1] = fmaf(Wt, q1 + Qz[c][1], A[1]);
                A[2] = fmaf(Wt, q2 + Qz[c][2], A[2]);
              }
            }
        }
      }
      cur = nxt;
      has0 = has1;
      i0 = i1;
      any = any1;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3) + 4 * AL::CH;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) g[ch * AL::CH] += acc[k][ch];
      __syncthreads();
    }
  }
  // flush: consecutive lanes -> consecutive z of one (channel, x, y) row
  int nb[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) nb[o] = nbr[(size_t)geo.block * 8 + o];
  for (int k = lane; k < 7 * 216; k += 64) {
    const int ch = k / 216, node = k % 216;
    const int x = node / 36, y = (node / 6) % 6, z = node % 6;
    const float v = arena[ch * AL::CH + AL::at(x, y, z)];
    if (v == 0.f) continue;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    int bn = nb[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) bn = (slot == q) ? nb[q] : bn;
    if (bn < 0) continue;
    unsafeAtomicAdd(grid + ((size_t)bn * 7 + ch) * NC + cell, v);
  }
}


// ---- "wide" cached-stress P2G: ONE wave per bin carries all 7 channels (27 x 7 = 189 register accumulators).
// The four-wave split above repeats the arena / weight / address work in every wave (PMC: 1113 VALU instructions per
// 64-particle round, SQ_INSTS_VALU x 4 cycles = 96 % of the SIMD cycles: that kernel is VALU-bound).  Here the per-particle
// work is done once (~600 VALU per round).  The price is 2 waves per SIMD; the latency the occupancy no longer hides is
// covered by asynchronous global -> LDS loads (global_load_lds_dword: no staging VGPRs) issued one round ahead into a
// double-buffered record area of the LDS.
constexpr int P2GW_NF = 16 + STRESS_N;  // rows of a record: m, x, v, C, symmetric stress
constexpr int P2GW_MQ_CAP = 256;  // in-bin movers the wide P2G takes through its LDS queue  // m, x(3), v(3), C(9), P F^T vol(9)

// `tileBase`: wave-uniform element offset of a tile at or before the bin's first particle.  The per-lane part of every address
// is then a 32-bit byte offset from a scalar base (global_load_lds_dword v_off, s[base:base+1] offset:imm): ONE address VGPR
// per round instead of a 64-bit pointer per attribute.
#ifndef ZS_P2GW_AUX
#define ZS_P2GW_AUX 0  // cache-policy bits of the record loads (measurement builds: 2 = nt)
#endif
template <int LW>
__device__ __forceinline__ void p2gw_issue(const ParticlesDev &ps, size_t i, bool has, float *buf, size_t tileBase) {
  // every lane of the wave executes the 25 instructions (LDS destination = wave-uniform row + lane * 4); lanes without a
  // particle in this round are masked off by exec
  if (has) {
    const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
    const unsigned voff = LW != 0 ? (unsigned)((o.o - tileBase) * sizeof(float)) : 0u;
    auto ptr = [&](const Port<float> &p, int comp) -> const float * {
      if constexpr (LW != 0)
        return reinterpret_cast<const float *>(reinterpret_cast<const char *>(p.base + tileBase) + (size_t)voff) + (size_t)comp * LW;
      else
        return p.base + p.off(o.o) + (size_t)comp * p.cstride();
    };
    __builtin_amdgcn_global_load_lds(ptr(ps.mass, 0), (__attribute__((address_space(3))) void *)(buf + 0 * 64), 4, 0, ZS_P2GW_AUX);
#pragma unroll
    for (int d = 0; d < 3; ++d) __builtin_amdgcn_global_load_lds(ptr(ps.pos, d), (__attribute__((address_space(3))) void *)(buf + (1 + d) * 64), 4, 0, ZS_P2GW_AUX);
#pragma unroll
    for (int d = 0; d < 3; ++d) __builtin_amdgcn_global_load_lds(ptr(ps.vel, d), (__attribute__((address_space(3))) void *)(buf + (4 + d) * 64), 4, 0, ZS_P2GW_AUX);
#pragma unroll
    for (int d = 0; d < 9; ++d) __builtin_amdgcn_global_load_lds(ptr(ps.C, d), (__attribute__((address_space(3))) void *)(buf + (7 + d) * 64), 4, 0, ZS_P2GW_AUX);
#pragma unroll
    for (int d = 0; d < STRESS_N; ++d) __builtin_amdgcn_global_load_lds(ptr(ps.stress, d), (__attribute__((address_space(3))) void *)(buf + (16 + d) * 64), 4, 0, ZS_P2GW_AUX);
  }
}
template <int LW> __device__ __forceinline__ size_t p2gw_tile_base(const ParticlesDev &ps, int start) {
  if constexpr (LW != 0) return ((size_t)start / LW) * (size_t)ps.pos.chns * LW;
  else return 0;
}

// one particle record (LDS row layout of p2gw_issue) -> the lane's 27 x 7 register stencil.
// r05, "Q form" (see stage_qform): per vector channel the value at the stencil's centre node and its change per node step,
//   momentum d: alpha = m (v_d + C[d, :] . (dx - lp)), b_k = m dx C[d + 3 k];  force d: alpha = kscale S[d, :] . (dx - lp), b_k = kscale dx S[d, k]
// (S = the symmetric P F^T vol), then per node W_abc (alpha + (a - 1) bx + (b - 1) by + (c - 1) bz): the offsets x_i - x_p and the products
// C (x_i - x_p) are no longer rebuilt per node, and ONE weight product W_abc serves all seven channels.
__device__ __forceinline__ void p2gw_accumulate(const MpmDev &mp, const Arena &ar, const float *rec, float kscale, float (&acc)[27][7]) {
  const float m = rec[0];
  float lc[3];  // centre node - particle
#pragma unroll
  for (int k = 0; k < 3; ++k) lc[k] = mp.dx - ar.lp[k];
  float al[6], bx[6], by[6], bz[6];
  {
    const float mdx = m * mp.dx, ksdx = kscale * mp.dx;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = rec[(4 + d) * 64], c0 = rec[(7 + d) * 64], c1 = rec[(10 + d) * 64], c2 = rec[(13 + d) * 64];
      al[d] = m * (v + (c0 * lc[0] + c1 * lc[1] + c2 * lc[2]));
      bx[d] = mdx * c0;
      by[d] = mdx * c1;
      bz[d] = mdx * c2;
      // row d of the symmetric P F^T vol {xx, xy, xz, yy, yz, zz} (rows 16..21 of the record)
      const float s0 = rec[(16 + d) * 64], s1 = rec[(16 + (d == 0 ? 1 : d == 1 ? 3 : 4)) * 64], s2 = rec[(16 + (d == 0 ? 2 : d == 1 ? 4 : 5)) * 64];
      al[3 + d] = kscale * (s0 * lc[0] + s1 * lc[1] + s2 * lc[2]);
      bx[3 + d] = ksdx * s0;
      by[3 + d] = ksdx * s1;
      bz[3 + d] = ksdx * s2;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float qa[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) qa[j] = a == 0 ? al[j] - bx[j] : (a == 1 ? al[j] : al[j] + bx[j]);
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float wxy = ar.w[0][a] * ar.w[1][bb];
      const float W0 = wxy * ar.w[2][0], W1 = wxy * ar.w[2][1], W2 = wxy * ar.w[2][2];
      float(&A0)[7] = acc[(a * 3 + bb) * 3], (&A1)[7] = acc[(a * 3 + bb) * 3 + 1], (&A2)[7] = acc[(a * 3 + bb) * 3 + 2];
      A0[0] = fmaf(W0, m, A0[0]);
      A1[0] = fmaf(W1, m, A1[0]);
      A2[0] = fmaf(W2, m, A2[0]);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float qab = bb == 0 ? qa[j] - by[j] : (bb == 1 ? qa[j] : qa[j] + by[j]);
        A0[1 + j] = fmaf(W0, qab - bz[j], A0[1 + j]);
        A1[1 + j] = fmaf(W1, qab, A1[1 + j]);
        A2[1 + j] = fmaf(W2, qab + bz[j], A2[1 + j]);
      }
    }
  }
}

// LDS arena shared by the G bins of one workgroup of p2g_wide_kernel: G = 1 one bin (6^3 nodes, ArenaLds), G = 2 the two bins of a
// block that are neighbours in z (4 x 4 x 8 cells, 6 x 6 x 10 nodes), G = 4 the four bins of a half block (4 x 8 x 8 cells, 6 x 10 x 10
// nodes).  Strides from a search over (SY, SX): for a fixed stencil offset the 64 cells of a bin land on 32 distinct banks per half wave.
template <int G> struct ArenaLdsG;
template <> struct ArenaLdsG<1> {
  static constexpr int WX = 6, WY = 6, WZ = 6, SY = ArenaLds::SY, SX = ArenaLds::SX, CH = WX * SX;
  __device__ static constexpr int at(int x, int y, int z) { return x * SX + y * SY + z; }
};
template <> struct ArenaLdsG<2> {
  static constexpr int WX = 6, WY = 6, WZ = 10, SY = 12, SX = 80, CH = WX * SX;
  __device__ static constexpr int at(int x, int y, int z) { return x * SX + y * SY + z; }
};
template <> struct ArenaLdsG<4> {
  static constexpr int WX = 6, WY = 10, WZ = 10, SY = 12, SX = 144, CH = WX * SX;
  __device__ static constexpr int at(int x, int y, int z) { return x * SX + y * SY + z; }
};

// tail shared by the wide P2G kernels: the queued in-bin movers go into the arena by LDS atomics (same values as the exact path), then
// the arena goes to the grid -- origin of the workgroup's arena inside its block = the origin of its first bin
template <int SIDE, int G>
__device__ __forceinline__ void p2gw_movers_and_flush(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, float *arena, const int *mqw,
                                                      int mqCountW, int lane, int ay, int az, const int *nbr, float *grid) {
  using AL = ArenaLdsG<G>;
  constexpr int NC = SIDE * SIDE * SIDE;
  const float kscale = mp.fscale;
  {  // post-pass: the queued in-bin particles, one lane each, added to the arena with LDS atomics (same values as the exact path)
    const int nm = mqCountW < P2GW_MQ_CAP ? mqCountW : P2GW_MQ_CAP;
    for (int q = lane; q < nm; q += 64) {
      const size_t i = (size_t)mqw[q];
      float pos[3], vel[3], C[9], PF[9];
      load_attr<3>(ps.pos, i, pos);
      load_attr<3>(ps.vel, i, vel);
      load_attr<9>(ps.C, i, C);
      {
        float S[STRESS_N];
        load_attr<STRESS_N>(ps.stress, i, S);
        stress_unpack(S, PF);
      }
      const float m = ps.mass.base[ps.mass.off(i)];
#pragma unroll
      for (int d = 0; d < 9; ++d) PF[d] *= kscale;
      Arena ar;
      make_arena(mp.dx, mp.dxi, pos, ar);
      float *b0 = arena + AL::at(ar.corner[0] - geo.org[0], ar.corner[1] - geo.org[1] + ay, ar.corner[2] - geo.org[2] + az);
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
  // flush: origin of the workgroup's arena inside its block = the origin of its first bin
  const int o0[3] = {geo.o[0], geo.o[1] - ay, geo.o[2] - az};
  for (int node = threadIdx.x; node < AL::WX * AL::WY * AL::WZ; node += 64 * G) {
    const int x = node / (AL::WY * AL::WZ), y = (node / AL::WZ) % AL::WY, z = node % AL::WZ;
    int slot2, cell;
    arena_to_grid<SIDE>(o0, x, y, z, slot2, cell);
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

#ifdef ZS_PROBE_P2G  // measurement-only build (tools/ab_build.sh): s_memtime stamps of a wave's phases in p2g_wide_kernel, summed over the sampled waves
__device__ unsigned long long g_p2g_probe[16];
#define P2G_NOW() ((unsigned long long)__builtin_readcyclecounter())
#define P2G_STAMP(slot, dt) do { if (pSample) atomicAdd(&g_p2g_probe[slot], (unsigned long long)(dt)); } while (0)
#else
#define P2G_NOW() 0ull
#define P2G_STAMP(slot, dt) do { } while (0)
#endif
// DEPTH = rounds of records in flight ahead of the one being computed (DEPTH + 1 LDS buffers of P2GW_NF x 256 B per wave).
// G = bins (= waves) per workgroup.  Every wave streams its own bin exactly as a one-wave workgroup would (nothing is shared while the
// records flow); what the G waves share is the flush: their register stencils go into ONE arena and the workgroup issues one set of
// global float atomics for it.  The atomics are what the kernel writes (every atomic instruction writes the 32-byte sectors it touches
// through to memory, whatever the launch order: profiles/r04_launch_order.md), and neighbouring bins' aprons overlap: per 8^3 block
// 8 x 7 x 36 rows x 1.5 sectors = 3024 sector writes with G = 1, 2016 with G = 2 (a row of 10 z-nodes = the block's own 32-byte row + 8
// bytes of the next block's), 1680 with G = 4.
template <int SIDE, int LW, int DEPTH, int G>
static __global__ __launch_bounds__(64 * G, 2) void p2g_wide_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *binStart,
                                                           const unsigned *cellCount, const int *nbr, int *stale, int *staleCount) {
  static_assert(G == 1 || (SIDE == 8 && (G == 2 || G == 4)), "G bins of one block");
  using AL = ArenaLdsG<G>;
  constexpr int NC = SIDE * SIDE * SIDE;
  constexpr int NB = DEPTH + 1;
  constexpr int WBUF = NB * P2GW_NF * 64;  // floats of record buffers per wave
  // the record buffers and the flush arena are never live at the same time: one LDS region serves both
  constexpr int LDSF = G * WBUF > 7 * AL::CH ? G * WBUF : 7 * AL::CH;
  __shared__ float lds[LDSF];
  __shared__ int mq[G][P2GW_MQ_CAP];  // particles that sit in another cell of their bin (moved since the last re-bin)
  __shared__ int mqCount[G];
  const int w = G == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (w in an SGPR)
  const int bin0 = (int)xcd_chunked(blockIdx.x, gridDim.x) * G, bin = bin0 + w;
#ifdef ZS_PROBE_P2G
  const bool pSample = lane == 0 && (blockIdx.x & 15) == 0;
  const unsigned long long tp0 = P2G_NOW();
  unsigned long long tpWait = 0, tpRounds = 0;
#endif
  if (binStart[bin0] == binStart[bin0 + G]) return;  // none of the G bins holds a particle (workgroup-uniform)
  float *arena = lds;
  float(*pbuf)[P2GW_NF * 64] = reinterpret_cast<float(*)[P2GW_NF * 64]>(lds + w * WBUF);
  const int start = binStart[bin], end = binStart[bin + 1];
  if (lane == 0) mqCount[w] = 0;
  __syncthreads();
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const unsigned cnt = start == end ? 0u : cellCount[(size_t)bin * 64 + lane];
  const float dxi = mp.dxi;
  const float kscale = mp.fscale;  // contrib = -dt D_inv (P F^T vol)
  float acc[27][7];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) acc[k][ch] = 0.f;
  // two walks over the same counts: `lead` runs DEPTH rounds ahead and issues the loads, `walk` consumes
  const size_t tileBase = p2gw_tile_base<LW>(ps, start);
  RoundWalk lead(cnt, start), walk(cnt, start);
  int li;
  bool lany = true;
  int issued = 0;  // rounds issued and not yet consumed (wave-uniform)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    if (lany) {
      const bool lh = lead.next(li, lany);
      if (lany) {
        p2gw_issue<LW>(ps, (size_t)li, lh, pbuf[d % NB], tileBase);
        ++issued;
      }
    }
  }
  int slot = 0, lslot = DEPTH % NB;
  int i0;
  bool any;
  bool has0 = walk.next(i0, any);
#ifdef ZS_PROBE_P2G
  const unsigned long long tp1 = P2G_NOW();  // head done: counts known, first DEPTH rounds requested
#endif
  while (any) {
    if (lany) {
      const bool lh = lead.next(li, lany);
      if (lany) {
        p2gw_issue<LW>(ps, (size_t)li, lh, pbuf[lslot], tileBase);
        lslot = lslot + 1 == NB ? 0 : lslot + 1;
        ++issued;
      }
    }
    // wait until only the records issued AFTER the current one are still in flight
    // (a record is P2GW_NF loads; vmcnt holds 6 bits: two records in flight is the most that can be told apart)
    static_assert(2 * P2GW_NF <= 63, "vmcnt range");
#ifdef ZS_PROBE_P2G
    const unsigned long long tw0 = P2G_NOW();
#endif
    if (issued >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P2GW_NF) : "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P2GW_NF) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ZS_PROBE_P2G
    tpWait += P2G_NOW() - tw0;
    ++tpRounds;
#endif
    if (has0) {
      const float *rec = pbuf[slot] + lane;
      const float pos[3] = {rec[1 * 64], rec[2 * 64], rec[3 * 64]};
      Arena ar;
      make_arena(mp.dx, mp.dxi, pos, ar);
      const int ocx = ar.corner[0] - geo.org[0], ocy = ar.corner[1] - geo.org[1], ocz = ar.corner[2] - geo.org[2];
      if (ocx == cx && ocy == cy && ocz == cz) {
        p2gw_accumulate(mp, ar, rec, kscale, acc);
      } else {
        // another cell of the same bin: queued for the post-pass into the arena; outside the bin: exact path afterwards
        bool queued = false;
        if ((unsigned)ocx < 4u && (unsigned)ocy < 4u && (unsigned)ocz < 4u) {
          const int q = atomicAdd(&mqCount[w], 1);
          if (q < P2GW_MQ_CAP) {
            mq[w][q] = i0;
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
#ifdef ZS_PROBE_P2G
  const unsigned long long tp2 = P2G_NOW();  // stream done
#endif
  __syncthreads();  // every record of every wave has been consumed: the region becomes the arena
  for (int k = threadIdx.x; k < 7 * AL::CH; k += 64 * G) arena[k] = 0.f;
  __syncthreads();
#ifdef ZS_PROBE_P2G
  const unsigned long long tp3 = P2G_NOW();  // waited for the other waves, arena cleared
#endif
  // this bin's corner inside the workgroup's arena: the G bins differ in z (G = 2) or in y and z (G = 4)
  const int ay = G == 4 ? (w >> 1) * 4 : 0, az = G == 1 ? 0 : (w & 1) * 4;
  float *a0 = arena + AL::at(cx, cy + ay, cz + az);
#pragma unroll
  for (int k = 0; k < 27; ++k) {  // 27 conflict-free phases: in a phase the 64 G lanes of the workgroup own 64 G distinct nodes
    float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) g[ch * AL::CH] += acc[k][ch];
    if constexpr (G == 1) __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations execute in order
    else __syncthreads();
  }
  __syncthreads();
#ifdef ZS_PROBE_P2G
  const unsigned long long tp4 = P2G_NOW();  // 27 phases done
#endif
  p2gw_movers_and_flush<SIDE, G>(mp, ps, geo, arena, mq[w], mqCount[w], lane, ay, az, nbr, grid);
#ifdef ZS_PROBE_P2G
  {
    const unsigned long long tp5 = P2G_NOW();  // atomics issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tp6 = P2G_NOW();
    P2G_STAMP(0, tp6 - tp0);   // life of the wave
    P2G_STAMP(1, tp1 - tp0);   // head: bin range, cell counts, block key, first requests
    P2G_STAMP(2, tp2 - tp1);   // record stream (all rounds)
    P2G_STAMP(3, tpWait);      // ... of which inside s_waitcnt vmcnt
    P2G_STAMP(4, tp3 - tp2);   // barrier with the group's other waves + arena clear
    P2G_STAMP(5, tp4 - tp3);   // 27 register -> arena phases
    P2G_STAMP(6, tp5 - tp4);   // movers' post-pass + arena -> grid atomics issued
    P2G_STAMP(7, tp6 - tp5);   // atomics drained
    P2G_STAMP(8, 1);           // sampled waves
    P2G_STAMP(9, tpRounds);    // rounds
  }
#endif
}

// The lane's 27 x 7 node sums of p2g_tile_kernel with the six vector channels as three register PAIRS per node, {mv_x, mv_y},
// {mv_z, f_x}, {f_y, f_z}: the Q-form accumulation then issues one v_pk_add_f32 / v_pk_fma_f32 where p2gw_accumulate issues two scalar
// instructions (the node's weight is broadcast from the low half of its pair: op_sel_hi:[0,1,1]).  A wave issues one instruction every
// ~5 cycles whatever it is (profiles/r02_valu_opcode_rates.md) and this kernel has two waves per SIMD, so its record stream is bound by
// the NUMBER of instructions a wave issues, not by the VALU rate (where a packed instruction costs 1.7 scalar ones: r03, fused kernels).
// Each half is the same IEEE operation as the scalar form; the products are formed in the same order as in p2gw_accumulate.
typedef float p2g_f2 __attribute__((ext_vector_type(2)));
struct P2GAccPk {
  float m[27];
  p2g_f2 q[27][3];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      m[k] = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) q[k][j] = p2g_f2{0.f, 0.f};
    }
  }
  __device__ __forceinline__ float get(int k, int ch) const { return ch == 0 ? m[k] : (((ch - 1) & 1) ? q[k][(ch - 1) >> 1].y : q[k][(ch - 1) >> 1].x); }
};
// `own` false: the lane adds zeros (its record and its weights are replaced by zeros first: they may be anything, NaN included) -- the
// accumulation is NOT a divergent region, whose join would copy every register pair.
__device__ __forceinline__ void p2gw_accumulate_pk(const MpmDev &mp, const Arena &ar0, const float *rec0, float kscale, bool own, P2GAccPk &A) {
  Arena ar;
  float recv[P2GW_NF];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    ar.lp[d] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ar.w[d][k] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < P2GW_NF; ++r) recv[r] = 0.f;
  if (own) {  // ONE divergent region: the record's reads and the copies of the weights
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ar.lp[d] = ar0.lp[d];
#pragma unroll
      for (int k = 0; k < 3; ++k) ar.w[d][k] = ar0.w[d][k];
    }
    recv[0] = rec0[0];
#pragma unroll
    for (int r = 4; r < P2GW_NF; ++r) recv[r] = rec0[r * 64];
  }
  struct { const float *v; __device__ __forceinline__ float operator[](int i) const { return v[i / 64]; } } rec{recv};
  const float m = rec[0];
  float lc[3];  // centre node - particle
#pragma unroll
  for (int k = 0; k < 3; ++k) lc[k] = mp.dx - ar.lp[k];
  float al[6], bx[6], by[6], bz[6];
  {
    const float mdx = m * mp.dx, ksdx = kscale * mp.dx;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = rec[(4 + d) * 64], c0 = rec[(7 + d) * 64], c1 = rec[(10 + d) * 64], c2 = rec[(13 + d) * 64];
      al[d] = m * (v + (c0 * lc[0] + c1 * lc[1] + c2 * lc[2]));
      bx[d] = mdx * c0;
      by[d] = mdx * c1;
      bz[d] = mdx * c2;
      const float s0 = rec[(16 + d) * 64], s1 = rec[(16 + (d == 0 ? 1 : d == 1 ? 3 : 4)) * 64], s2 = rec[(16 + (d == 0 ? 2 : d == 1 ? 4 : 5)) * 64];
      al[3 + d] = kscale * (s0 * lc[0] + s1 * lc[1] + s2 * lc[2]);
      bx[3 + d] = ksdx * s0;
      by[3 + d] = ksdx * s1;
      bz[3 + d] = ksdx * s2;
    }
  }
  p2g_f2 alp[3], bxp[3], byp[3], bzp[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    alp[j] = p2g_f2{al[2 * j], al[2 * j + 1]};
    bxp[j] = p2g_f2{bx[2 * j], bx[2 * j + 1]};
    byp[j] = p2g_f2{by[2 * j], by[2 * j + 1]};
    bzp[j] = p2g_f2{bz[2 * j], bz[2 * j + 1]};
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p2g_f2 qa[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) qa[j] = a == 0 ? alp[j] - bxp[j] : (a == 1 ? alp[j] : alp[j] + bxp[j]);
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float wxy = ar.w[0][a] * ar.w[1][bb];
      const float W0 = wxy * ar.w[2][0], W1 = wxy * ar.w[2][1], W2 = wxy * ar.w[2][2];
      const int k0 = (a * 3 + bb) * 3;
      A.m[k0] = fmaf(W0, m, A.m[k0]);
      A.m[k0 + 1] = fmaf(W1, m, A.m[k0 + 1]);
      A.m[k0 + 2] = fmaf(W2, m, A.m[k0 + 2]);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const p2g_f2 qab = bb == 0 ? qa[j] - byp[j] : (bb == 1 ? qa[j] : qa[j] + byp[j]);
        A.q[k0][j] = __builtin_elementwise_fma((p2g_f2)(W0), qab - bzp[j], A.q[k0][j]);
        A.q[k0 + 1][j] = __builtin_elementwise_fma((p2g_f2)(W1), qab, A.q[k0 + 1][j]);
        A.q[k0 + 2][j] = __builtin_elementwise_fma((p2g_f2)(W2), qab + bzp[j], A.q[k0 + 2][j]);
      }
    }
  }
}

// private flush arena of one wave of p2g_tile_kernel: 6^3 nodes, one float4 per node and plane (plane 0: m, mv; plane 1: f), strides in
// nodes z + 12 y + 72 x: a 16-byte access of the wave is served in four passes of 16 lanes = the 4 x 4 (y, z) cells of one x, and
// 12 y + z (+ a phase offset) takes 16 distinct values mod 16 there -- no bank conflict in any of the 27 phases.
struct ArenaPriv {
  static constexpr int SY = 12, SX = 72, PLANE = 6 * SX;
  __device__ static constexpr int at(int x, int y, int z) { return x * SX + y * SY + z; }
};

// ---- tile-stream variant of the wide P2G (LW = 64 only): the record loads are decoupled from the rounds.
// A bin's particles are the contiguous range [start, end) of the compact order, i.e. the tiles start / 64 .. (end - 1) / 64 of the AoSoA
// container.  The wave requests WHOLE TILES (22 rows x 256 B by 6 - 8 `global_load_lds_dwordx4` of 1 KiB each instead of 22 dword
// requests per round of ~42 particles: an LDS-direct load costs the wave tens of cycles of issue whatever its width) into a ring of NB tile buffers as soon as
// the bin's range is known -- before its cell counts arrive, so the head of a wave is ONE memory round trip instead of two -- and a
// round's lane reads its particle at ring position (index mod 64) of tile (index / 64).  A tile buffer is re-requested when the walk has
// passed the tile's last particle: NB - 1 tiles (1.5 - 3 rounds) stay in flight ahead of the round being accumulated.
// One request = one `global_load_lds_dwordx4`: lane l moves 16 bytes from (row base + 16 l) to (LDS base + 16 l), i.e. FOUR consecutive
// 256-byte channel rows of the tile per wave-instruction (1 KiB), and the instruction offset advances both addresses.  MERGED: the host
// found m, x, v, C in 16 adjacent channels (the layout of zpc_amd.mpm and of the reference's particles TileVector {m, x, v, C, ...}):
// 4 + 2 requests per tile; otherwise one base per attribute, 8 requests (the last of an attribute with the lanes of its remaining rows).
// The requests are inline assembly on purpose: for the builtin the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read that
// follows an LDS-direct load it cannot tell apart (SIInsertWaitcnts: any DS read may alias a pending LDS-DMA write), i.e. in front of
// the first record read of EVERY round -- the ring would never hold a tile in flight.  The kernel orders reads behind arrivals itself
// (the vmcnt switch in front of a round), so the compiler must not know these are loads into LDS.  M0 = LDS base of the request.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void p2gt_dma4(unsigned long long sbase, unsigned voff, unsigned ldsAddr) {  // one dword per lane
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(ldsAddr) : "memory", "m0");
}
template <int OFF0, int CNT> __device__ __forceinline__ void p2gt_dma16_run(unsigned long long sbase, unsigned voff, unsigned ldsAddr) {
  // CNT requests 1 KiB apart (global and LDS address advance together through the instruction offset), one M0 write
  static_assert(CNT >= 1 && CNT <= 4, "instruction offsets up to 3072");
  if constexpr (CNT == 1)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(ldsAddr), "n"(OFF0) : "memory", "m0");
  else if constexpr (CNT == 2)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4" ::"v"(voff), "s"(sbase), "s"(ldsAddr),
                 "n"(OFF0), "n"(OFF0 + 1024) : "memory", "m0");
  else if constexpr (CNT == 3)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4\n\tglobal_load_lds_dwordx4 %0, %1 offset:%5" ::"v"(voff),
                 "s"(sbase), "s"(ldsAddr), "n"(OFF0), "n"(OFF0 + 1024), "n"(OFF0 + 2048) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4\n\tglobal_load_lds_dwordx4 %0, %1 offset:%5\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:%6" ::"v"(voff), "s"(sbase), "s"(ldsAddr), "n"(OFF0), "n"(OFF0 + 1024), "n"(OFF0 + 2048), "n"(OFF0 + 3072) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int ROW0, int N>
__device__ __forceinline__ void p2gt_issue_attr(const Port<float> &p, size_t tb, int lane, float *buf) {
  // wave-uniform row base in an SGPR pair + one 32-bit lane offset
  const unsigned long long ub = (unsigned long long)(p.base + tb);
  const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub);
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(buf + ROW0 * 64);
  constexpr int FULL = N / 4, REST = N % 4;
  static_assert(FULL <= 4, "up to 16 + 3 rows per base");
  if constexpr (FULL > 0) p2gt_dma16_run<0, FULL>(sb, voff, l);
  if constexpr (REST > 0)
    if (lane < REST * 16) p2gt_dma16_run<FULL * 1024, 1>(sb, voff, l);
}
template <bool MERGED> constexpr int p2gt_requests() { return MERGED ? 4 + (STRESS_N + 3) / 4 : 1 + 1 + 1 + 3 + (STRESS_N + 3) / 4; }
// [lo, hi): the particles of the tile that belong to this bin (0, 64 for an inner tile).  A lane moves the 4 particles 4 (lane mod 16) ...
// + 3 of a row; lanes whose four lie outside the range stay out of ALL the tile's requests, so that a 128-byte line of a boundary tile that
// only the neighbouring bin needs is not fetched here as well (the whole-tile form read 4 % more than the records: profiles/r06_pmc_p2g.md).
template <bool MERGED>
__device__ __forceinline__ void p2gt_issue(const ParticlesDev &ps, int tile, int lo, int hi, int lane, float *buf) {
  // element offset of the tile (wave-uniform); the stress attribute may live in a TileVector of its own (another channel count: the
  // reference-order P2G keeps it in a temporary, see zs_rocm_mpm_p2g)
  const size_t tb = (size_t)tile * (size_t)ps.pos.chns * 64, tbs = (size_t)tile * (size_t)ps.stress.chns * 64;
  const int pl = (lane & 15) * 4;
  if (pl + 3 >= lo && pl < hi) {
    if constexpr (MERGED) {
      p2gt_issue_attr<0, 16>(ps.mass, tb, lane, buf);
    } else {
      p2gt_issue_attr<0, 1>(ps.mass, tb, lane, buf);
      p2gt_issue_attr<1, 3>(ps.pos, tb, lane, buf);
      p2gt_issue_attr<4, 3>(ps.vel, tb, lane, buf);
      p2gt_issue_attr<7, 9>(ps.C, tb, lane, buf);
    }
    p2gt_issue_attr<16, STRESS_N>(ps.stress, tbs, lane, buf);
  }
}

template <int SIDE, int NB, int G, bool MERGED>
static __global__ __launch_bounds__(64 * G, 2) void p2g_tile_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *binStart,
                                                           const unsigned *cellCount, const int *nbr, int *stale, int *staleCount) {
  static_assert(G == 1 || (SIDE == 8 && (G == 2 || G == 4)), "G bins of one block");
  constexpr int NL = p2gt_requests<MERGED>();  // load instructions per tile
  static_assert((NB - 1) * NL <= 63, "vmcnt range");
  using AL = ArenaLdsG<G>;
  constexpr int NC = SIDE * SIDE * SIDE;
  constexpr int TILEF = P2GW_NF * 64;     // floats of one tile buffer
  constexpr int WBUF = NB * TILEF;        // ... of a wave's ring
  constexpr int LDSF = G * WBUF > 7 * AL::CH ? G * WBUF : 7 * AL::CH;  // ring and flush arena are never live at the same time
  __shared__ float lds[LDSF];
  __shared__ int mq[G][P2GW_MQ_CAP];
  __shared__ int mqCount[G];
  __shared__ unsigned cntLds[G][64];
  const int w = G == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bin0 = (int)xcd_chunked(blockIdx.x, gridDim.x) * G, bin = bin0 + w;
#ifdef ZS_PROBE_P2G
  const bool pSample = lane == 0 && (blockIdx.x & 15) == 0;
  const unsigned long long tp0 = P2G_NOW();
  unsigned long long tpWait = 0, tpRounds = 0;
#endif
  // the G + 1 range words of the workgroup's bins and this wave's cell counts are requested together
  int bs[G + 1];
#pragma unroll
  for (int k = 0; k <= G; ++k) bs[k] = binStart[bin0 + k];
  int start = bs[0], end = bs[1];
#pragma unroll
  for (int k = 1; k < G; ++k)
    if (w == k) start = bs[k], end = bs[k + 1];
  // the wave's cell counts (all zero for an empty bin) come through LDS like the tiles, requested in front of them: a load into a
  // register that the compiler tracks would get its `s_waitcnt vmcnt(n)` computed without the tile requests behind it (i.e. wait for
  // every tile requested so far), and one it does not track could be copied before it has landed
  {
    const unsigned long long cb = (unsigned long long)(cellCount + (size_t)bin * 64);
    const unsigned long long scb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cb >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cb);
    p2gt_dma4(scb, (unsigned)lane * 4u, (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)cntLds[w]);
  }
  float *arena = lds;
  float *ring = lds + w * WBUF;
  const int tile0 = start >> 6, tileEnd = (end + 63) >> 6;  // the bin's tiles (none if start == end)
  int tIssue = tile0;  // next tile to request; its buffer is ring[(tIssue - tile0) % NB] = islot
  int islot = 0;
  auto request = [&]() {
    const int lo = tIssue == tile0 ? (start & 63) : 0, hi = tIssue == tileEnd - 1 ? end - (tIssue << 6) : 64;
    p2gt_issue<MERGED>(ps, tIssue, lo, hi, lane, ring + islot * TILEF);
    ++tIssue;
    islot = islot + 1 == NB ? 0 : islot + 1;
  };
  if (start != end) {
#pragma unroll 1
    for (int k = 0; k < NB; ++k)
      if (tIssue < tileEnd) request();
  }
  if (bs[0] == bs[G]) return;  // none of the G bins holds a particle (workgroup-uniform)
  if (lane == 0) mqCount[w] = 0;  // (only this wave touches mq[w] / mqCount[w]: its LDS operations execute in order)
  BinGeom<SIDE> geo(bin);
  {  // the block's key by scalar loads (constant address space + uniform address)
    const auto *ak = reinterpret_cast<const __attribute__((address_space(4))) int *>(reinterpret_cast<unsigned long long>(t.activeKeys));
#pragma unroll
    for (int d = 0; d < 3; ++d) geo.org[d] = ak[3 * (size_t)geo.block + d] * (SIDE / mp.kscale) + geo.o[d];
  }
  // the block's 8 neighbour numbers {+0, +1}^3 for the flush: wave-uniform, requested now (scalar loads) instead of one dependent
  // vector load per flushed node at the end of the wave's life
  int nbs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)  // (constant address space + uniform address = s_load: vmcnt stays the record requests' own)
    nbs[k] = reinterpret_cast<const __attribute__((address_space(4))) int *>(reinterpret_cast<unsigned long long>(nbr))[(size_t)geo.block * 8 + k];
  const float kscale = mp.fscale;  // contrib = -dt D_inv (P F^T vol)
#ifdef ZS_P2GT_SCALAR  // measurement builds: the scalar accumulation of p2g_wide_kernel
  struct { float a[27][7]; __device__ __forceinline__ float get(int k, int ch) const { return a[k][ch]; } } acc;
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) acc.a[k][ch] = 0.f;
#else
  P2GAccPk acc;
  acc.clear();
#endif
  int base = start;      // first particle of the round (wave-uniform)
  int tDone = tile0;     // tiles below have arrived
  int cslot = 0;         // ring slot of tile base / 64
  unsigned r = 0;
#ifdef ZS_PROBE_P2G
  const unsigned long long tp1 = P2G_NOW();
#endif
  {  // the cell counts were requested before the first tiles: they have arrived once at most the tiles' requests are outstanding
    const int req = tIssue - tile0;
    if (req >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB >= 3 ? 3 * NL : 0) : "memory");
    else if (req == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
    else if (req == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned cnt = cntLds[w][lane];
  bool has = cnt > r;
  unsigned long long m = __ballot(has);
  while (m != 0ull) {
    const int nr = __popcll(m);
    const int p = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    const int tb = base >> 6, tLast = (base + nr - 1) >> 6;
    // a tile buffer is free once the walk has passed the tile: tile tIssue - NB was left when base reached (tIssue - NB + 1) * 64
    if (tIssue < tileEnd && tb > tIssue - NB) request();
#ifdef ZS_P2GT_LOOPSYNC
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#ifdef ZS_PROBE_P2G
    const unsigned long long tw0 = P2G_NOW();
#endif
    if (tLast >= tDone) {
      const int ahead = tIssue - 1 - tLast;  // requested tiles the round does not need yet
      if (NB >= 4 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB >= 4 ? 3 * NL : 0) : "memory");
      else if (NB >= 3 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tDone = tLast + 1;
    }
#ifdef ZS_PROBE_P2G
    tpWait += P2G_NOW() - tw0;
    ++tpRounds;
#endif
    {
      // every lane reads "its" record (a lane without a particle in this round reads some record of the ring: never used) so that the
      // round has ONE divergent region, the accumulation; movers are rare and sit behind a wave-uniform branch
      const int nslot = cslot + 1 == NB ? 0 : cslot + 1;
      const float *rec = ring + ((p >> 6) == tb ? cslot : nslot) * TILEF + (p & 63);
      const float pos[3] = {rec[1 * 64], rec[2 * 64], rec[3 * 64]};
      Arena ar;
      make_arena(mp.dx, mp.dxi, pos, ar);
      const int ocx = ar.corner[0] - geo.org[0], ocy = ar.corner[1] - geo.org[1], ocz = ar.corner[2] - geo.org[2];
      const bool inBin = (unsigned)(ocx | ocy | ocz) < 4u;  // all three in 0..3
      const bool own = has && inBin && ((ocx << 4) | (ocy << 2) | ocz) == lane;  // lane = cell: (x, y, z) = (lane >> 4, (lane >> 2) & 3, lane & 3)
#ifdef ZS_P2GT_SCALAR
      if (own) p2gw_accumulate(mp, ar, rec, kscale, acc.a);
#else
      p2gw_accumulate_pk(mp, ar, rec, kscale, own, acc);
#endif
      if (__ballot(has && !own) != 0ull) {  // some particle has left the cell it is stored under (wave-uniform, rare)
        if (has && !own) {
          bool queued = false;
          if (inBin) {  // another cell of the same bin: queued for the post-pass into the arena; outside the bin: exact path afterwards
            const int q = atomicAdd(&mqCount[w], 1);
            if (q < P2GW_MQ_CAP) {
              mq[w][q] = p;
              queued = true;
            }
          }
          if (!queued) stale[atomicAdd(staleCount, 1)] = p;
        }
      }
    }
    base += nr;
    if ((base >> 6) != tb) cslot = cslot + 1 == NB ? 0 : cslot + 1;
    ++r;
    has = cnt > r;
    m = __ballot(has);
  }
#ifdef ZS_PROBE_P2G
  const unsigned long long tp2 = P2G_NOW();
#endif
  // ---- tail.  Every tile the wave requested has been consumed, so its ring is free: it becomes the wave's PRIVATE 6^3 arena, two
  // planes of one float4 per node ({m, mv} and {f, -}: ArenaPriv).  No other wave touches it until the group's barrier below, and a
  // wave's LDS operations execute in order, so the 27 read-add-write phases need no barrier and no wait for a write: the 16-byte read
  // of (phase k + 1, plane p) is issued right behind the write of (phase k, plane p).
  using AP = ArenaPriv;
  static_assert(2 * AP::PLANE * 4 <= WBUF && WBUF % 4 == 0, "private arena inside the wave's ring, 16-byte accesses");
  float4 *priv = reinterpret_cast<float4 *>(ring);
  for (int k = lane; k < 2 * AP::PLANE; k += 64) priv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#define P2GT_LDS_ORDER()                                   \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)
  P2GT_LDS_ORDER();
#ifdef ZS_PROBE_P2G
  const unsigned long long tp3 = P2G_NOW();
#endif
  {
    float4 *a0 = priv + AP::at(lane >> 4, (lane >> 2) & 3, lane & 3);
    float4 va = a0[0], vb = a0[AP::PLANE];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      float4 *g = a0 + AP::at(k / 9, (k / 3) % 3, k % 3);
      float4 *gn = a0 + AP::at((k + 1) / 9, ((k + 1) / 3) % 3, (k + 1) % 3);
      g[0] = make_float4(va.x + acc.get(k, 0), va.y + acc.get(k, 1), va.z + acc.get(k, 2), va.w + acc.get(k, 3));
      P2GT_LDS_ORDER();
      if (k + 1 < 27) va = gn[0];
      g[AP::PLANE] = make_float4(vb.x + acc.get(k, 4), vb.y + acc.get(k, 5), vb.z + acc.get(k, 6), 0.f);
      P2GT_LDS_ORDER();
      if (k + 1 < 27) vb = gn[AP::PLANE];
    }
  }
  {  // the queued in-bin movers, one lane each, by LDS atomics into the private arena (same values as the exact path)
    const int nm = mqCount[w] < P2GW_MQ_CAP ? mqCount[w] : P2GW_MQ_CAP;
    for (int q = lane; q < nm; q += 64) {
      const size_t i = (size_t)mq[w][q];
      float pos[3], vel[3], C[9], PF[9];
      load_attr<3>(ps.pos, i, pos);
      load_attr<3>(ps.vel, i, vel);
      load_attr<9>(ps.C, i, C);
      {
        float S[STRESS_N];
        load_attr<STRESS_N>(ps.stress, i, S);
        stress_unpack(S, PF);
      }
      const float pm = ps.mass.base[ps.mass.off(i)];
#pragma unroll
      for (int d = 0; d < 9; ++d) PF[d] *= kscale;
      Arena ar;
      make_arena(mp.dx, mp.dxi, pos, ar);
      float *b0 = reinterpret_cast<float *>(priv + AP::at(ar.corner[0] - geo.org[0], ar.corner[1] - geo.org[1], ar.corner[2] - geo.org[2]));
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float W = ar.w[0][a] * ar.w[1][b] * ar.w[2][c];
            const float x0 = (float)a * mp.dx - ar.lp[0], x1 = (float)b * mp.dx - ar.lp[1], x2 = (float)c * mp.dx - ar.lp[2];
            float *g = b0 + 4 * AP::at(a, b, c);
            atomicAdd(g, W * pm);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              atomicAdd(g + 1 + d, W * pm * (vel[d] + (C[d] * x0 + C[3 + d] * x1 + C[6 + d] * x2)));
              atomicAdd(g + 4 * AP::PLANE + d, (PF[d] * x0 + PF[3 + d] * x1 + PF[6 + d] * x2) * W);
            }
          }
    }
  }
  if constexpr (G == 1) P2GT_LDS_ORDER();
  else __syncthreads();  // the G private arenas are complete
#undef P2GT_LDS_ORDER
#ifdef ZS_PROBE_P2G
  const unsigned long long tp4 = P2G_NOW();
#endif
  // flush.  The group's nodes (6 x 6 x 6 G: the bins differ in z (G = 2) or in y and z (G = 4)) go to the grid once each: an apron
  // node between two (four) bins is the sum of what their private arenas hold for it.
  {
    const int wy = G == 4 ? (w >> 1) * 4 : 0, wz = G == 1 ? 0 : (w & 1) * 4;
    const int o0[3] = {geo.o[0], geo.o[1] - wy, geo.o[2] - wz};  // origin of the group inside its block = the origin of its first bin
    constexpr int NODES = AL::WX * AL::WY * AL::WZ, ITER = (NODES + 64 * G - 1) / (64 * G);
    float4 va[ITER], vb[ITER];
    int goff[ITER];  // element offset of the node's first channel in the grid, -1: no such block / no such node
#pragma unroll
    for (int it = 0; it < ITER; ++it) {  // every LDS read of the flush first ...
      const int node = (int)threadIdx.x + it * 64 * G;
      va[it] = vb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      goff[it] = -1;
      if (node < NODES) {
        const int x = node / (AL::WY * AL::WZ), y = (node / AL::WZ) % AL::WY, z = node % AL::WZ;
        int slot2, cell;
        arena_to_grid<SIDE>(o0, x, y, z, slot2, cell);
        const int b01 = (slot2 & 1) ? nbs[1] : nbs[0], b23 = (slot2 & 1) ? nbs[3] : nbs[2], b45 = (slot2 & 1) ? nbs[5] : nbs[4],
                  b67 = (slot2 & 1) ? nbs[7] : nbs[6];
        const int b03 = (slot2 & 2) ? b23 : b01, b47 = (slot2 & 2) ? b67 : b45;
        const int bn = (slot2 & 4) ? b47 : b03;
        if (bn >= 0) goff[it] = bn * (7 * NC) + cell;
#pragma unroll
        for (int gy = 0; gy < (G == 4 ? 2 : 1); ++gy)
#pragma unroll
          for (int gz = 0; gz < (G == 1 ? 1 : 2); ++gz) {
            const int ly = y - 4 * gy, lz = z - 4 * gz;
            if ((unsigned)ly < 6u && (unsigned)lz < 6u) {
              const float4 *a = reinterpret_cast<const float4 *>(lds + (gy * 2 + gz) * WBUF) + AP::at(x, ly, lz);
              const float4 pa = a[0], pb = a[AP::PLANE];
              va[it] = make_float4(va[it].x + pa.x, va[it].y + pa.y, va[it].z + pa.z, va[it].w + pa.w);
              vb[it] = make_float4(vb[it].x + pb.x, vb[it].y + pb.y, vb[it].z + pb.z, 0.f);
            }
          }
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it)  // ... then the float atomics
      if (goff[it] >= 0) {
        float *g = grid + (size_t)(unsigned)goff[it];
        const float val[7] = {va[it].x, va[it].y, va[it].z, va[it].w, vb[it].x, vb[it].y, vb[it].z};
#pragma unroll
        for (int ch = 0; ch < 7; ++ch)
          if (val[ch] != 0.f) unsafeAtomicAdd(g + ch * NC, val[ch]);
      }
  }
#ifdef ZS_P2GT_ENDWAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef ZS_PROBE_P2G
  {
    const unsigned long long tp5 = P2G_NOW();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tp6 = P2G_NOW();
    P2G_STAMP(0, tp6 - tp0);
    P2G_STAMP(1, tp1 - tp0);
    P2G_STAMP(2, tp2 - tp1);
    P2G_STAMP(3, tpWait);
    P2G_STAMP(4, tp3 - tp2);
    P2G_STAMP(5, tp4 - tp3);
    P2G_STAMP(6, tp5 - tp4);
    P2G_STAMP(7, tp6 - tp5);
    P2G_STAMP(8, 1);
    P2G_STAMP(9, tpRounds);
  }
#endif
}

// exact path for the queued particles (persistent grid-stride over a device-side count)
template <int SIDE, int MODEL>
static __global__ __launch_bounds__(256) void p2g_stale_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *stale,
                                                        const int *staleCount) {
  const int n = *staleCount;
  const float dxi = mp.dxi;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    p2g_scatter_global<SIDE, MODEL>(mp, ps, (size_t)stale[j], t, grid, 4.f * dxi * dxi);
}

// ======================================================================================= grid update
template <int SIDE>
static __global__ __launch_bounds__(256) void grid_update_kernel(float *grid, size_t nblocks, float dt, float e0, float e1, float e2,
                                                          float *maxVelSqr) {
  constexpr int NC = SIDE * SIDE * SIDE;
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float vsq = 0.f;
  if (g < nblocks * NC) {
    const size_t b = g / NC, c = g % NC;
    float *blk = grid + b * 7 * NC + c;
    float mass = blk[0];
    if (mass != 0.f) {
      mass = 1.f / mass;
      const float v0 = blk[1 * NC] * mass + e0 * dt, v1 = blk[2 * NC] * mass + e1 * dt, v2 = blk[3 * NC] * mass + e2 * dt;
      blk[1 * NC] = v0;
      blk[2 * NC] = v1;
      blk[3 * NC] = v2;
      vsq = v0 * v0 + v1 * v1 + v2 * v2;
    }
  }
  if (maxVelSqr) {  // atomic_max(maxVel, |v|^2) (GridOp.hpp:103-104): workgroup max, then at most one int-ordered atomic
    // One device-wide word takes ~90 atomics per microsecond: an atomic per wave of a 27 200-block grid (217 k of them) cost 2.4 ms.
    // The maximum only grows, so a workgroup first looks at the current value and stays silent unless it can raise it.
    __shared__ float wmax[4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vsq = fmaxf(vsq, shfl_down(vsq, d));
    if (lane_id() == 0) wmax[wave_id()] = vsq;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (m > 0.f && __float_as_int(m) > __hip_atomic_load((int *)maxVelSqr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax((int *)maxVelSqr, __float_as_int(m));
    }
  }
}

// ======================================================================================= G2P
// constitutive update for the NEXT P2G, fused into the tail of G2P where the VALU is otherwise idle (G2P is HBM-bound, P2G
// is VALU-bound by the SVD): stress(F_new, logJp) -> particles.stress (P F^T vol, unscaled), logJp updated.  Exactly what the
// next P2G would compute from the same F (P2G.hpp:60-101); SMODEL < 0: disabled.
template <int SMODEL, int LW = 0>
__device__ __forceinline__ void update_stress(const MpmDev &mp, const ParticlesDev &ps, POff<LW> o, float (&F)[9], const float (&C)[9]) {
  if constexpr (SMODEL >= 0) {
    float PF[9], Fl[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) Fl[d] = F[d];  // the plastic models project their local copy only
    float lj = 0.f;
    if constexpr (model_uses_logjp(SMODEL)) lj = pload1<LW>(ps.logJp, o);
    model_stress<SMODEL>(mp.mat, lj, Fl, PF, C);
    if constexpr (model_uses_logjp(SMODEL)) pstore1<LW>(ps.logJp, o, lj);
    float S[STRESS_N];
    stress_pack(PF, S);
    pstore<LW, STRESS_N>(ps.stress, o, S);
  }
}

template <int SIDE, int SMODEL, int LW = 0>
__device__ __forceinline__ void g2p_finish_loaded(const MpmDev &mp, const ParticlesDev &ps, size_t i, float (&pos)[3], const float (&oldF)[9],
                                                  const float (&vel)[3], const float (&C)[9]) {
  const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
#pragma unroll
  for (int d = 0; d < 3; ++d) pos[d] += vel[d] * mp.dt;
  float F[9];
  advance_state<model_is_fluid(SMODEL)>(oldF, C, mp.dt, F);
  pstore_state<LW, model_is_fluid(SMODEL)>(ps.F, o, F);
  pstore<LW, 3>(ps.pos, o, pos);
  pstore<LW, 3>(ps.vel, o, vel);
  pstore<LW, 9>(ps.C, o, C);
  update_stress<SMODEL, LW>(mp, ps, o, F, C);
}
template <int SIDE, int SMODEL>
__device__ __forceinline__ void g2p_finish(const MpmDev &mp, const ParticlesDev &ps, size_t i, float (&pos)[3], const float (&vel)[3],
                                           const float (&C)[9]) {
  float oldF[9];
  load_state<model_is_fluid(SMODEL)>(ps.F, i, oldF);
  g2p_finish_loaded<SIDE, SMODEL>(mp, ps, i, pos, oldF, vel, C);
}

template <int SIDE, int SMODEL>
__device__ __forceinline__ void g2p_gather_global(const MpmDev &mp, const ParticlesDev &ps, size_t i, const BhtDev &t, const float *grid,
                                                  float D_inv) {
  constexpr int NC = SIDE * SIDE * SIDE;
  float pos[3];
  load_attr<3>(ps.pos, i, pos);
  Arena ar;
  make_arena(mp.dx, mp.dxi, pos, ar);
  int loc[3], key[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    loc[d] = ar.corner[d] & (SIDE - 1);
    key[d] = (ar.corner[d] - loc[d]) / SIDE * mp.kscale;
  }
  int blk[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const bool need = (!(o & 4) || loc[0] + 2 >= SIDE) && (!(o & 2) || loc[1] + 2 >= SIDE) && (!(o & 1) || loc[2] + 2 >= SIDE);
    int k[3] = {key[0] + (o >> 2) * mp.kscale, key[1] + ((o >> 1) & 1) * mp.kscale, key[2] + (o & 1) * mp.kscale};
    blk[o] = need ? bht_query<3>(t, k) : -1;
  }
  float vel[3] = {0.f, 0.f, 0.f}, C[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int x = loc[0] + a, y = loc[1] + b, z = loc[2] + c;
        const int o = ((x >= SIDE) << 2) | ((y >= SIDE) << 1) | (z >= SIDE);
        int bn = blk[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) bn = (o == q) ? blk[q] : bn;
        float vi[3] = {0.f, 0.f, 0.f};
        if (bn >= 0) {
          const float *g = grid + (size_t)bn * 7 * NC + ((x & (SIDE - 1)) * SIDE + (y & (SIDE - 1))) * SIDE + (z & (SIDE - 1));
          vi[0] = g[1 * NC];
          vi[1] = g[2 * NC];
          vi[2] = g[3 * NC];
        }
        const float xi[3] = {(float)a * mp.dx - ar.lp[0], (float)b * mp.dx - ar.lp[1], (float)c * mp.dx - ar.lp[2]};
        float W = ar.w[0][a];
        W *= ar.w[1][b];
        W *= ar.w[2][c];
#pragma unroll
        for (int d = 0; d < 3; ++d) vel[d] += vi[d] * W;
#pragma unroll
        for (int d = 0; d < 9; ++d) C[d] += W * vi[d % 3] * xi[d / 3] * D_inv;
      }
  g2p_finish<SIDE, SMODEL>(mp, ps, i, pos, vel, C);
}

template <int SIDE, int SMODEL>
static __global__ __launch_bounds__(256) void g2p_global_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *grid) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ps.n) return;
  const float dxi = mp.dxi;
  g2p_gather_global<SIDE, SMODEL>(mp, ps, i, t, grid, 4.f * dxi * dxi);
}

// v = sum W v_i and B = sum W v_i (xi - xp)^T over the 27 register-resident node velocities of the lane's cell, by sum
// factorisation over z, then y, then x (W = wx wy wz): ~290 VALU ops instead of ~1000 for the node-by-node form of
// G2P.hpp:54-66 (same sums, different association)
__device__ __forceinline__ void g2p_gather_factorized(const MpmDev &mp, const Arena &ar, const float (&nv)[27][3], float D_inv,
                                                      float (&vel)[3], float (&C)[9]) {
  float xz[3], xy[3], xx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    xx[k] = ar.w[0][k] * node_off(mp.dx, k, ar.lp[0]);
    xy[k] = ar.w[1][k] * node_off(mp.dx, k, ar.lp[1]);
    xz[k] = ar.w[2][k] * node_off(mp.dx, k, ar.lp[2]);
  }
  float B[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // B[j][k]
#pragma unroll
  for (int j = 0; j < 3; ++j) vel[j] = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float t0[3] = {0.f, 0.f, 0.f}, t1[3] = {0.f, 0.f, 0.f}, t2[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      float s0[3], s1[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float v0 = nv[(a * 3 + bb) * 3 + 0][j], v1 = nv[(a * 3 + bb) * 3 + 1][j], v2 = nv[(a * 3 + bb) * 3 + 2][j];
        s0[j] = fmaf(ar.w[2][2], v2, fmaf(ar.w[2][1], v1, ar.w[2][0] * v0));
        s1[j] = fmaf(xz[2], v2, fmaf(xz[1], v1, xz[0] * v0));
        t0[j] = fmaf(ar.w[1][bb], s0[j], t0[j]);
        t1[j] = fmaf(xy[bb], s0[j], t1[j]);
        t2[j] = fmaf(ar.w[1][bb], s1[j], t2[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      vel[j] = fmaf(ar.w[0][a], t0[j], vel[j]);
      B[j][0] = fmaf(xx[a], t0[j], B[j][0]);
      B[j][1] = fmaf(ar.w[0][a], t1[j], B[j][1]);
      B[j][2] = fmaf(ar.w[0][a], t2[j], B[j][2]);
    }
  }
#pragma unroll
  for (int d = 0; d < 9; ++d) C[d] = B[d % 3][d / 3] * D_inv;  // C[d] += W v_i[d%3] xixp[d/3] D_inv (G2P.hpp:65)
}

struct ArenaLds;
template <class AL>
__device__ __forceinline__ void g2p_gather_lds(const MpmDev &mp, const Arena &ar, const float *a0, float D_inv, float (&vel)[3],
                                               float (&C)[9]);

template <int SIDE, int SMODEL, int LW>
static __global__ __launch_bounds__(64) void g2p_binned_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *grid, const int *binStart,
                                                        const unsigned *cellCount, const int *nbr, int *stale, int *staleCount) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float arena[3 * AL::CH];
  const int bin = (int)xcd_chunked(blockIdx.x, gridDim.x);
  const int start = binStart[bin], end = binStart[bin + 1];
  if (start == end) return;
  const int lane = threadIdx.x;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  for (int node = lane; node < 216; node += 64) {  // node decoded once for the 3 velocity channels
    const int x = node / 36, y = (node / 6) % 6, z = node % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = nbr[(size_t)geo.block * 8 + slot];
    float *a = arena + AL::at(x, y, z);
    const float *g = grid + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) a[ch * AL::CH] = bn >= 0 ? g[ch * NC] : 0.f;
  }
  __syncthreads();
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  // the 27 x 3 node velocities of this lane's cell, register resident for all its particles (re-reading them from LDS per
  // particle frees 50 VGPRs but measured 8 % slower)
  float nv[27][3];
  {
    const float *a0 = arena + AL::at(cx, cy, cz);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
      nv[k][0] = g[0];
      nv[k][1] = g[AL::CH];
      nv[k][2] = g[2 * AL::CH];
    }
  }
  const unsigned cnt = cellCount[(size_t)bin * 64 + lane];
  const float dxi = mp.dxi;
  const float D_inv = mp.D_inv;
  RoundWalk walk(cnt, start);
  int i0, i1;
  bool any, any1;
  bool has0 = walk.next(i0, any);
  // x and the deformation state (F, or J for the fluid; the fluid's C is recomputed here, not read)
  RecB<model_is_fluid(SMODEL) ? MPM_FLUID_NO_STRESS : ZS_MPM_FIXED_COROTATED, LW> cur, nxt;
  if (has0) cur.load(ps, (size_t)i0);
  while (any) {
    const bool has1 = walk.next(i1, any1);
    if (has1) nxt.load(ps, (size_t)i1);
    if (has0) {
      Arena ar;
      make_arena(mp.dx, mp.dxi, cur.pos, ar);
      const int ocx = ar.corner[0] - geo.org[0], ocy = ar.corner[1] - geo.org[1], ocz = ar.corner[2] - geo.org[2];
      if (ocx == cx && ocy == cy && ocz == cz) {
        float vel[3], C[9];
        g2p_gather_factorized(mp, ar, nv, D_inv, vel, C);
        g2p_finish_loaded<SIDE, SMODEL, LW>(mp, ps, (size_t)i0, cur.pos, cur.F, vel, C);
      } else if ((unsigned)ocx < 4u && (unsigned)ocy < 4u && (unsigned)ocz < 4u) {
        // another cell of the same bin (the particle moved since the last re-bin): its nodes are in the LDS arena
        float vel[3], C[9];
        g2p_gather_lds<AL>(mp, ar, arena + AL::at(ocx, ocy, ocz), D_inv, vel, C);
        g2p_finish_loaded<SIDE, SMODEL, LW>(mp, ps, (size_t)i0, cur.pos, cur.F, vel, C);
      } else {
        stale[atomicAdd(staleCount, 1)] = i0;  // outside the bin: exact path (hash queries)
      }
    }
    cur = nxt;
    has0 = has1;
    i0 = i1;
    any = any1;
  }
}

// ---- G2P with lane = PARTICLE (r06; the kernel zs_rocm_mpm_g2p launches for binned particles).  g2p_binned_kernel (kept for -DZS_G2P_AB
// comparisons) maps lane = cell so that a lane keeps its cell's 27 x 3 node velocities in
// registers -- and then runs the 860-instruction constitutive update at the lane occupancy of the rounds (~70 %: the fullest cell of a
// bin sets the number of rounds).  Here a workgroup owns a grid block: its particles are ONE contiguous range of the compact order
// (bins of a block are consecutive), the waves take them 64 at a time (every lane busy, loads and stores fully coalesced), and a
// particle gathers from the block's velocity arena in LDS at its own cell (sum-factorised, g2p_gather_lds).  A particle that moved to
// another cell of the block since the last re-bin needs nothing special; one outside the block's cells takes the exact path.
template <int SIDE> struct ArenaOfBlock { using type = ArenaBlk; };
template <> struct ArenaOfBlock<4> { using type = ArenaLds; };
template <int SIDE, int SMODEL, int LW>
static __global__ __launch_bounds__(SIDE == 8 ? 256 : 64) void g2p_packed_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *grid,
                                                                              const int *binStart, const int *nbr, int *stale, int *staleCount) {
  using AL = typename ArenaOfBlock<SIDE>::type;
  constexpr int NC = SIDE * SIDE * SIDE, W = SIDE + 2, NT = SIDE == 8 ? 256 : 64, BPB = bins_per_block<SIDE>();
  __shared__ float arena[3 * AL::CH];
  const int blk = (int)blockIdx.x;
  const int start = binStart[blk * BPB], end = binStart[blk * BPB + BPB];
  if (start == end) return;
  int borg[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) borg[d] = t.activeKeys[3 * (size_t)blk + d] * (SIDE / mp.kscale);
  for (int node = threadIdx.x; node < W * W * W; node += NT) {
    const int x = node / (W * W), y = (node / W) % W, z = node % W;
    const int slot = ((x >= SIDE) << 2) | ((y >= SIDE) << 1) | (z >= SIDE);
    const int cell = ((x & (SIDE - 1)) * SIDE + (y & (SIDE - 1))) * SIDE + (z & (SIDE - 1));
    const int bn = nbr[(size_t)blk * 8 + slot];
    float *a = arena + AL::at(x, y, z);
    const float *g = grid + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) a[ch * AL::CH] = bn >= 0 ? g[ch * NC] : 0.f;
  }
  __syncthreads();
  const float D_inv = mp.D_inv;
  RecB<model_is_fluid(SMODEL) ? MPM_FLUID_NO_STRESS : ZS_MPM_FIXED_COROTATED, LW> cur, nxt;
  int i0 = start + (int)threadIdx.x;
  if (i0 < end) cur.load(ps, (size_t)i0);
  while (i0 - (int)(threadIdx.x & 63) < end) {  // (wave-uniform: the chunk holds at least one particle)
    const int i1 = i0 + NT;
    if (i1 < end) nxt.load(ps, (size_t)i1);   // the wave's next chunk, in flight during this one
    if (i0 < end) {
      Arena ar;
      make_arena(mp.dx, mp.dxi, cur.pos, ar);
      const int ocx = ar.corner[0] - borg[0], ocy = ar.corner[1] - borg[1], ocz = ar.corner[2] - borg[2];
      if ((unsigned)ocx < (unsigned)SIDE && (unsigned)ocy < (unsigned)SIDE && (unsigned)ocz < (unsigned)SIDE) {
        float vel[3], C[9];
        g2p_gather_lds<AL>(mp, ar, arena + AL::at(ocx, ocy, ocz), D_inv, vel, C);
        g2p_finish_loaded<SIDE, SMODEL, LW>(mp, ps, (size_t)i0, cur.pos, cur.F, vel, C);
      } else {
        stale[atomicAdd(staleCount, 1)] = i0;  // the base node lies outside the block's cells: exact path (hash queries)
      }
    }
    cur = nxt;
    i0 = i1;
  }
}

template <int SIDE, int SMODEL>
static __global__ __launch_bounds__(256) void g2p_stale_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *grid, const int *stale,
                                                        const int *staleCount) {
  const int n = *staleCount;
  const float dxi = mp.dxi;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    g2p_gather_global<SIDE, SMODEL>(mp, ps, (size_t)stale[j], t, grid, 4.f * dxi * dxi);
}

// ======================================================================================= G2P2G (fused)
// G2P of step n and P2G of step n+1 in ONE pass over the particles (the reference has the same idea as G2P2GTransfer,
// simulation/transfer/G2P2G.hpp): a particle is read once (m, x, F, logJp: 56 B), gathered from grid A, advected, its F and
// constitutive model updated, and scattered straight into grid B; only x, F, logJp go back to HBM (52 B).  v, C and
// P F^T vol never leave the chip (WRITE_ALL stores them for callers that want the full state).  Unfused, the same work
// moves 296.5 B per particle and step.
//
// One workgroup of four waves owns a bin; lane = cell.  Per chunk of four rounds:
//   phase 1   wave w runs round 4c + w through G2P (node velocities read from the LDS arena) + F update + constitutive model
//             and stages {m, x', v', C', P F^T} of its 64 particles in LDS;
//   phase 2   waves 0/1 accumulate mass + momentum (4 channels, 108 register accumulators) of staged rounds {0,1} / {2,3},
//             waves 2/3 the three stress channels of the same rounds; two LDS arenas collect the two halves.
// The kernel is VALU-bound (SQ_INSTS_VALU x 4 cycles ~ 85 % of the SIMD cycles), so the design minimises instructions: the
// first version kept the 81 node velocities in registers and used the four channel roles of p2g_binned_split_kernel in
// phase 2 (each staged round consumed by 4 waves: 4x the arena / weight work) and took 5.0 ms per 67.1 M-particle step.
// Particles that are not in the cell they are stored under are exact as before: mis-binned at read -> queue G (global
// gather + global scatter afterwards); moved out of the cell by this step's advection -> queue P (state stored, global
// scatter afterwards).
constexpr int G2P2G_NF = 25;
constexpr int G2P2G_MQ_CAP = 512;  // in-bin movers a workgroup can take through its LDS queue (a bin holds ~512 particles)  // staged floats per particle: m, x(3), v(3), C(9), P F^T vol(9)

// gather of g2p_gather_factorized with the node velocities read from the LDS arena (81 ds_read per particle; the fused
// kernel is VALU-bound and needs the 81 VGPRs a register-resident copy would cost for its P2G stencil)
template <class AL>
__device__ __forceinline__ void g2p_gather_lds(const MpmDev &mp, const Arena &ar, const float *a0, float D_inv, float (&vel)[3],
                                               float (&C)[9]) {
  float xz[3], xy[3], xx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    xx[k] = ar.w[0][k] * node_off(mp.dx, k, ar.lp[0]);
    xy[k] = ar.w[1][k] * node_off(mp.dx, k, ar.lp[1]);
    xz[k] = ar.w[2][k] * node_off(mp.dx, k, ar.lp[2]);
  }
  float B[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
  for (int j = 0; j < 3; ++j) vel[j] = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float t0[3] = {0.f, 0.f, 0.f}, t1[3] = {0.f, 0.f, 0.f}, t2[3] = {0.f, 0.f, 0.f};
    // the slab's 27 node values first, then the arithmetic: one LDS latency per slab instead of one per pair of reads
    float nv[3][3][3];
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float *g = a0 + AL::at(a, bb, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        nv[bb][j][0] = g[j * AL::CH];
        nv[bb][j][1] = g[j * AL::CH + 1];
        nv[bb][j][2] = g[j * AL::CH + 2];
      }
    }
#ifndef ZS_GATHER_NO_FENCE
    asm volatile("" ::: "memory");  // (keeps the compiler from sinking the reads back between the fmas)
#endif
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float v0 = nv[bb][j][0], v1 = nv[bb][j][1], v2 = nv[bb][j][2];
        const float s0 = fmaf(ar.w[2][2], v2, fmaf(ar.w[2][1], v1, ar.w[2][0] * v0));
        const float s1 = fmaf(xz[2], v2, fmaf(xz[1], v1, xz[0] * v0));
        t0[j] = fmaf(ar.w[1][bb], s0, t0[j]);
        t1[j] = fmaf(xy[bb], s0, t1[j]);
        t2[j] = fmaf(ar.w[1][bb], s1, t2[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      vel[j] = fmaf(ar.w[0][a], t0[j], vel[j]);
      B[j][0] = fmaf(xx[a], t0[j], B[j][0]);
      B[j][1] = fmaf(ar.w[0][a], t1[j], B[j][1]);
      B[j][2] = fmaf(ar.w[0][a], t2[j], B[j][2]);
    }
  }
#pragma unroll
  for (int d = 0; d < 9; ++d) C[d] = B[d % 3][d / 3] * D_inv;
}

// phase-2 consumer of one staged record.  STRESS = false: mass + momentum (4 channels), true: rhs (3 channels)
template <bool STRESS>
__device__ __forceinline__ void g2p2g_consume(const MpmDev &mp, const float *st, int lane, float kscale, float (&acc)[27][STRESS ? 3 : 4]) {
  auto f = [&](int k) { return st[k * 64 + lane]; };
  const float pos[3] = {f(1), f(2), f(3)};
  Arena ar;
  make_arena(mp.dx, mp.dxi, pos, ar);
  float xo[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int d = 0; d < 3; ++d) xo[d][k] = (float)k * mp.dx - ar.lp[d];
  float Px[3][3], Py[3][3], Pz[3][3], wzs[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float c0 = STRESS ? f(16 + d) : f(7 + d), c1 = STRESS ? f(19 + d) : f(10 + d), c2 = STRESS ? f(22 + d) : f(13 + d);
    const float v = STRESS ? 0.f : f(4 + d);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Px[k][d] = c0 * xo[0][k];
      Py[k][d] = c1 * xo[1][k];
      Pz[k][d] = STRESS ? c2 * xo[2][k] : fmaf(c2, xo[2][k], v);
    }
  }
  const float scale = STRESS ? kscale : f(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) wzs[k] = ar.w[2][k] * scale;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float wxy = ar.w[0][a] * ar.w[1][bb];
      const float q0 = Px[a][0] + Py[bb][0], q1 = Px[a][1] + Py[bb][1], q2 = Px[a][2] + Py[bb][2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float Ws = wxy * wzs[c];
        auto &A = acc[(a * 3 + bb) * 3 + c];
        if constexpr (!STRESS) {
          A[0] += Ws;
          A[1] = fmaf(Ws, q0 + Pz[c][0], A[1]);
          A[2] = fmaf(Ws, q1 + Pz[c][1], A[2]);
          A[3] = fmaf(Ws, q2 + Pz[c][2], A[3]);
        } else {
          A[0] = fmaf(Ws, q0 + Pz[c][0], A[0]);
          A[1] = fmaf(Ws, q1 + Pz[c][1], A[1]);
          A[2] = fmaf(Ws, q2 + Pz[c][2], A[2]);
        }
      }
    }
}

template <int LW, bool DP, bool FLUID = false> struct RecG {  // fused-step inputs: m, x, F or J (, logJp)
  float pos[3], F[9], m, logJp;
  // `delta`: element offset from the (output) attribute arrays of `ps` to the input arrays -- 0 in place; in the re-ordering
  // step the inputs are read from the other buffer of the same layout at the particle's OLD index
  __device__ __forceinline__ void load(const ParticlesDev &ps, size_t i, long long delta = 0) {
    const POff<LW> o = particle_offset<LW>(ps.pos.chns, i);
    Port<float> pp = ps.pos, pf = ps.F, pm = ps.mass, pl = ps.logJp;
    pp.base += delta; pf.base += delta; pm.base += delta;
    // m and logJp FIRST: they are used after the constitutive update, and the wait counter is in order -- as the last loads of the
    // record their wait (s_waitcnt vmcnt(0) behind the SVD) also waited for every store issued in front of the SVD
    m = pload1<LW>(pm, o);
    if constexpr (DP) {
      pl.base += delta;
      logJp = pload1<LW>(pl, o);
    }
    pload<LW, 3>(pp, o, pos);
    pload_state<LW, FLUID>(pf, o, F);
  }
};

// W = wave index: phase 1 handles round 4c + W; phase 2 role: waves 0/1 take mass + momentum of staged rounds {0,1} / {2,3},
// waves 2/3 the stress channels of rounds {0,1} / {2,3}; waves 0,2 accumulate into arena 0, waves 1,3 into arena 1
template <int SIDE, int SMODEL, int LW, bool WRITE_ALL, int W, bool REORDER>
__device__ __forceinline__ void g2p2g_body(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, int start, unsigned cnt,
                                           int lane, const float *varena, float *parena, float *stage, unsigned long long *smask,
                                           int *staleG, int *staleGCount, int *staleP, int *stalePCount, int *mq, int *mqCount,
                                           const int *order, long long inDelta) {
  using AL = ArenaLds;
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool STRESS = W >= 2;
  constexpr int NCH = STRESS ? 3 : 4;
  constexpr int R0 = (W & 1) * 2;  // first of this wave's two staged rounds in phase 2
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = mp.dxi;
  const float D_inv = mp.D_inv;
  const float kscale = mp.fscale;
  const float *v0 = varena + AL::at(cx, cy, cz);
  float acc[27][NCH];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int q = 0; q < NCH; ++q) acc[k][q] = 0.f;
  RoundWalk walk(cnt, start);
  auto next_chunk = [&](int &idx, bool &has, bool &any) {
    any = false;
    has = false;
    idx = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int i;
      bool a;
      const bool h = walk.next(i, a);
      if (r == 0) any = a;
      if (r == W) {
        idx = i;
        has = h;
      }
    }
  };
  int i0, i1;
  bool has0, has1, any, any1;
  next_chunk(i0, has0, any);
  RecG<LW, DP, model_is_fluid(SMODEL)> cur, nxt;
  // re-ordering step (order != nullptr): slot i of the (new) binned order holds the particle stored at order[i] of the input
  // buffer; everything this kernel stores goes to slot i of the output buffer, so the physical re-bin costs no pass of its own
  if (has0) cur.load(ps, REORDER ? (size_t)order[i0] : (size_t)i0, REORDER ? inDelta : 0ll);
  int par = 0;  // stage / mask buffer of this chunk (double buffered: ONE barrier per chunk)
  while (any) {
    float *myStage = stage + (size_t)(par * 4 + W) * (G2P2G_NF * 64);
    next_chunk(i1, has1, any1);
    if (has1) nxt.load(ps, REORDER ? (size_t)order[i1] : (size_t)i1, REORDER ? inDelta : 0ll);  // in flight during this chunk
    // ---------------- phase 1: G2P + update of this wave's round
    bool valid = false;
    if (has0) {
      Arena ar;
      make_arena(mp.dx, mp.dxi, cur.pos, ar);
      // the particle's cell relative to the bin.  Anywhere inside the bin the node velocities are in the LDS arena, so a
      // particle that has wandered into a neighbouring cell of the same bin is still gathered here; only one that is outside
      // the bin altogether takes the exact path (hash queries into grid A)
      const int ocx = ar.corner[0] - geo.org[0], ocy = ar.corner[1] - geo.org[1], ocz = ar.corner[2] - geo.org[2];
      if ((unsigned)ocx >= 4u || (unsigned)ocy >= 4u || (unsigned)ocz >= 4u) {
        if constexpr (REORDER) {  // the exact path works on slot i0 of the output buffer: give it the inputs
          const POff<LW> oo = particle_offset<LW>(ps.pos.chns, (size_t)i0);
          pstore<LW, 3>(ps.pos, oo, cur.pos);
          pstore_state<LW, model_is_fluid(SMODEL)>(ps.F, oo, cur.F);
          pstore1<LW>(ps.mass, oo, cur.m);
          if constexpr (DP) pstore1<LW>(ps.logJp, oo, cur.logJp);
        }
        staleG[atomicAdd(staleGCount, 1)] = i0;  // outside the bin: exact gather + scatter afterwards
        // drift guard of the split launch: the exact path of an interior block may only reach blocks within two of its own
        if ((unsigned)(ocx + 4) >= 12u || (unsigned)(ocy + 4) >= 12u || (unsigned)(ocz + 4) >= 12u) staleGCount[8] = 1;
      } else {
        float vel[3], C[9];
        g2p_gather_lds<AL>(mp, ar, varena + AL::at(ocx, ocy, ocz), D_inv, vel, C);
        const POff<LW> o = particle_offset<LW>(ps.pos.chns, (size_t)i0);
        float pos[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) pos[d] = cur.pos[d] + vel[d] * mp.dt;
        float F[9], PF[9];
        advance_state<model_is_fluid(SMODEL)>(cur.F, C, mp.dt, F);
        pstore_state<LW, model_is_fluid(SMODEL)>(ps.F, o, F);
        pstore<LW, 3>(ps.pos, o, pos);
        if constexpr (REORDER) pstore1<LW>(ps.mass, o, cur.m);  // the mass moves with the particle
        {  // F has been stored above: the plastic models may project this local copy
          float lj = 0.f;
          if constexpr (DP) lj = cur.logJp;
          model_stress<SMODEL>(mp.mat, lj, F, PF, C);
          if constexpr (DP) pstore1<LW>(ps.logJp, o, lj);
        }
        // where is it now?  same cell as this lane: register accumulation (phase 2).  Another cell of the same bin: queued in
        // LDS and scattered into the bin's arena by the dense post-pass of the kernel.  Outside the bin: exact path.
        const int ncx = (int)floorf(pos[0] * dxi - 0.5f) - geo.org[0], ncy = (int)floorf(pos[1] * dxi - 0.5f) - geo.org[1],
                  ncz = (int)floorf(pos[2] * dxi - 0.5f) - geo.org[2];
        const bool moved = ncx != cx || ncy != cy || ncz != cz;
        if (WRITE_ALL || moved) {
          pstore<LW, 3>(ps.vel, o, vel);
          pstore<LW, 9>(ps.C, o, C);
          {
            float S[STRESS_N];
            stress_pack(PF, S);
            pstore<LW, STRESS_N>(ps.stress, o, S);
          }
        }
        if (moved) {
          bool queued = false;
          if ((unsigned)ncx < 4u && (unsigned)ncy < 4u && (unsigned)ncz < 4u) {
            const int slot = atomicAdd(mqCount, 1);
            if (slot < G2P2G_MQ_CAP) {
              mq[slot] = i0;
              queued = true;
            }
          }
          if (!queued) {
            staleP[atomicAdd(stalePCount, 1)] = i0;  // left the bin during this step: exact scatter afterwards
            if ((unsigned)(ncx + 4) >= 12u || (unsigned)(ncy + 4) >= 12u || (unsigned)(ncz + 4) >= 12u) staleGCount[8] = 1;
          }
        } else {
          valid = true;
          myStage[0 * 64 + lane] = cur.m;
#pragma unroll
          for (int d = 0; d < 3; ++d) myStage[(1 + d) * 64 + lane] = pos[d];
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
      if (lane == 0) smask[par * 4 + W] = vm;
    }
    __syncthreads();  // this chunk is staged; everybody has finished consuming the chunk before the previous one
    // ---------------- phase 2: two staged rounds per wave, 4 (mass + momentum) or 3 (stress) channels
#pragma unroll 1
    for (int rr = R0; rr < R0 + 2; ++rr) {
      const unsigned long long vm = smask[par * 4 + rr];
      if (vm == 0ull) continue;
      if ((vm >> lane) & 1ull) g2p2g_consume<STRESS>(mp, stage + (size_t)(par * 4 + rr) * (G2P2G_NF * 64), lane, kscale, acc);
    }
    par ^= 1;
    cur = nxt;
    has0 = has1;
    i0 = i1;
    any = any1;
  }
  float *a0 = parena + (size_t)(W & 1) * (7 * AL::CH) + AL::at(cx, cy, cz);
  // Every wave owns its (arena, channel set): waves 0/2 write arena 0 (channels 0-3 / 4-6), waves 1/3 arena 1.  In phase k the 64
  // lanes of a wave add to 64 distinct nodes; the next phase touches nodes other lanes wrote in this one, so the phases must stay
  // ordered -- but only inside the wave: LDS operations of one wave execute in order, so a wavefront-scope fence (no instruction,
  // it only keeps the compiler from hoisting the next phase's reads over this phase's writes) replaces the 27 workgroup barriers.
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    float *g = a0 + AL::at(k / 9, (k / 3) % 3, k % 3);
#pragma unroll
    for (int q = 0; q < NCH; ++q) g[((STRESS ? 4 : 0) + q) * AL::CH] += acc[k][q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  __syncthreads();  // the post-pass and the flush read both arenas
}

template <int SIDE, int SMODEL, int LW, bool WRITE_ALL, bool REORDER = false>
static __global__ __launch_bounds__(256) void g2p2g_binned_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *gridA, float *gridB,
                                                           const int *binStart, const unsigned *cellCount, const int *nbr, int *staleG,
                                                           int *staleGCount, int *staleP, int *stalePCount, int binBase,
                                                           const int *order, long long inDelta) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float varena[3 * AL::CH];
  __shared__ float parena[2 * 7 * AL::CH];
  __shared__ float stage[2 * 4 * G2P2G_NF * 64];
  __shared__ unsigned long long smask[2 * 4];
  __shared__ int mq[G2P2G_MQ_CAP];
  __shared__ int mqCount;
  if (threadIdx.x == 0) mqCount = 0;
  const int bin = (int)xcd_chunked(blockIdx.x, gridDim.x) + binBase;  // a launch covers a range of blocks (boundary blocks first, see zs_rocm_mpm_g2p2g_range)
  const int start = binStart[bin], end = binStart[bin + 1];
  if (start == end) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  if (tid < 216) {  // node decoded once for the 3 velocity channels
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = nbr[(size_t)geo.block * 8 + slot];
    float *a = varena + AL::at(x, y, z);
    const float *g = gridA + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) a[ch * AL::CH] = bn >= 0 ? g[ch * NC] : 0.f;
  }
  for (int k = tid; k < 2 * 7 * AL::CH; k += 256) parena[k] = 0.f;
  const unsigned cnt = cellCount[(size_t)bin * 64 + lane];
  __syncthreads();
  if (w == 0) g2p2g_body<SIDE, SMODEL, LW, WRITE_ALL, 0, REORDER>(mp, ps, geo, start, cnt, lane, varena, parena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, order, inDelta);
  else if (w == 1) g2p2g_body<SIDE, SMODEL, LW, WRITE_ALL, 1, REORDER>(mp, ps, geo, start, cnt, lane, varena, parena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, order, inDelta);
  else if (w == 2) g2p2g_body<SIDE, SMODEL, LW, WRITE_ALL, 2, REORDER>(mp, ps, geo, start, cnt, lane, varena, parena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, order, inDelta);
  else g2p2g_body<SIDE, SMODEL, LW, WRITE_ALL, 3, REORDER>(mp, ps, geo, start, cnt, lane, varena, parena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, order, inDelta);
  // dense post-pass over the particles that changed cell inside this bin: one thread per particle, contributions added to the
  // bin's arena with LDS atomics (the register stencils of the lanes are keyed to cells).  Their state was stored by other
  // lanes of this workgroup a moment ago: read it at agent scope so that a stale L1 line (x was loaded in phase 1) cannot serve it.
  // (Measured alternative: records parked in LDS and walked one by one with lane = node and plain read-add-write per wave-owned
  // channel -- no atomics, but a serial, latency-bound walk: 20 % slower on the 200-step free fall.)
  {
    // The queued particles' state was stored by OTHER waves of this workgroup during the loop, with plain stores; the reads below are
    // agent-scope loads.  A barrier orders instructions, not the arrival of stores at L2 (outside threadgroup-split mode a workgroup-scope
    // release does not wait for vmcnt), so every wave drains its stores and the workgroup meets once more before the post-pass reads.
    // (Added while hunting the rare deviation of the 24-step test; that turned out to be something else -- profiles/r03_compact_outliers.md --
    // but the ordering is not guaranteed without it.)
    if (mqCount > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int nm = mqCount < G2P2G_MQ_CAP ? mqCount : G2P2G_MQ_CAP;  // the body ended with a barrier
    const float dxi = mp.dxi;
    const float kscale = mp.fscale;
    for (int q = tid; q < nm; q += 256) {
      const size_t i = (size_t)mq[q];
      auto cload = [&](const Port<float> &p, int comp) {
        return __hip_atomic_load(p.base + p.off(i) + (size_t)comp * p.cstride(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      const float m = ps.mass.base[ps.mass.off(i)];
      float pos[3], vel[3], C[9], PF[9];
#pragma unroll
      for (int d = 0; d < 3; ++d) { pos[d] = cload(ps.pos, d); vel[d] = cload(ps.vel, d); }
#pragma unroll
      for (int d = 0; d < 9; ++d) C[d] = cload(ps.C, d);
      {
        float S[STRESS_N];
#pragma unroll
        for (int d = 0; d < STRESS_N; ++d) S[d] = cload(ps.stress, d) * kscale;
        stress_unpack(S, PF);
      }
      Arena ar;
      make_arena(mp.dx, mp.dxi, pos, ar);
      const int kx = ar.corner[0] - geo.org[0], ky = ar.corner[1] - geo.org[1], kz = ar.corner[2] - geo.org[2];
      if ((unsigned)kx >= 4u || (unsigned)ky >= 4u || (unsigned)kz >= 4u) {
        // the queueing test rounds pos * (1/dx) - 0.5 in one step, make_arena in two: on an exact cell face they can disagree
        staleP[atomicAdd(stalePCount, 1)] = (int)i;
        continue;
      }
      float *a0 = parena + AL::at(kx, ky, kz);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float W = ar.w[0][a] * ar.w[1][b] * ar.w[2][c];
            const float x0 = (float)a * mp.dx - ar.lp[0], x1 = (float)b * mp.dx - ar.lp[1], x2 = (float)c * mp.dx - ar.lp[2];
            float *g = a0 + AL::at(a, b, c);
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
  if (tid < 216) {
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = nbr[(size_t)geo.block * 8 + slot];
    const float *a = parena + AL::at(x, y, z);
    if (bn >= 0) {
      float *g = gridB + (size_t)bn * 7 * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        const float v = a[ch * AL::CH] + a[(7 + ch) * AL::CH];
        if (v != 0.f) unsafeAtomicAdd(g + ch * NC, v);
      }
    }
    else if (a[0] + a[7 * AL::CH] != 0.f) {
      staleGCount[9] = 1;  // mass for a node whose block is not in the partition: the partition no longer covers the particles
    }
  }
}
// ---------------------------------------------------------------------------------------------------------------------------
// Role-split variant of the fused pass (the default; g2p2g_binned_kernel above stays for the re-ordering step and for A/B runs
// with ZS_ROCM_G2P2G_CLASSIC=1).  Measured on the four-wave kernel (tools/ablate.sh, 64 Mi particles): the costs of its parts
// ADD UP instead of overlapping -- constitutive update 1.0 ms + phase-2 accumulation 0.9 + gather 0.5 + streaming skeleton 2.2
// + head/tail of a bin 0.45 = 5.0 ms -- because at 223 VGPRs / 74.5 KB LDS only two waves share a SIMD, each of them parked 37 %
// of its life (SQ_WAIT_ANY), and one wave alone issues a VALU instruction only every ~5 cycles.  The accumulators (27 nodes x 7
// channels per cell) are what costs the registers, so they move to waves of their own:
//   waves 0-3  PRODUCERS  round 4c + w of chunk c: G2P from the LDS velocity arena, advection, F update, constitutive model,
//                         stores, {m, x', v', C', P F^T} staged in LDS -- no accumulators: < 128 VGPRs
//   waves 4-7  CONSUMERS  of the chunk staged one iteration earlier: each owns a channel set {m, mv_x} {mv_y, mv_z} {f_x, f_y}
//                         {f_z} of ALL four staged rounds: 54 accumulators, < 128 VGPRs; each channel of the bin's single LDS
//                         arena belongs to one wave, so the final flush needs no barrier between its 27 phases
// One barrier per chunk (stage double-buffered), 512 threads, 66 KB LDS: two workgroups = 16 waves per CU = 4 per SIMD.  The
// per-record arena / weight set-up is repeated by four consumers instead of two (+190 VALU per 64 particles, +9 %).
#ifdef ZS_PROBE  // measurement-only build (tools/ablate.sh PROBE): s_memtime stamps of a workgroup's phases, summed over all workgroups
__device__ unsigned long long g_probe[16];
#define ZS_STAMP(slot, t0) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x & 127) == 0) atomicAdd(&g_probe[slot], (unsigned long long)(__builtin_readcyclecounter() - (t0))); } while (0)
#else
#define ZS_STAMP(slot, t0) do { } while (0)
#endif
template <int CS> struct ConsumerSet {  // CS 0: m + mv_x, 1: mv_y + mv_z, 2: f_x + f_y, 3: f_z
  static constexpr bool STRESS = CS >= 2;
  static constexpr bool MASS = CS == 0;
  static constexpr int NV = CS == 1 || CS == 2 ? 2 : 1;        // vector-valued channels (a direction d each)
  static constexpr int NA = NV + (MASS ? 1 : 0);               // accumulators per node
  static constexpr int D0 = CS == 0 ? 0 : (CS == 1 ? 1 : (CS == 2 ? 0 : 2));  // first direction; the second is D0 + 1
  static constexpr int CH0 = CS == 0 ? 0 : (CS == 1 ? 2 : (CS == 2 ? 4 : 6));  // first grid channel of the set
};
// Staged record of the role-split kernels (r05, "Q form"): what a particle adds to node (a, b, c) of its stencil in vector channel j is
//   W_abc (alpha_j + (a - 1) bx_j + (b - 1) by_j + (c - 1) bz_j),
// alpha = the channel's value at the CENTRE node of the stencil, b. = its change per node step -- momentum d (P2G.hpp:112-119):
// alpha = m (v_d + C[d, :] . (dx - lp)), b_k = m C[d + 3 k] dx; force d (:104-110): alpha = kscale (P F^T)[d, :] . (dx - lp),
// b_k = kscale (P F^T)[d + 3 k] dx.  The producer (lane = particle, every lane busy) forms the 24 coefficients once; the four
// consumers (lane = cell, a third of the lanes idle, everything repeated per channel set) no longer rebuild the offsets x_i - x_p and
// the products C . (x_i - x_p) per node: 766 -> 585 VALU instructions per consumed round.
//   [0] m, [1..3] d0 = x'/dx - base node (local position in cells, [0.5, 1.5)), [4 + 4 j + {0, 1, 2, 3}] = alpha, bx, by, bz of
//   channel j = mv_x, mv_y, mv_z, f_x, f_y, f_z
constexpr int G2P2G_QF = 28;
__device__ __forceinline__ void stage_qform(const MpmDev &mp, float *st, float pm, const float (&lpn)[3], const float (&vel)[3], const float (&C)[9],
                                            const float (&PF)[9]) {
  const float dxi = mp.dxi;
  const float kscale = mp.fscale;
  float lc[3];  // centre node - particle
#pragma unroll
  for (int k = 0; k < 3; ++k) lc[k] = fmaf(-lpn[k], mp.dx, mp.dx);
  st[0] = pm;
#pragma unroll
  for (int d = 0; d < 3; ++d) st[(1 + d) * 64] = lpn[d];
  const float pmdx = pm * mp.dx, ksdx = mp.fscaleDx;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float *q = st + (4 + 4 * d) * 64;
    q[0] = pm * (vel[d] + (C[d] * lc[0] + C[3 + d] * lc[1] + C[6 + d] * lc[2]));
    q[64] = pmdx * C[d];
    q[128] = pmdx * C[3 + d];
    q[192] = pmdx * C[6 + d];
    float *g = st + (16 + 4 * d) * 64;
    g[0] = kscale * (PF[d] * lc[0] + PF[3 + d] * lc[1] + PF[6 + d] * lc[2]);
    g[64] = ksdx * PF[d];
    g[128] = ksdx * PF[3 + d];
    g[192] = ksdx * PF[6 + d];
  }
}
template <int CS>
__device__ __forceinline__ void g2p2g_consume_set(const MpmDev &mp, const float *st, int lane, float (&acc)[27][ConsumerSet<CS>::NA]) {
  using S = ConsumerSet<CS>;
  auto f = [&](int k) { return st[k * 64 + lane]; };
  // every staged value of the set first (LDS reads in one go), then the arithmetic: one LDS latency per particle
  float d0s[3], al[S::NV], bx[S::NV], by[S::NV], bz[S::NV];
#pragma unroll
  for (int d = 0; d < 3; ++d) d0s[d] = f(1 + d);
  float pm = 0.f;
  if constexpr (S::MASS) pm = f(0);
#pragma unroll
  for (int j = 0; j < S::NV; ++j) {
    const int q = 4 + 4 * ((S::STRESS ? 3 : 0) + S::D0 + j);
    al[j] = f(q);
    bx[j] = f(q + 1);
    by[j] = f(q + 2);
    bz[j] = f(q + 3);
  }
  asm volatile("" ::: "memory");  // (keeps the compiler from sinking the reads back between the fmas)
  float w[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float d0 = d0s[d];
    w[d][0] = 0.5f * (1.5f - d0) * (1.5f - d0);
    const float d1 = d0 - 1.0f;
    w[d][1] = 0.75f - d1 * d1;
    const float zz = 0.5f + d1;
    w[d][2] = 0.5f * zz * zz;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float qa[S::NV];
#pragma unroll
    for (int j = 0; j < S::NV; ++j) qa[j] = a == 0 ? al[j] - bx[j] : (a == 1 ? al[j] : al[j] + bx[j]);
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float wxy = w[0][a] * w[1][bb];
      const float W0 = wxy * w[2][0], W1 = wxy * w[2][1], W2 = wxy * w[2][2];
      auto &A0 = acc[(a * 3 + bb) * 3], &A1 = acc[(a * 3 + bb) * 3 + 1], &A2 = acc[(a * 3 + bb) * 3 + 2];
      if constexpr (S::MASS) {
        A0[0] = fmaf(W0, pm, A0[0]);
        A1[0] = fmaf(W1, pm, A1[0]);
        A2[0] = fmaf(W2, pm, A2[0]);
      }
#pragma unroll
      for (int j = 0; j < S::NV; ++j) {
        const float qab = bb == 0 ? qa[j] - by[j] : (bb == 1 ? qa[j] : qa[j] + by[j]);
        constexpr int o = S::MASS ? 1 : 0;
        A0[o + j] = fmaf(W0, qab - bz[j], A0[o + j]);
        A1[o + j] = fmaf(W1, qab, A1[o + j]);
        A2[o + j] = fmaf(W2, qab + bz[j], A2[o + j]);
      }
    }
  }
}
template <int CS>
__device__ __forceinline__ void g2p2g_rs_consumer(const MpmDev &mp, int lane, int nchunks, const float *stage, const unsigned long long *smask,
                                                  float *parena) {
  using S = ConsumerSet<CS>;
  using AL = ArenaLds;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = mp.dxi;
  const float kscale = mp.fscale;
  float acc[27][S::NA];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int q = 0; q < S::NA; ++q) acc[k][q] = 0.f;
  for (int k = (int)threadIdx.x - 256; k < 7 * AL::CH; k += 256) parena[k] = 0.f;  // the four consumer waves clear the bin's arena
  __syncthreads();  // (the producers fill the velocity arena meanwhile)
  for (int it = 0; it <= nchunks; ++it) {
    if (it > 0) {
      const int par = (it - 1) & 1;
#pragma unroll 1
      for (int rr = 0; rr < 4; ++rr) {
        const unsigned long long vm = smask[par * 4 + rr];
        if (vm == 0ull) continue;
        if ((vm >> lane) & 1ull) g2p2g_consume_set<CS>(mp, stage + (size_t)(par * 4 + rr) * (G2P2G_QF * 64), lane, acc);
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
// producer wave W (0..3): round 4c + W of every chunk c
template <int SIDE, int SMODEL, int LW, bool WRITE_ALL, int W>
__device__ __forceinline__ void g2p2g_rs_producer(const MpmDev &mp, const ParticlesDev &ps, const BinGeom<SIDE> &geo, int start, unsigned cnt,
                                                  int lane, int nchunks, float *varena, float *stage, unsigned long long *smask,
                                                  int *staleG, int *staleGCount, int *staleP, int *stalePCount, int *mq, int *mqCount,
                                                  const float *gridA, const int *nbr) {
  using AL = ArenaLds;
  constexpr bool DP = model_uses_logjp(SMODEL);
  constexpr bool FLUID = model_is_fluid(SMODEL);
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const float dxi = mp.dxi;
  const float D_inv = mp.D_inv;
  RoundWalk walk(cnt, start);
  auto next_chunk = [&](int &idx, bool &has) {
    has = false;
    idx = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int i;
      bool a;
      const bool h = walk.next(i, a);
      if (r == W) {
        idx = i;
        has = h;
      }
    }
  };
  int i0 = 0, i1 = 0;
  bool has0 = false, has1 = false;
  RecG<LW, DP, FLUID> cur, nxt;
  // head of the bin: the first records are requested BEFORE the velocity arena is filled -- both need only what the bin number
  // gives (binStart / cellCount / block key / nbr row arrive together), so a bin starts after two memory round trips, not four
  // (requested into `nxt` and handed over at the top of the iteration that uses it: see g2p2g_slot_producer)
  if (nchunks > 0) {
    next_chunk(i1, has1);
    if (has1) nxt.load(ps, (size_t)i1);
  }
  {
    constexpr int NC = SIDE * SIDE * SIDE;
    const int tid = (int)threadIdx.x;  // the four producer waves are threads 0..255
    if (tid < 216) {  // node decoded once for the 3 velocity channels
      const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
      int slot, cell;
      arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
      const int bn = nbr[(size_t)geo.block * 8 + slot];
      float *a = varena + AL::at(x, y, z);
      const float *g = gridA + ((size_t)(bn < 0 ? 0 : bn) * 7 + 1) * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) a[ch * AL::CH] = bn >= 0 ? g[ch * NC] : 0.f;
    }
  }
#ifdef ZS_PROBE
  const unsigned long long tP0 = __builtin_readcyclecounter();
#endif
  __syncthreads();
  ZS_STAMP(4, tP0);  // [4] producers: wait at the barrier behind the arena fill (grid loads landing)
#ifdef ZS_PROBE
  unsigned long long tBar = 0, tP1 = __builtin_readcyclecounter();
#endif
  for (int it = 0; it <= nchunks; ++it) {
    if (it < nchunks) {
      const int par = it & 1;
      float *myStage = stage + (size_t)(par * 4 + W) * (G2P2G_QF * 64);
      cur = nxt;
      has0 = has1;
      i0 = i1;
      has1 = false;
      if (it + 1 < nchunks) {
        next_chunk(i1, has1);
        if (has1) nxt.load(ps, (size_t)i1);  // in flight during this chunk
      }
      bool valid = false;
      if (has0) {
        Arena ar;
        make_arena(mp.dx, mp.dxi, cur.pos, ar);
        const int ocx = ar.corner[0] - geo.org[0], ocy = ar.corner[1] - geo.org[1], ocz = ar.corner[2] - geo.org[2];
        if ((unsigned)ocx >= 4u || (unsigned)ocy >= 4u || (unsigned)ocz >= 4u) {
          staleG[atomicAdd(staleGCount, 1)] = i0;  // outside the bin: exact gather + scatter afterwards
          if ((unsigned)(ocx + 4) >= 12u || (unsigned)(ocy + 4) >= 12u || (unsigned)(ocz + 4) >= 12u) staleGCount[8] = 1;
        } else {
          float vel[3], C[9];
          g2p_gather_lds<AL>(mp, ar, varena + AL::at(ocx, ocy, ocz), D_inv, vel, C);
          const POff<LW> o = particle_offset<LW>(ps.pos.chns, (size_t)i0);
          float pos[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) pos[d] = cur.pos[d] + vel[d] * mp.dt;
          float F[9], PF[9];
          advance_state<FLUID>(cur.F, C, mp.dt, F);
          pstore_state<LW, FLUID>(ps.F, o, F);
          pstore<LW, 3>(ps.pos, o, pos);
          {  // F has been stored above: the plastic models may project this local copy
            float lj = 0.f;
            if constexpr (DP) lj = cur.logJp;
            model_stress<SMODEL>(mp.mat, lj, F, PF, C);
            if constexpr (DP) pstore1<LW>(ps.logJp, o, lj);
          }
          // base node and normalised local position of the NEW position, exactly as make_arena derives them
          float lpn[3];
          int nc[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const float X = pos[d] * dxi;
            const float fl = floorf(X - 0.5f);
            nc[d] = (int)fl - geo.org[d];
            lpn[d] = X - fl;
          }
          const int ncx = nc[0], ncy = nc[1], ncz = nc[2];
          // The reference derives the weights from localPos - base_node(localPos) (InterpolationKernel.hpp:108) although localPos is already
          // relative to the base node (simulation/Utils.hpp:59-60).  The second base_node is 0 -- except when X - floor(X - 0.5) ROUNDS up to
          // 1.5, or X - 0.5 rounds up to an integer and leaves it just below 0.5 (only possible for |X| < 1, next to the coordinate origin):
          // then it is +-1 and the weights are those of d0 -+ 1 on the unchanged corner.  make_arena restates that; the consumers take the
          // staged lpn as d0 without the second floor, so such a particle goes the way of the in-bin movers (post-pass, make_arena) instead.
          // profiles/r03_compact_outliers.md
          const bool moved = ncx != cx || ncy != cy || ncz != cz ||
                             !(lpn[0] >= 0.5f && lpn[0] < 1.5f && lpn[1] >= 0.5f && lpn[1] < 1.5f && lpn[2] >= 0.5f && lpn[2] < 1.5f);
          if (WRITE_ALL || moved) {
            pstore<LW, 3>(ps.vel, o, vel);
            pstore<LW, 9>(ps.C, o, C);
            {
            float S[STRESS_N];
            stress_pack(PF, S);
            pstore<LW, STRESS_N>(ps.stress, o, S);
          }
          }
          if (moved) {
            bool queued = false;
            if ((unsigned)ncx < 4u && (unsigned)ncy < 4u && (unsigned)ncz < 4u) {
              const int slot = atomicAdd(mqCount, 1);
              if (slot < G2P2G_MQ_CAP) {
                mq[slot] = i0;
                queued = true;
              }
            }
            if (!queued) {
              staleP[atomicAdd(stalePCount, 1)] = i0;  // left the bin during this step: exact scatter afterwards
              if ((unsigned)(ncx + 4) >= 12u || (unsigned)(ncy + 4) >= 12u || (unsigned)(ncz + 4) >= 12u) staleGCount[8] = 1;
            }
          } else {
            valid = true;
            stage_qform(mp, myStage + lane, cur.m, lpn, vel, C, PF);
          }
        }
      }
      {
        const unsigned long long vm = __ballot(valid);
        if (lane == 0) smask[par * 4 + W] = vm;
      }
    }
#ifdef ZS_PROBE
    const unsigned long long tb = __builtin_readcyclecounter();
    __syncthreads();
    tBar += __builtin_readcyclecounter() - tb;
#else
    __syncthreads();
#endif
  }
#ifdef ZS_PROBE
  if (lane == 0 && (blockIdx.x & 127) == 0) {
    atomicAdd(&g_probe[5], tBar);                                    // [5] producers: time inside the per-chunk barriers
    atomicAdd(&g_probe[6], __builtin_readcyclecounter() - tP1);      // [6] producers: the chunk loop
  }
#endif
}

template <int SIDE, int SMODEL, int LW, bool WRITE_ALL>
static __global__ __launch_bounds__(512, 4) void g2p2g_rs_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *gridA, float *gridB,
                                                          const int *binStart, const unsigned *cellCount, const int *nbr, int *staleG,
                                                          int *staleGCount, int *staleP, int *stalePCount, int binBase) {
  using AL = ArenaLds;
  constexpr int NC = SIDE * SIDE * SIDE;
  __shared__ float varena[3 * AL::CH];
  __shared__ float parena[7 * AL::CH];
  __shared__ float stage[2 * 4 * G2P2G_QF * 64];
  __shared__ unsigned long long smask[2 * 4];
  __shared__ int mq[G2P2G_MQ_CAP];
  __shared__ int mqCount;
  if (threadIdx.x == 0) mqCount = 0;
#ifdef ZS_PROBE
  const unsigned long long tEntry = __builtin_readcyclecounter();
#endif
  const int bin = (int)xcd_chunked(blockIdx.x, gridDim.x) + binBase;
  const int start = binStart[bin], end = binStart[bin + 1];
  if (start == end) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const BinGeom<SIDE> geo(t, bin, mp.kscale);
  const unsigned cnt = cellCount[(size_t)bin * 64 + lane];
  // rounds of this bin = the fullest cell; every wave needs the number of chunks (uniform loop with one barrier per chunk)
  unsigned mx = cnt;
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)mx, sft, 64);
    mx = o > mx ? o : mx;
  }
  const int nchunks = (int)((mx + 3u) >> 2);
  if (w == 0) g2p2g_rs_producer<SIDE, SMODEL, LW, WRITE_ALL, 0>(mp, ps, geo, start, cnt, lane, nchunks, varena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, gridA, nbr);
  else if (w == 1) g2p2g_rs_producer<SIDE, SMODEL, LW, WRITE_ALL, 1>(mp, ps, geo, start, cnt, lane, nchunks, varena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, gridA, nbr);
  else if (w == 2) g2p2g_rs_producer<SIDE, SMODEL, LW, WRITE_ALL, 2>(mp, ps, geo, start, cnt, lane, nchunks, varena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, gridA, nbr);
  else if (w == 3) g2p2g_rs_producer<SIDE, SMODEL, LW, WRITE_ALL, 3>(mp, ps, geo, start, cnt, lane, nchunks, varena, stage, smask, staleG, staleGCount, staleP, stalePCount, mq, &mqCount, gridA, nbr);
  else if (w == 4) g2p2g_rs_consumer<0>(mp, lane, nchunks, stage, smask, parena);
  else if (w == 5) g2p2g_rs_consumer<1>(mp, lane, nchunks, stage, smask, parena);
  else if (w == 6) g2p2g_rs_consumer<2>(mp, lane, nchunks, stage, smask, parena);
  else g2p2g_rs_consumer<3>(mp, lane, nchunks, stage, smask, parena);
  ZS_STAMP(w < 4 ? 0 : 1, tEntry);  // [0] producers / [1] consumers: entry -> end of the role body (summed over 4 waves each)
  __syncthreads();  // all channel sets are in the arena
  ZS_STAMP(2, tEntry);              // [2] entry -> past the barrier behind the bodies (8 waves)
  // in-bin movers: dense post-pass with LDS atomics (see g2p2g_binned_kernel)
  {
    // The queued particles' state was stored by OTHER waves of this workgroup during the loop, with plain stores; the reads below are
    // agent-scope loads.  A barrier orders instructions, not the arrival of stores at L2 (outside threadgroup-split mode a workgroup-scope
    // release does not wait for vmcnt), so every wave drains its stores and the workgroup meets once more before the post-pass reads.
    // (Added while hunting the rare deviation of the 24-step test; that turned out to be something else -- profiles/r03_compact_outliers.md --
    // but the ordering is not guaranteed without it.)
    if (mqCount > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int nm = mqCount < G2P2G_MQ_CAP ? mqCount : G2P2G_MQ_CAP;
    const float dxi = mp.dxi;
    const float kscale = mp.fscale;
    for (int q = tid; q < nm; q += 512) {
      const size_t i = (size_t)mq[q];
      auto cload = [&](const Port<float> &p, int comp) {
        return __hip_atomic_load(p.base + p.off(i) + (size_t)comp * p.cstride(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      const float m = ps.mass.base[ps.mass.off(i)];
      float pos[3], vel[3], C[9], PF[9];
#pragma unroll
      for (int d = 0; d < 3; ++d) { pos[d] = cload(ps.pos, d); vel[d] = cload(ps.vel, d); }
#pragma unroll
      for (int d = 0; d < 9; ++d) C[d] = cload(ps.C, d);
      {
        float S[STRESS_N];
#pragma unroll
        for (int d = 0; d < STRESS_N; ++d) S[d] = cload(ps.stress, d) * kscale;
        stress_unpack(S, PF);
      }
      Arena ar;
      make_arena(mp.dx, mp.dxi, pos, ar);
      const int kx = ar.corner[0] - geo.org[0], ky = ar.corner[1] - geo.org[1], kz = ar.corner[2] - geo.org[2];
      if ((unsigned)kx >= 4u || (unsigned)ky >= 4u || (unsigned)kz >= 4u) {
        staleP[atomicAdd(stalePCount, 1)] = (int)i;
        continue;
      }
      float *a0 = parena + AL::at(kx, ky, kz);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float W = ar.w[0][a] * ar.w[1][b] * ar.w[2][c];
            const float x0 = (float)a * mp.dx - ar.lp[0], x1 = (float)b * mp.dx - ar.lp[1], x2 = (float)c * mp.dx - ar.lp[2];
            float *g = a0 + AL::at(a, b, c);
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
  if (tid < 216) {
    const int x = tid / 36, y = (tid / 6) % 6, z = tid % 6;
    int slot, cell;
    arena_to_grid<SIDE>(geo.o, x, y, z, slot, cell);
    const int bn = nbr[(size_t)geo.block * 8 + slot];
    const float *a = parena + AL::at(x, y, z);
    if (bn >= 0) {
      float *g = gridB + (size_t)bn * 7 * NC + cell;
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        const float v = a[ch * AL::CH];
        if (v != 0.f) unsafeAtomicAdd(g + ch * NC, v);
      }
    } else if (a[0] != 0.f) {
      staleGCount[9] = 1;  // mass for a node whose block is not in the partition
    }
  }
  ZS_STAMP(3, tEntry);  // [3] entry -> exit (8 waves)
#ifdef ZS_PROBE
  if (threadIdx.x == 0 && (blockIdx.x & 127) == 0) atomicAdd(&g_probe[7], 1ull);  // sampled workgroups
#endif
}
// queue G: gather from grid A with hash queries (stores the full state), then scatter to grid B; queue P: scatter only
template <int SIDE, int SMODEL>
static __global__ __launch_bounds__(256) void g2p2g_stale_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, const float *gridA, float *gridB,
                                                          const int *staleG, const int *staleGCount, const int *staleP,
                                                          const int *stalePCount, int *driftFlag) {
  const int ng = *staleGCount, np = *stalePCount;
  if (driftFlag && blockIdx.x == 0 && threadIdx.x == 0) {  // status words for the host: [0] drift flag, [1] exact-path particles
    if (staleGCount[8]) driftFlag[0] = 1;
    atomicAdd(&driftFlag[1], ng + np);
    if (staleGCount[9]) driftFlag[2] = 1;
  }
  const float dxi = mp.dxi;
  const float D_inv = mp.D_inv;
  // queue G only: exact gather + update; the scatter of both queues follows in stale_scatter_coop_kernel
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ng; j += gridDim.x * blockDim.x)
    g2p_gather_global<SIDE, SMODEL>(mp, ps, (size_t)staleG[j], t, gridA, D_inv);
}

// Exact scatter of the queued particles, 32 lanes per particle: lane = stencil node (27 active).  The 8 candidate blocks are
// queried by lanes 0-7 at once, and the three z-neighbours of a node row sit in one 128-B line of the channel, so one atomic
// instruction of a half-wave touches 9 lines instead of the 27 (x 64 particles) of the thread-per-particle form.  Values and
// order of additions per node are those of p2g_scatter_global.
template <int SIDE>
static __global__ __launch_bounds__(256) void stale_scatter_coop_kernel(MpmDev mp, ParticlesDev ps, BhtDev t, float *grid, const int *qa,
                                                                 const int *na, const int *qb, const int *nb, int *status) {
  constexpr int NC = SIDE * SIDE * SIDE;
  const int n0 = *na, n = n0 + *nb;
  const int sub = threadIdx.x & 31;
  const int ngrp = (int)((gridDim.x * blockDim.x) >> 5);
  const float dxi = mp.dxi;
  const float kscale = mp.fscale;
  const int a = sub / 9, b = (sub / 3) % 3, c = sub % 3;  // lanes 27-31 idle
  for (int j = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5); j < n; j += ngrp) {
    const size_t i = (size_t)(j < n0 ? qa[j] : qb[j - n0]);
    float pos[3], vel[3], C[9], contrib[9];
    load_attr<3>(ps.pos, i, pos);
    load_attr<3>(ps.vel, i, vel);
    load_attr<9>(ps.C, i, C);
    {
      float S[STRESS_N];
      load_attr<STRESS_N>(ps.stress, i, S);
      stress_unpack(S, contrib);
    }
    const float mass = ps.mass.base[ps.mass.off(i)];
#pragma unroll
    for (int d = 0; d < 9; ++d) contrib[d] = contrib[d] * kscale;
    Arena ar;
    make_arena(mp.dx, mp.dxi, pos, ar);
    int loc[3], key[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      loc[d] = ar.corner[d] & (SIDE - 1);
      key[d] = (ar.corner[d] - loc[d]) / SIDE * mp.kscale;
    }
    int myblk = -1;
    if (sub < 8) {
      const bool need = (!(sub & 4) || loc[0] + 2 >= SIDE) && (!(sub & 2) || loc[1] + 2 >= SIDE) && (!(sub & 1) || loc[2] + 2 >= SIDE);
      int k[3] = {key[0] + (sub >> 2) * mp.kscale, key[1] + ((sub >> 1) & 1) * mp.kscale, key[2] + (sub & 1) * mp.kscale};
      if (need) myblk = bht_query<3>(t, k);
    }
    const int x = loc[0] + a, y = loc[1] + b, z = loc[2] + c;
    const int o = sub < 27 ? (((x >= SIDE) << 2) | ((y >= SIDE) << 1) | (z >= SIDE)) : 0;
    const int bn = __shfl(myblk, o, 32);
    if (sub < 27 && bn < 0 && status) status[2] = 1;  // a stencil node outside the partition: its contribution is lost
    if (sub < 27 && bn >= 0) {
      const int cell = ((x & (SIDE - 1)) * SIDE + (y & (SIDE - 1))) * SIDE + (z & (SIDE - 1));
      float *g = grid + (size_t)bn * 7 * NC + cell;
      const float xi0 = (float)a * mp.dx - ar.lp[0], xi1 = (float)b * mp.dx - ar.lp[1], xi2 = (float)c * mp.dx - ar.lp[2];
      float W = ar.w[0][a];
      W *= ar.w[1][b];
      W *= ar.w[2][c];
      unsafeAtomicAdd(g, mass * W);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        unsafeAtomicAdd(g + (1 + d) * NC, W * mass * (vel[d] + (C[d] * xi0 + C[3 + d] * xi1 + C[6 + d] * xi2)));
        unsafeAtomicAdd(g + (4 + d) * NC, (contrib[d] * xi0 + contrib[3 + d] * xi1 + contrib[6 + d] * xi2) * W);
      }
    }
  }
}

// stand-alone constitutive update (first step, or after the host changed F / logJp)
template <int SMODEL> __global__ __launch_bounds__(256) void update_stress_kernel(MpmDev mp, ParticlesDev ps) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ps.n) return;
  float F[9], C[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  load_state<model_is_fluid(SMODEL)>(ps.F, i, F);
  if constexpr (model_is_fluid(SMODEL)) load_attr<9>(ps.C, i, C);
  update_stress<SMODEL, 0>(mp, ps, particle_offset<0>(0u, i), F, C);
}

// ======================================================================================= misc kernels
template <int MODEL> __global__ void stress_kernel(MpmDev mp, float *F, float *logJp, size_t n, float *PF) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float f[9], pf[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) f[d] = F[9 * i + d];
  float lj = 0.f;
  if constexpr (model_uses_logjp(MODEL)) lj = logJp[i];
  const float C0[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // the fluid's viscous part needs C: zero through this entry
  model_stress<MODEL, true>(mp.mat, lj, f, pf, C0);
  if constexpr (model_uses_logjp(MODEL)) logJp[i] = lj;
  if constexpr (MODEL != ZS_MPM_FIXED_COROTATED) {  // the plastic models return the projected F
#pragma unroll
    for (int d = 0; d < 9; ++d) F[9 * i + d] = f[d];
  }
#pragma unroll
  for (int d = 0; d < 9; ++d) PF[9 * i + d] = pf[d];
}
static __global__ void svd_kernel(const float *F, size_t n, float *U, float *S, float *V) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float f[9], u[9], s[3], v[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) f[d] = F[9 * i + d];
  svd3(f, u, s, v);
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    U[9 * i + d] = u[d];
    V[9 * i + d] = v[d];
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) S[3 * i + d] = s[d];
}
// owner rank of every particle under the block-aligned box split of zpc_amd/dist.py (cell_box): cell = floor(x / dx) clamped to
// the global box; along axis d the box [lo, hi) is cut at lo + (n k / dims) rounded down to a multiple of `align`
struct OwnerSplit {
  int lo[3], hi[3], dims[3], align;
};
static __global__ __launch_bounds__(256) void owner_rank_kernel(Port<float> pos, size_t n, float dxinv, OwnerSplit sp, int *owner) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[3];
  load_attr<3>(pos, i, p);
  int rc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int c = (int)floorf(p[d] * dxinv);
    c = c < sp.lo[d] ? sp.lo[d] : (c >= sp.hi[d] ? sp.hi[d] - 1 : c);
    const int len = sp.hi[d] - sp.lo[d];
    int r = 0;
    for (int k = 1; k < sp.dims[d]; ++k) {
      int cut = sp.lo[d] + (int)(((long long)len * k) / sp.dims[d]);
      cut = floordiv(cut, sp.align) * sp.align;
      r += c >= cut;
    }
    rc[d] = r;
  }
  owner[i] = (rc[0] * sp.dims[1] + rc[1]) * sp.dims[2] + rc[2];
}
// per-workgroup LDS histogram of the owner ranks, one global atomic per (workgroup, rank that occurs)
static __global__ __launch_bounds__(256) void owner_count_kernel(const int *owner, size_t n, int world, int *counts) {
  extern __shared__ int ocHist[];
  for (int r = threadIdx.x; r < world; r += 256) ocHist[r] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int o = owner[i];
    if ((unsigned)o < (unsigned)world) atomicAdd(&ocHist[o], 1);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < world; r += 256)
    if (ocHist[r]) atomicAdd(&counts[r], ocHist[r]);
}
static __global__ void halo_pack_kernel(const float *grid, const int *blocks, size_t nb, int nc, int chn0, int nchn, float *buf) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)nchn * nc;
  if (g >= nb * per) return;
  const size_t i = g / per, r = g % per;
  buf[g] = grid[((size_t)blocks[i] * 7 + chn0) * nc + r];
}
// MODE 0: set, 1: add (each block appears once in `blocks`), 2: atomic add (the list may name a block several times, e.g. the
// concatenated messages of several peers that all share a corner block)
template <int MODE> __global__ void halo_unpack_kernel(float *grid, const int *blocks, size_t nb, int nc, int chn0, int nchn, const float *buf) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)nchn * nc;
  if (g >= nb * per) return;
  const size_t i = g / per, r = g % per;
  float *dst = grid + ((size_t)blocks[i] * 7 + chn0) * nc + r;
  if constexpr (MODE == 2) unsafeAtomicAdd(dst, buf[g]);
  else if constexpr (MODE == 1) *dst += buf[g];
  else *dst = buf[g];
}

// ======================================================================================= host helpers
// exact-path kernels: grid-stride over a device-side count.  The walk of one particle is a chain of 27 dependent hash
// queries, so the kernel is latency-bound and wants every wave slot of the chip: 8 blocks of 256 per CU (with 256 blocks a
// queue of 640 k particles took 0.37 ms, i.e. half of an 8 M-particle step)
constexpr unsigned STALE_BLOCKS = 2048;
static MpmDev make_dev(const zs_rocm_mpm_params *p) {
  MpmDev d;
  d.model = p->model;
  d.dx = p->dx;
  d.dt = p->dt;
  d.mat.volume = p->volume;
  d.mat.mu = (float)(0.5 * p->E / (1 + p->nu));  // lame_parameters (physics/ConstitutiveModel.hpp:34-38)
  d.mat.lam = (float)(p->E * p->nu / ((1 + p->nu) * (1 - 2 * p->nu)));
  d.mat.cohesion = p->cohesion;
  d.mat.beta = p->beta;
  d.mat.yieldSurface = p->yieldSurface;
  d.mat.volCorrection = p->volCorrection;
  d.mat.yieldStress = p->yieldStress;
  // NACCConfig::bulk() (physics/ConstitutiveModel.hpp:767-769), float arithmetic as written there
  d.mat.bm = 2.f / 3.f * (p->E / (2 * (1 + p->nu))) + (p->E * p->nu / ((1 + p->nu) * (1 - 2 * p->nu)));
  d.mat.xi = p->xi;
  d.mat.Msqr = p->Msqr;
  d.mat.hardeningOn = p->hardeningOn;
  d.mat.bulk = p->bulk;
  d.mat.viscosity = p->viscosity;
  d.kscale = p->keyIsOrigin ? p->side : 1;
  // derived values, in the float arithmetic the kernels used to repeat (IEEE division; fma where the device contracted)
  d.dxi = 1.0f / d.dx;
  d.D_inv = 4.f * d.dxi * d.dxi;
  d.fscale = -d.dt * d.D_inv;
  d.fscaleDx = d.fscale * d.dx;
  d.mat.smu = 2.f * d.mat.mu;
  d.mat.dpCoef = fmaf(3.f, d.mat.lam, d.mat.smu) / d.mat.smu;
  d.mat.expCohesion = expf(d.mat.cohesion);
  return d;
}
static ParticlesDev make_particles(const zs_rocm_particles &p) {
  ParticlesDev d;
  d.mass = make_port<float>(p.mass);
  d.pos = make_port<float>(p.pos);
  d.vel = make_port<float>(p.vel);
  d.C = make_port<float>(p.C);
  d.F = make_port<float>(p.F);
  d.logJp = make_port<float>(p.logJp);
  d.stress = make_port<float>(p.stress);
  d.n = p.n;
  return d;
}

// lane width LW of the fast addressing path (64 or 32) when all used attributes share one TileVector layout, else 0
static int uniform_lane_width(const zs_rocm_particles &p, bool useLogJp, bool useStress) {
  const zs_rocm_attr *a[7] = {&p.mass, &p.pos, &p.vel, &p.C, &p.F, useLogJp ? &p.logJp : nullptr, useStress ? &p.stress : nullptr};
  const zs_rocm_attr &r = p.pos;
  if (r.tileMask != 63u && r.tileMask != 31u) return 0;
  for (auto *q : a) {
    if (!q) continue;
    if (!q->base || q->idx != 0 || q->numTileBits != r.numTileBits || q->tileMask != r.tileMask || q->numChns != r.numChns) return 0;
  }
  if ((1u << r.numTileBits) != r.tileMask + 1u) return 0;
  return (int)r.tileMask + 1;
}
#define ZSR_DISPATCH_LW(lw, CALL, S, M)          \
  do {                                           \
    if ((lw) == 64) { CALL(S, M, 64); }          \
    else if ((lw) == 32) { CALL(S, M, 32); }     \
    else { CALL(S, M, 0); }                      \
  } while (0)

// CALL(SIDE, MODEL) for the runtime (side, model); `other` = the template value for anything that is not one of the four
// constitutive models (MPM_CACHED_STRESS for P2G, -1 = "no constitutive update" for G2P)
#define ZSR_DISPATCH_MODEL_(S, model, other, CALL)                                                        \
  switch (model) {                                                                                        \
    case ZS_MPM_FIXED_COROTATED: { CALL(S, ZS_MPM_FIXED_COROTATED); } break;                              \
    case ZS_MPM_DRUCKER_PRAGER: { CALL(S, ZS_MPM_DRUCKER_PRAGER); } break;                                \
    case ZS_MPM_VONMISES_FIXED_COROTATED: { CALL(S, ZS_MPM_VONMISES_FIXED_COROTATED); } break;            \
    case ZS_MPM_NACC: { CALL(S, ZS_MPM_NACC); } break;                                                    \
    case ZS_MPM_EQUATION_OF_STATE: { CALL(S, ZS_MPM_EQUATION_OF_STATE); } break;                          \
    case MPM_FLUID_NO_STRESS: { CALL(S, MPM_FLUID_NO_STRESS); } break;                                    \
    default: { CALL(S, other); } break;                                                                   \
  }
// the five constitutive models only (callers reject anything else first)
#define ZSR_DISPATCH_PURE_(S, model, CALL)                                                                \
  switch (model) {                                                                                        \
    case ZS_MPM_FIXED_COROTATED: { CALL(S, ZS_MPM_FIXED_COROTATED); } break;                              \
    case ZS_MPM_DRUCKER_PRAGER: { CALL(S, ZS_MPM_DRUCKER_PRAGER); } break;                                \
    case ZS_MPM_VONMISES_FIXED_COROTATED: { CALL(S, ZS_MPM_VONMISES_FIXED_COROTATED); } break;            \
    case ZS_MPM_NACC: { CALL(S, ZS_MPM_NACC); } break;                                                    \
    default: { CALL(S, ZS_MPM_EQUATION_OF_STATE); } break;                                                \
  }
#define ZSR_DISPATCH_SIDE_PURE(side, model, CALL)        \
  do {                                                   \
    if ((side) == 4) { ZSR_DISPATCH_PURE_(4, model, CALL) } \
    else { ZSR_DISPATCH_PURE_(8, model, CALL) }          \
  } while (0)
#define ZSR_DISPATCH_SIDE_MODEL(side, model, CALL)                          \
  do {                                                                      \
    if ((side) == 4) { ZSR_DISPATCH_MODEL_(4, model, MPM_CACHED_STRESS, CALL) } \
    else { ZSR_DISPATCH_MODEL_(8, model, MPM_CACHED_STRESS, CALL) }         \
  } while (0)
// G2P: third argument = stress model to evaluate at the end (-1: none)
#define ZSR_DISPATCH_SIDE_SMODEL(side, smodel, CALL)                        \
  do {                                                                      \
    if ((side) == 4) { ZSR_DISPATCH_MODEL_(4, smodel, -1, CALL) }           \
    else { ZSR_DISPATCH_MODEL_(8, smodel, -1, CALL) }                       \
  } while (0)

// fused G2P2G launch for one block side: defined in mpm_fused_impl.hpp, instantiated in mpm_fused4.hip / mpm_fused8.hip (the
// 30 instantiations per side of the largest kernel compile in parallel)
struct FusedArgs {
  const float *gridA;
  float *gridB;
  const int *binStart;
  const unsigned *cellCount;
  const int *nbr;
  int *staleG, *staleP, *counts, *driftFlag;
  unsigned nbins;
  int binBase, writeAll, lw, model;
  const int *order;   // re-ordering step: input slot of output slot i (nullptr: in place)
  long long inDelta;  // element offset from the output attribute arrays to the input ones
};
template <int S> void g2p2g_launch_side(Launch &L, const MpmDev &mp, const ParticlesDev &pd, const BhtDev &t, const FusedArgs &a);

}  // namespace zsr
