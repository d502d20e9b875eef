/*
 * mpm.c -- CPU restatement of the MLS-MPM particle<->grid transfer path of zpc.
 * TEST INFRASTRUCTURE ONLY (see zpc_oracle.h).
 *
 *   orc_svd3                  math/matrix/SVD.hpp:15-1030 (McAdams et al. 2011: 4 cyclic Jacobi sweeps
 *                             with the approximate Givens angle on A^T A, quaternion accumulation,
 *                             singular-value sort, Givens QR)
 *   orc_stress_fixedcorotated physics/ConstitutiveModel_Vol_dP.hpp:10-47
 *   orc_stress_sand           physics/ConstitutiveModel_Vol_dP.hpp:246-326
 *   orc_arena                 simulation/Utils.hpp:47-75 + math/curve/InterpolationKernel.hpp:47-55,93-130
 *   orc_mpm_build_partition   simulation/sparsity/SparsityOp.hpp:59-115
 *   orc_mpm_p2g               simulation/transfer/P2G.hpp:51-125
 *   orc_mpm_grid_update       simulation/grid/GridOp.hpp:71-108
 *   orc_mpm_g2p               simulation/transfer/G2P.hpp:44-83
 *
 * Matrices are 9-vectors in column-major order (M[r + 3*c]) exactly as the reference stores F and C.
 * Build with -ffp-contract=off so that float arithmetic is not fused (the reference is built
 * by plain g++ -O3 on x86-64, which does not contract).
 */
#include "zpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#  include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ SVD */
#define GAMMA 5.8284273147583007813f /* 3 + 2 sqrt 2  (SVD.hpp:33) */
#define CSTAR 0.9238795325112867f    /* cos(pi/8)     (SVD.hpp:30, bits 1064076127) */
#define SSTAR 0.3826834323650898f    /* sin(pi/8)     (SVD.hpp:29, bits 1053028117) */
#define TINY 1.e-20f                 /* SVD.hpp:32 */
#define SMALL 1.e-12f                /* SVD.hpp:31 */

static inline float rsqrtf_(float x) { return 1.0f / sqrtf(x); } /* zs::rsqrt on host */

/* one Jacobi conjugation in the (x,y) plane (rotation about axis z = 3-x-y) of the symmetric
 * matrix S (full storage), accumulating the rotation into quaternion q = (w, v0, v1, v2) */
static void jacobi_conj(int x, int y, int z, float S[3][3], float q[4]) {
  /* approximate Givens half-angle: ch : sh = (s_xx - s_yy) : s_xy / 2  (SVD.hpp:103-131) */
  float sh = S[x][y] * 0.5f;
  float ch = S[x][x] - S[y][y];
  if (!(sh * sh >= TINY)) { sh = 0.f; ch = 1.f; }
  float sh2 = sh * sh, ch2 = ch * ch;
  float w = rsqrtf_(sh2 + ch2);
  sh *= w;
  ch *= w;
  if (ch2 <= GAMMA * sh2) { sh = SSTAR; ch = CSTAR; }
  /* full-angle rotation from the half-angle pair */
  sh2 = sh * sh; ch2 = ch * ch;
  float c = ch2 - sh2, s = 2.f * sh * ch; /* |(c,s)| = ch2 + sh2 = 1 */
  /* S <- Q^T S Q with Q = rotation by (c,s) in the (x,y) plane */
  float sxx = S[x][x], sxy = S[x][y], syy = S[y][y], sxz = S[x][z], syz = S[y][z];
  float t1 = c * sxx + s * sxy, t2 = c * sxy + s * syy;
  float t3 = -s * sxx + c * sxy, t4 = -s * sxy + c * syy;
  S[x][x] = c * t1 + s * t2;
  S[x][y] = S[y][x] = c * t3 + s * t4;
  S[y][y] = -s * t3 + c * t4;
  S[x][z] = S[z][x] = c * sxz + s * syz;
  S[y][z] = S[z][y] = -s * sxz + c * syz;
  /* q <- q * (ch, sh e_z) */
  float qw = q[0], qx = q[1 + x], qy = q[1 + y], qz = q[1 + z];
  q[0] = qw * ch - qz * sh;
  q[1 + x] = qx * ch + qy * sh;
  q[1 + y] = qy * ch - qx * sh;
  q[1 + z] = qz * ch + qw * sh;
}

/* Givens rotation (c,s) zeroing a2 against a1: [c s; -s c]^T [a1;a2] = [r;0]  (SVD.hpp QR stage) */
static void qr_givens(float a1, float a2, float *c, float *s) {
  float rho = sqrtf(a1 * a1 + a2 * a2);
  if (rho > SMALL) { *c = a1 / rho; *s = a2 / rho; }
  else { *c = 1.f; *s = 0.f; }
}

void orc_svd3(const float A[9], float U[9], float Sg[3], float V[9]) {
  /* normal equations S = A^T A (SVD.hpp:52-91) */
  float S[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      S[i][j] = A[0 + 3 * i] * A[0 + 3 * j] + A[1 + 3 * i] * A[1 + 3 * j] + A[2 + 3 * i] * A[2 + 3 * j];
  float q[4] = {1.f, 0.f, 0.f, 0.f};
  for (int sweep = 0; sweep < 4; ++sweep) { /* SVD.hpp:101 */
    jacobi_conj(0, 1, 2, S, q);
    jacobi_conj(1, 2, 0, S, q);
    jacobi_conj(2, 0, 1, S, q);
  }
  /* normalise quaternion -> V */
  float n = rsqrtf_(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float w = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
  float Vm[3][3]; /* Vm[r][c] */
  Vm[0][0] = 1 - 2 * (y * y + z * z); Vm[0][1] = 2 * (x * y - w * z);     Vm[0][2] = 2 * (x * z + w * y);
  Vm[1][0] = 2 * (x * y + w * z);     Vm[1][1] = 1 - 2 * (x * x + z * z); Vm[1][2] = 2 * (y * z - w * x);
  Vm[2][0] = 2 * (x * z - w * y);     Vm[2][1] = 2 * (y * z + w * x);     Vm[2][2] = 1 - 2 * (x * x + y * y);
  /* B = A V */
  float B[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      B[r][c] = A[r + 0] * Vm[0][c] + A[r + 3] * Vm[1][c] + A[r + 6] * Vm[2][c];
  /* sort columns by decreasing norm; each swap negates one column so det(V) stays +1 */
  float rho[3];
  for (int c = 0; c < 3; ++c) rho[c] = B[0][c] * B[0][c] + B[1][c] * B[1][c] + B[2][c] * B[2][c];
#define CSWAP(a, b)                                                             \
  if (rho[a] < rho[b]) {                                                        \
    float tr = rho[a]; rho[a] = rho[b]; rho[b] = tr;                            \
    for (int r = 0; r < 3; ++r) {                                               \
      float tb = B[r][a]; B[r][a] = B[r][b]; B[r][b] = -tb;                     \
      float tv = Vm[r][a]; Vm[r][a] = Vm[r][b]; Vm[r][b] = -tv;                 \
    }                                                                           \
  }
  CSWAP(0, 1) CSWAP(0, 2) CSWAP(1, 2)
#undef CSWAP
  /* QR by three Givens rotations: zero B10, B20, B21; U accumulates the rotations */
  float Um[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int k = 0; k < 3; ++k) {
    int p = pr[k][0], r2 = pr[k][1];
    float c, s;
    qr_givens(B[p][p], B[r2][p], &c, &s);
    for (int col = 0; col < 3; ++col) { /* rows p, r2 of B <- G^T B */
      float bp = B[p][col], br = B[r2][col];
      B[p][col] = c * bp + s * br;
      B[r2][col] = -s * bp + c * br;
    }
    for (int row = 0; row < 3; ++row) { /* columns p, r2 of U <- U G */
      float up = Um[row][p], ur = Um[row][r2];
      Um[row][p] = c * up + s * ur;
      Um[row][r2] = -s * up + c * ur;
    }
  }
  Sg[0] = B[0][0]; Sg[1] = B[1][1]; Sg[2] = B[2][2];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { U[r + 3 * c] = Um[r][c]; V[r + 3 * c] = Vm[r][c]; }
}

/* ------------------------------------------------------------------------ constitutive models */
void orc_lame(float E, float nu, float *mu, float *lam) { /* ConstitutiveModel.hpp:34-38: double math, float result */
  *mu = (float)(0.5 * E / (1 + nu));
  *lam = (float)(E * nu / ((1 + nu) * (1 - 2 * nu)));
}

/* out = M1 diag(d) M2^T  (math/matrix/MatrixUtils.h:26-47) */
static void mat_diag_matT(float *out, const float *m1, const float *d, const float *m2) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      out[r + 3 * c] = m1[r] * d[0] * m2[c] + m1[r + 3] * d[1] * m2[c + 3] + m1[r + 6] * d[2] * m2[c + 6];
}
/* PF^T * volume (ConstitutiveModel_Vol_dP.hpp:37-46) */
static void pft_vol(const float *P, const float *F, float volume, float *PF) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      PF[r + 3 * c] = (P[r] * F[c] + P[r + 3] * F[c + 3] + P[r + 6] * F[c + 6]) * volume;
}

void orc_stress_fixedcorotated(float volume, float mu, float lam, const float F[9], float PF[9]) {
  float U[9], S[3], V[9];
  orc_svd3(F, U, S, V);
  float J = S[0] * S[1] * S[2];
  float scaled_mu = 2.f * mu;
  float scaled_lambda = lam * (J - 1.f);
  float Ph[3];
  Ph[0] = scaled_mu * (S[0] - 1.f) + scaled_lambda * (S[1] * S[2]);
  Ph[1] = scaled_mu * (S[1] - 1.f) + scaled_lambda * (S[0] * S[2]);
  Ph[2] = scaled_mu * (S[2] - 1.f) + scaled_lambda * (S[0] * S[1]);
  float P[9];
  /* P = U diag(Ph) V^T written as Ph_k * U_rk * V_ck (:24-33) */
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      P[r + 3 * c] = Ph[0] * U[r] * V[c] + Ph[1] * U[r + 3] * V[c + 3] + Ph[2] * U[r + 6] * V[c + 6];
  pft_vol(P, F, volume, PF);
}

/* math::sqrtNewtonRaphson<float> (math/MathUtils.h:239-252): what the HOST header uses where the CUDA one calls sqrtf */
static float sqrt_newton(float n) {
  const float relTol = 1e-6f, eps = 128.f * 1.1920929e-07f;
  if (n < -eps) return NAN;
  if (n < eps) return 0.f;
  float xn = 1.f, xnp1 = 0.5f * (xn + n / xn);
  const float tol = (n * relTol > eps) ? n * relTol : eps;
  for (; fabsf(xnp1 - xn) > tol; xnp1 = 0.5f * (xn + n / xn)) xn = xnp1;
  return xnp1;
}

/* compute_stress_vonmisesfixedcorotated.  hostVariant = 1: physics/ConstitutiveModel_Vol_dP.hpp:48-111 (Newton square roots,
   negative discriminant clamped) -- the form oracle/_ref pins; hostVariant = 0: cuda/physics/ConstitutiveModel.hpp:47-116
   (sqrtf) -- the form the GPU path is compared with.  F is projected in place. */
void orc_stress_vonmises(float volume, float mu, float lam, float yieldStress, int hostVariant, float F[9], float PF[9]) {
  float U[9], S[3], V[9], Sc[3];
  orc_svd3(F, U, S, V);
  for (int d = 0; d < 3; ++d) Sc[d] = 1e-4f > S[d] ? 1e-4f : S[d];
  float J = Sc[0] * Sc[1] * Sc[2]; /* prod(): ((1 * s0) * s1) * s2 */
  float tau[3], st[3];
  for (int d = 0; d < 3; ++d) tau[d] = 2 * mu * (Sc[d] - 1) * Sc[d] + lam * (J - 1) * J;
  float tr = tau[0] + tau[1] + tau[2];
  for (int d = 0; d < 3; ++d) st[d] = tau[d] - (tr / 3.f);
  float l2 = st[0] * st[0] + st[1] * st[1] + st[2] * st[2];
  float s_norm = hostVariant ? sqrt_newton(l2) : sqrtf(l2);
  float scaled_tauy = (hostVariant ? sqrt_newton(2.f / (6.f - 3.f)) : sqrtf(2.f / (6.f - 3.f))) * yieldStress;
  if (s_norm - scaled_tauy > 0) {
    float alpha = scaled_tauy / s_norm;
    J = 1.f;
    for (int d = 0; d < 3; ++d) {
      float tau_new = alpha * st[d] + (tr / 3.f);
      float b2m4ac = mu * mu - 2 * mu * (lam * (J - 1) * J - tau_new);
      float sq = hostVariant ? (b2m4ac < 0 ? 0.f : sqrt_newton(b2m4ac)) : sqrtf(b2m4ac);
      S[d] = (mu + sq) / (2 * mu);
    }
    mat_diag_matT(F, U, S, V);
  }
  J = S[0] * S[1] * S[2];
  float smu = 2.f * mu, slam = lam * (J - 1.f), Ph[3], P[9];
  Ph[0] = smu * (S[0] - 1.f) + slam * (S[1] * S[2]);
  Ph[1] = smu * (S[1] - 1.f) + slam * (S[0] * S[2]);
  Ph[2] = smu * (S[2] - 1.f) + slam * (S[0] * S[1]);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) P[r + 3 * c] = Ph[0] * U[r] * V[c] + Ph[1] * U[r + 3] * V[c + 3] + Ph[2] * U[r + 6] * V[c + 6];
  pft_vol(P, F, volume, PF);
}

/* compute_stress_nacc.  hostVariant = 1: physics/ConstitutiveModel_Vol_dP.hpp:113-243, whose yield pressure reads
   p0 = bm * 0.00001 + sin((double)(xi * max(-logJp, 0))); hostVariant = 0: cuda/physics/ConstitutiveModel.hpp:118-243 with
   p0 = bm * (0.00001 + sinh(xi * max(-logJp, 0))) -- the CudaExecutionPolicy behaviour the GPU path follows.  Everything
   after that line is the same code in both headers.  F is projected in place, logJp hardened. */
void orc_stress_nacc(float volume, float mu, float lam, float bm, float xi, float beta, float Msqr, int hardeningOn, int hostVariant,
                     float *logJp, float F[9], float PF[9]) {
  float U[9], S[3], V[9];
  orc_svd3(F, U, S, V);
  float a = -*logJp > 0 ? -*logJp : 0;
  float p0 = hostVariant ? (float)(bm * 0.00001f + sin((double)(xi * a))) : bm * (0.00001f + sinhf(xi * a));
  float p_min = -beta * p0;
  float Je_trial = S[0] * S[1] * S[2];
  float Bh[3] = {S[0] * S[0], S[1] * S[1], S[2] * S[2]};
  float trB = (Bh[0] + Bh[1] + Bh[2]) / 3.f;
  float Jm = mu * powf(Je_trial, -2.f / 3.f);
  float sh[3] = {Jm * (Bh[0] - trB), Jm * (Bh[1] - trB), Jm * (Bh[2] - trB)};
  float psi = bm * 0.5f * (Je_trial - 1.f / Je_trial);
  float p_trial = -psi * Je_trial;
  float ys = 3.f / 2.f * (1 + 2.f * beta);
  float yp = (Msqr * (p_trial - p_min) * (p_trial - p0));
  float sn = sh[0] * sh[0] + sh[1] * sh[1] + sh[2] * sh[2];
  float y = (ys * sn) + yp;
  if (p_trial > p0) {
    float Je_new = sqrtf(-2.f * p0 / bm + 1.f);
    S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
    mat_diag_matT(F, U, S, V);
    if (hardeningOn) *logJp += logf(Je_trial / Je_new);
  } else if (p_trial < p_min) {
    float Je_new = sqrtf(-2.f * p_min / bm + 1.f);
    S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
    mat_diag_matT(F, U, S, V);
    if (hardeningOn) *logJp += logf(Je_trial / Je_new);
  } else if (y >= 1e-4) {
    float Bs = powf(Je_trial, 2.f / 3.f) / mu * sqrtf(-yp / ys) / sqrtf(sn);
    for (int i = 0; i < 3; ++i) S[i] = sqrtf(sh[i] * Bs + trB);
    mat_diag_matT(F, U, S, V);
    if (hardeningOn && p0 > 1e-4 && p_trial < p0 - 1e-4 && p_trial > 1e-4 + p_min) {
      float pc = (1.f - beta) * p0 / 2;
      float q_trial = sqrtf(3.f / 2.f * sn);
      float dir[2] = {pc - p_trial, -q_trial};
      float dn = sqrtf(dir[0] * dir[0] + dir[1] * dir[1]);
      dir[0] /= dn;
      dir[1] /= dn;
      float Cq = Msqr * (pc - p_min) * (pc - p0);
      float Bq = Msqr * dir[0] * (2 * pc - p0 - p_min);
      float Aq = Msqr * dir[0] * dir[0] + (1 + 2 * beta) * dir[1] * dir[1];
      float l1 = (-Bq + sqrtf(Bq * Bq - 4 * Aq * Cq)) / (2 * Aq);
      float l2 = (-Bq - sqrtf(Bq * Bq - 4 * Aq * Cq)) / (2 * Aq);
      float p1 = pc + l1 * dir[0], p2 = pc + l2 * dir[0];
      float pf = (p_trial - pc) * (p1 - pc) > 0 ? p1 : p2;
      float tJ = (-2 * pf / bm + 1);
      float Jf = sqrtf(tJ > 0 ? tJ : -tJ);
      if (Jf > 1e-4) *logJp += logf(Je_trial / Jf);
    }
  }
  float J = S[0] * S[1] * S[2];
  float b[9];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) b[r + 3 * c] = F[r] * F[c] + F[r + 3] * F[c + 3] + F[r + 6] * F[c + 6]; /* MatrixUtils.h:244-256 */
  float trb = (b[0] + b[4] + b[8]) / 3.f;
  b[0] -= trb; b[4] -= trb; b[8] -= trb;
  float dc = mu * powf(J, -2.f / 3.f), ic = bm * .5f * (J * J - 1.f);
  for (int d = 0; d < 9; ++d) PF[d] = (dc * b[d] + ((d & 3) ? 0.f : ic)) * volume;
}
/* NACCConfig::bulk(), Msqr() (physics/ConstitutiveModel.hpp:767-785), dim = 3 */
float orc_nacc_bulk(float E, float nu) { return 2.f / 3.f * (E / (2 * (1 + nu))) + (E * nu / ((1 + nu) * (1 - 2 * nu))); }
float orc_nacc_msqr(float fa) {
  float sin_phi = sinf(fa);
  float mcf = sqrtf(2.f / 3.f) * 2.f * sin_phi / (3.f - sin_phi);
  float M = mcf * 3 / sqrtf(2.f / (6.f - 3));
  return M * M;
}

void orc_stress_sand(float volume, float mu, float lam, float cohesion, float beta, float yieldSurface,
                     int volCorrection, float *logJp, float F[9], float PF[9]) {
  float U[9], S[3], V[9];
  orc_svd3(F, U, S, V);
  float scaled_mu = 2.f * mu;
  float epsilon[3], New_S[3] = {0, 0, 0}, New_F[9];
  for (int i = 0; i < 3; ++i) {
    float abs_S = S[i] > 0 ? S[i] : -S[i];
    abs_S = abs_S > 1e-4 ? abs_S : 1e-4; /* double literal as in the reference (:262) */
    epsilon[i] = logf(abs_S) - cohesion;
  }
  float sum_epsilon = epsilon[0] + epsilon[1] + epsilon[2];
  float trace_epsilon = sum_epsilon + *logJp;
  float epsilon_hat[3];
  for (int i = 0; i < 3; ++i) epsilon_hat[i] = epsilon[i] - (trace_epsilon / 3.f);
  float epsilon_hat_norm = sqrtf(epsilon_hat[0] * epsilon_hat[0] + epsilon_hat[1] * epsilon_hat[1]
                                 + epsilon_hat[2] * epsilon_hat[2]);
  if (trace_epsilon >= 0.f) { /* case II: cone tip */
    New_S[0] = New_S[1] = New_S[2] = expf(cohesion);
    mat_diag_matT(New_F, U, New_S, V);
    memcpy(F, New_F, sizeof(New_F));
    if (volCorrection) *logJp = beta * sum_epsilon + *logJp;
  } else if (mu != 0) {
    *logJp = 0;
    float delta_gamma = epsilon_hat_norm + (3.f * lam + scaled_mu) / scaled_mu * trace_epsilon * yieldSurface;
    float H[3];
    if (delta_gamma <= 0) { /* case I: inside the cone */
      for (int i = 0; i < 3; ++i) H[i] = epsilon[i] + cohesion;
    } else { /* case III: project to the cone surface */
      for (int i = 0; i < 3; ++i) H[i] = epsilon[i] - (delta_gamma / epsilon_hat_norm) * epsilon_hat[i] + cohesion;
    }
    for (int i = 0; i < 3; ++i) New_S[i] = expf(H[i]);
    mat_diag_matT(New_F, U, New_S, V);
    memcpy(F, New_F, sizeof(New_F));
  }
  float New_S_log[3] = {logf(New_S[0]), logf(New_S[1]), logf(New_S[2])};
  float trace_log_S = New_S_log[0] + New_S_log[1] + New_S_log[2];
  float P_hat[3];
  for (int i = 0; i < 3; ++i) P_hat[i] = (scaled_mu * New_S_log[i] + lam * trace_log_S) / New_S[i];
  float P[9];
  mat_diag_matT(P, U, P_hat, V);
  pft_vol(P, F, volume, PF);
}

/* ------------------------------------------------------------------------------------ arena */
/* LocalArena<collocated, quadratic>::init: X = pos/dx; corner = floor(X - 0.5); d0 = X - corner;
 * w = {0.5(1.5-d0)^2, 0.75-(d0-1)^2, 0.5(d0-0.5)^2}; localPos = d0*dx.   w[3*d + k] */
void orc_arena(float dx, const float pos[3], int32_t corner[3], float localPos[3], float w[9]) {
  for (int d = 0; d < 3; ++d) {
    float X = pos[d] / dx;
    corner[d] = (int32_t)floorf(X - 0.5f);
    float lp = X - (float)corner[d];
    /* quadratic_bspline_weights re-derives d0 = x - floor(x - 0.5) from lp (InterpolationKernel.hpp:107) */
    float d0 = lp - (float)((int32_t)floorf(lp - 0.5f));
    w[3 * d + 0] = 0.5f * (1.5f - d0) * (1.5f - d0);
    float d1 = d0 - 1.0f;
    w[3 * d + 1] = 0.75f - d1 * d1;
    float zz = 0.5f + d1;
    w[3 * d + 2] = 0.5f * zz * zz;
    localPos[d] = lp * dx;
  }
}

/* ------------------------------------------------------------------------------------ partition */
static int32_t floordiv(int32_t a, int32_t b) { /* blockid: coord + (coord<0 ? -b+1 : 0), then C division */
  return (a + (a < 0 ? -b + 1 : 0)) / b;
}

void orc_mpm_build_partition(orc_bht *table, const float *pos, size_t n, float dx, int side) {
  float dxinv = 1.0f / dx;
  for (size_t i = 0; i < n; ++i) { /* ComputeSparsity (SparsityOp.hpp:75-84), offset -2, displacement 0.5 */
    int32_t b[3];
    for (int d = 0; d < 3; ++d) {
      int32_t coord = (int32_t)floorf(pos[3 * i + d] * dxinv + 0.5f) + (-2);
      b[d] = floordiv(coord, side);
    }
    orc_bht_insert(table, b);
  }
  int32_t nb = orc_bht_size(table); /* EnlargeSparsity lo=0, hi=2 over the blocks present so far (:95-110) */
  for (int32_t i = 0; i < nb; ++i) {
    int32_t base[3];
    memcpy(base, orc_bht_active_keys(table) + 3 * (size_t)i, 12);
    for (int ddx = 0; ddx < 2; ++ddx)
      for (int ddy = 0; ddy < 2; ++ddy)
        for (int ddz = 0; ddz < 2; ++ddz) {
          int32_t k[3] = {base[0] + ddx, base[1] + ddy, base[2] + ddz};
          orc_bht_insert(table, k);
        }
  }
}

/* ------------------------------------------------------------------------------------ transfers */
static inline void atomic_add_f32(float *dst, float val, int atomic) {
  if (!atomic) { *dst += val; return; }
  /* CAS loop like host atomic_add (execution/Atomics.hpp:61-73) */
  uint32_t *p = (uint32_t *)dst;
  uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED), neu;
  do {
    float f;
    memcpy(&f, &old, 4);
    f += val;
    memcpy(&neu, &f, 4);
  } while (!__atomic_compare_exchange_n(p, &old, neu, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}

/* unpack_coord_in_grid (simulation/Utils.hpp:12-31) + grid_traits::coord_to_cellid (Structure.hpp:323-333) */
static inline int locate(const orc_bht *table, const int32_t coord[3], int side, int32_t *cellid) {
  int32_t loc[3], blk[3];
  for (int d = 0; d < 3; ++d) {
    loc[d] = coord[d] & (side - 1);
    blk[d] = (coord[d] - loc[d]) / side;
  }
  *cellid = (loc[0] * side + loc[1]) * side + loc[2];
  return orc_bht_query(table, blk);
}

/* the constitutive switch shared by P2GTransfer (P2G.hpp:60-103) and the P2C2G functors (P2C2G.hpp:98-142): stress of one particle,
 * contrib = P F^T vol (or the fluid's viscous stress); logJp is updated for the plastic models, F is only a scratch copy */
static void model_contrib(const orc_mpm_params *p, float mu, float lam, const float *C, float *F, float *logJp, float contrib[9]) {
  if (p->model == 4) { /* EquationOfStateConfig, P2G.hpp:60-81: J is kept in component 0 of the F slot */
    float J = F[0];
    float vol = p->volume * J;
    float pressure = p->bulk;
    {
      float J2 = J * J;
      float J4 = J2 * J2;
      pressure = pressure * (1 / (J * J2 * J4) - 1);
    }
    contrib[0] = ((C[0] + C[0]) * p->viscosity - pressure) * vol;
    contrib[1] = (C[1] + C[3]) * p->viscosity * vol;
    contrib[2] = (C[2] + C[6]) * p->viscosity * vol;
    contrib[3] = (C[3] + C[1]) * p->viscosity * vol;
    contrib[4] = ((C[4] + C[4]) * p->viscosity - pressure) * vol;
    contrib[5] = (C[5] + C[7]) * p->viscosity * vol;
    contrib[6] = (C[6] + C[2]) * p->viscosity * vol;
    contrib[7] = (C[7] + C[5]) * p->viscosity * vol;
    contrib[8] = ((C[8] + C[8]) * p->viscosity - pressure) * vol;
  } else if (p->model == 0) {
    orc_stress_fixedcorotated(p->volume, mu, lam, F, contrib);
  } else if (p->model == 2) { /* P2G.hpp:86-88 */
    orc_stress_vonmises(p->volume, mu, lam, p->yieldStress, p->hostVariant, F, contrib);
  } else if (p->model == 3) { /* P2G.hpp:96-101 */
    float lj = *logJp;
    orc_stress_nacc(p->volume, mu, lam, orc_nacc_bulk(p->E, p->nu), p->xi, p->beta, p->Msqr, p->hardeningOn, p->hostVariant, &lj, F, contrib);
    *logJp = lj;
  } else {
    float lj = *logJp;
    orc_stress_sand(p->volume, mu, lam, p->cohesion, p->beta, p->yieldSurface, p->volCorrection, &lj, F, contrib);
    *logJp = lj; /* P2G.hpp:101 -- note: the projected F is NOT written back */
  }
}

void orc_mpm_p2g(const orc_mpm_params *p, const orc_bht *table, size_t n, const float *mass,
                 const float *pos, const float *vel, const float *Cm, const float *Fm, float *logJp,
                 float *grid) {
  const float dx = p->dx, dx_inv = 1.0f / dx;
  const float D_inv = 4.f * dx_inv * dx_inv;
  const int side = p->side, ncell = side * side * side;
  const int atomic = p->nthreads > 1;
  float mu, lam;
  orc_lame(p->E, p->nu, &mu, &lam);
#pragma omp parallel for num_threads(p->nthreads > 1 ? p->nthreads : 1) schedule(static) if (p->nthreads > 1)
  for (size_t i = 0; i < n; ++i) {
    float contrib[9], F[9];
    const float *C = Cm + 9 * i;
    memcpy(F, Fm + 9 * i, 36);
    model_contrib(p, mu, lam, C, F, logJp ? logJp + i : NULL, contrib);
    for (int d = 0; d < 9; ++d) contrib[d] = contrib[d] * -p->dt * D_inv; /* P2G.hpp:105 */
    int32_t corner[3];
    float lp[3], w[9];
    orc_arena(dx, pos + 3 * i, corner, lp, w);
    const float m = mass[i];
    const float *v = vel + 3 * i;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        for (int c = 0; c < 3; ++c) {
          int32_t coord[3] = {corner[0] + a, corner[1] + b, corner[2] + c};
          int32_t cellid;
          int blockno = locate(table, coord, side, &cellid);
          float *blk = grid + (size_t)blockno * 7 * (size_t)ncell;
          float xixp[3] = {(float)a * dx - lp[0], (float)b * dx - lp[1], (float)c * dx - lp[2]};
          float W = 1.f; /* weight_impl: ret = 1; ret *= w_x; ret *= w_y; ret *= w_z */
          W *= w[0 + a]; W *= w[3 + b]; W *= w[6 + c];
          atomic_add_f32(&blk[0 * ncell + cellid], m * W, atomic);
          for (int d = 0; d < 3; ++d) {
            atomic_add_f32(&blk[(1 + d) * ncell + cellid],
                           W * m * (v[d] + (C[d] * xixp[0] + C[3 + d] * xixp[1] + C[6 + d] * xixp[2])), atomic);
            atomic_add_f32(&blk[(4 + d) * ncell + cellid],
                           (contrib[d] * xixp[0] + contrib[3 + d] * xixp[1] + contrib[6 + d] * xixp[2]) * W, atomic);
          }
        }
  }
}

void orc_mpm_grid_update(const orc_mpm_params *p, size_t nblocks, float *grid, const float extf[3],
                         float *maxVelSqr) {
  const int ncell = p->side * p->side * p->side;
  float mx = maxVelSqr ? *maxVelSqr : 0.f;
  for (size_t b = 0; b < nblocks; ++b) {
    float *blk = grid + b * 7 * (size_t)ncell;
    for (int c = 0; c < ncell; ++c) {
      float mass = blk[c];
      if (mass != 0.f) {
        mass = 1.f / mass;
        float vsq = 0.f;
        for (int d = 0; d < 3; ++d) {
          float v = blk[(1 + d) * ncell + c] * mass + extf[d] * p->dt;
          blk[(1 + d) * ncell + c] = v;
          vsq += v * v;
        }
        if (vsq > mx) mx = vsq;
      }
    }
  }
  if (maxVelSqr) *maxVelSqr = mx;
}

void orc_mpm_g2p(const orc_mpm_params *p, const orc_bht *table, size_t n, float *pos, float *vel,
                 float *Cm, float *Fm, const float *grid) {
  const float dx = p->dx, dx_inv = 1.0f / dx;
  const float D_inv = 4.f * dx_inv * dx_inv;
  const int side = p->side, ncell = side * side * side;
#pragma omp parallel for num_threads(p->nthreads > 1 ? p->nthreads : 1) schedule(static) if (p->nthreads > 1)
  for (size_t i = 0; i < n; ++i) {
    float v[3] = {0, 0, 0}, C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int32_t corner[3];
    float lp[3], w[9];
    orc_arena(dx, pos + 3 * i, corner, lp, w);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        for (int c = 0; c < 3; ++c) {
          int32_t coord[3] = {corner[0] + a, corner[1] + b, corner[2] + c};
          int32_t cellid;
          int blockno = locate(table, coord, side, &cellid);
          const float *blk = grid + (size_t)blockno * 7 * (size_t)ncell;
          float xixp[3] = {(float)a * dx - lp[0], (float)b * dx - lp[1], (float)c * dx - lp[2]};
          float W = 1.f;
          W *= w[0 + a]; W *= w[3 + b]; W *= w[6 + c];
          float vi[3] = {blk[1 * ncell + cellid], blk[2 * ncell + cellid], blk[3 * ncell + cellid]};
          for (int d = 0; d < 3; ++d) v[d] += vi[d] * W;
          for (int d = 0; d < 9; ++d) C[d] += W * vi[d % 3] * xixp[d / 3] * D_inv;
        }
    for (int d = 0; d < 3; ++d) pos[3 * i + d] += v[d] * p->dt;
    /* F <- (I + dt C) F   (G2P.hpp:74-78, MatrixUtils.h:136-146) */
    float tmp[9], oldF[9], F[9];
    memcpy(oldF, Fm + 9 * i, 36);
    if (p->model == 4) { /* G2P.hpp:70-74: J <- (1 + tr(C) dt) J */
      memcpy(F, oldF, 36);
      F[0] = (1 + (C[0] + C[4] + C[8]) * p->dt) * oldF[0];
    } else {
      for (int d = 0; d < 9; ++d) tmp[d] = C[d] * p->dt + ((d & 0x3) ? 0.f : 1.f);
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
          F[r + 3 * c] = tmp[r] * oldF[3 * c] + tmp[r + 3] * oldF[3 * c + 1] + tmp[r + 6] * oldF[3 * c + 2];
    }
    memcpy(Fm + 9 * i, F, 36);
    memcpy(vel + 3 * i, v, 12);
    memcpy(Cm + 9 * i, C, 36);
  }
}

/* ------------------------------------------------------------------------------------ gather-style transfers
 * P2C2GTransfer / P2C2GTransferMomentum / P2C2GTransferForce (simulation/transfer/P2C2G.hpp:53-189, :346-439, :547-679) and
 * G2C2PTransfer / PostG2C2PTransfer (simulation/transfer/G2C2P.hpp:59-135, :224-275): linear particle <-> cell-centre weights,
 * 1/8 cell <-> node weights; the launch range is Collapse{nblocks, side^3} over the partition's cells, visited here in
 * (block, cell) order, the 27 buckets in ndrange<3>(3) order, a bucket's particles in ascending id.
 *
 * One deliberate difference, documented in DESIGN.md: the reference functor re-runs the constitutive update for every
 * (cell, particle) pair -- up to 8 times per particle -- and stores logJp each time, so for the plastic models its result
 * depends on the order in which cells are visited.  The restatement evaluates every particle ONCE, from the logJp of the
 * previous step (what a race-free run of the reference computes for every pair that reads before the first write). */
static inline float dinv_axis(float x, float dx, float dx_inv) {
  float r = x - (float)(int32_t)floorf(x * dx_inv + 0.5f) * dx; /* P2C2G.hpp:88 */
  return 2.f / (dx * dx - 2 * r * r);                            /* :89 */
}
static inline int in_kernel_range(const float *posp, const float posc[3], float dx) { /* P2C2G.hpp:72-76 */
  for (int d = 0; d < 3; ++d)
    if (fabsf(posp[d] - posc[d]) > dx) return 0;
  return 1;
}

void orc_mpm_p2c2g(const orc_mpm_params *p, int kind, const orc_bht *table, const orc_hashtable *buckets, const int32_t *offsets,
                   const int32_t *indices, size_t n, const float *mass, const float *pos, const float *vel, const float *Bm,
                   const float *Fm, float *logJp, float *grid) {
  const float dx = p->dx, dx_inv = 1.0f / dx, dt = p->dt;
  const int side = p->side, ncell = side * side * side;
  float mu, lam;
  orc_lame(p->E, p->nu, &mu, &lam);
  /* per particle: contrib of P2C2G.hpp:94-146 (kind 0), :391 (kind 1: C Dinv mass), :589-644 (kind 2: stress only) */
  float *Q = (float *)malloc(sizeof(float) * 9 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    float Dinv[3], C[9], F[9], contrib[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < 3; ++d) Dinv[d] = dinv_axis(pos[3 * i + d], dx, dx_inv);
    for (int d = 0; d < 9; ++d) C[d] = Bm[9 * i + d] * Dinv[d / 3];
    if (kind != 1) {
      memcpy(F, Fm + 9 * i, 36);
      model_contrib(p, mu, lam, C, F, logJp ? logJp + i : NULL, contrib);
      for (int d = 0; d < 9; ++d) contrib[d] *= Dinv[d / 3] * -dt;
    }
    if (kind != 2)
      for (int d = 0; d < 9; ++d) contrib[d] += C[d] * mass[i];
    memcpy(Q + 9 * i, contrib, 36);
  }
  const int32_t *keys = orc_bht_active_keys(table);
  const int nblocks = orc_bht_size(table);
  for (int b = 0; b < nblocks; ++b)
    for (int cell = 0; cell < ncell; ++cell) {
      int32_t coord[3] = {keys[3 * b] * side + cell / (side * side), keys[3 * b + 1] * side + (cell / side) % side,
                          keys[3 * b + 2] * side + cell % side};
      float posc[3], m_c = 0.f, mv_c[3] = {0, 0, 0}, Q_c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, QXp_c[3] = {0, 0, 0};
      for (int d = 0; d < 3; ++d) posc[d] = ((float)coord[d] + 0.5f) * dx;
      for (int o = 0; o < 27; ++o) {
        int32_t bc[3] = {coord[0] - 1 + o / 9, coord[1] - 1 + (o / 3) % 3, coord[2] - 1 + o % 3};
        const int32_t bno = orc_hashtable_query(buckets, bc);
        if (bno < 0) continue;
        for (int st = offsets[bno], ed = offsets[bno + 1]; st != ed; ++st) {
          const int32_t id = indices[st];
          const float *posp = pos + 3 * (size_t)id;
          if (!in_kernel_range(posp, posc, dx)) continue;
          const float *contrib = Q + 9 * (size_t)id;
          float Wpc = 1.f;
          for (int d = 0; d < 3; ++d) {
            const float xabs = fabsf((posc[d] - posp[d]) * dx_inv);
            Wpc *= (kind == 0 || xabs <= 1) ? 1.f - xabs : 0.f; /* P2C2G.hpp:151 (kind 0: no clamp), :396-402, :649-655 */
          }
          if (kind != 2) {
            m_c += mass[id] * Wpc;
            for (int d = 0; d < 3; ++d) mv_c[d] += mass[id] * vel[3 * (size_t)id + d] * Wpc;
          }
          for (int d = 0; d < 3; ++d)
            QXp_c[d] += (contrib[d] * posp[0] + contrib[3 + d] * posp[1] + contrib[6 + d] * posp[2]) * Wpc;
          for (int d = 0; d < 9; ++d) Q_c[d] += contrib[d] * Wpc;
        }
      }
      /* stage 2 (c -> i), P2C2G.hpp:166-187 */
      for (int o = 0; o < 8; ++o) {
        int32_t ci[3] = {coord[0] + (o >> 2), coord[1] + ((o >> 1) & 1), coord[2] + (o & 1)};
        const float posi[3] = {(float)ci[0] * dx, (float)ci[1] * dx, (float)ci[2] * dx};
        int32_t cid;
        const int bno = locate(table, ci, side, &cid);
        if (bno < 0) continue;
        float *blk = grid + (size_t)bno * 7 * (size_t)ncell;
        const float Wci = 1.f / 8;
        if (kind != 2) blk[cid] += m_c * Wci;
        for (int d = 0; d < 3; ++d)
          blk[(1 + d) * ncell + cid]
              += (mv_c[d] + ((Q_c[d] * posi[0] + Q_c[3 + d] * posi[1] + Q_c[6 + d] * posi[2]) - QXp_c[d])) * Wci;
      }
    }
  free(Q);
}

/* G2C2PTransfer (G2C2P.hpp:59-135): v_p and B_p are ACCUMULATED (PreG2C2PTransfer, :215-218, zeroes them first) */
void orc_mpm_g2c2p(const orc_mpm_params *p, const orc_bht *table, const orc_hashtable *buckets, const int32_t *offsets,
                   const int32_t *indices, const float *pos, float *vel, float *Bm, const float *grid) {
  const float dx = p->dx, dx_inv = 1.0f / dx;
  const int side = p->side, ncell = side * side * side;
  const int32_t *keys = orc_bht_active_keys(table);
  const int nblocks = orc_bht_size(table);
  for (int b = 0; b < nblocks; ++b)
    for (int cell = 0; cell < ncell; ++cell) {
      int32_t coord[3] = {keys[3 * b] * side + cell / (side * side), keys[3 * b + 1] * side + (cell / side) % side,
                          keys[3 * b + 2] * side + cell % side};
      float v_c[3] = {0, 0, 0}, vx_c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, posc[3];
      for (int o = 0; o < 8; ++o) {
        int32_t ci[3] = {coord[0] + (o >> 2), coord[1] + ((o >> 1) & 1), coord[2] + (o & 1)};
        const float posi[3] = {(float)ci[0] * dx, (float)ci[1] * dx, (float)ci[2] * dx};
        int32_t cid;
        const int bno = locate(table, ci, side, &cid);
        if (bno < 0) continue;
        const float *blk = grid + (size_t)bno * 7 * (size_t)ncell;
        const float W = 1.f / 8;
        const float v_i[3] = {blk[1 * ncell + cid], blk[2 * ncell + cid], blk[3 * ncell + cid]};
        for (int d = 0; d < 3; ++d) v_c[d] += v_i[d] * W;
        for (int d = 0; d < 9; ++d) vx_c[d] += W * v_i[d % 3] * posi[d / 3];
      }
      for (int d = 0; d < 3; ++d) posc[d] = ((float)coord[d] + 0.5f) * dx;
      for (int o = 0; o < 27; ++o) {
        int32_t bc[3] = {coord[0] - 1 + o / 9, coord[1] - 1 + (o / 3) % 3, coord[2] - 1 + o % 3};
        const int32_t bno = orc_hashtable_query(buckets, bc);
        if (bno < 0) continue;
        for (int st = offsets[bno], ed = offsets[bno + 1]; st != ed; ++st) {
          const int32_t id = indices[st];
          const float *posp = pos + 3 * (size_t)id;
          if (!in_kernel_range(posp, posc, dx)) continue;
          float W = 1.f;
          for (int d = 0; d < 3; ++d) {
            const float xabs = fabsf((posc[d] - posp[d]) * dx_inv);
            W *= xabs <= 1 ? 1.f - xabs : 0.f;
          }
          for (int d = 0; d < 3; ++d) vel[3 * (size_t)id + d] += v_c[d] * W;
          for (int d = 0; d < 9; ++d) Bm[9 * (size_t)id + d] += W * (vx_c[d] - v_c[d % 3] * posp[d / 3]);
        }
      }
    }
}

/* PostG2C2PTransfer (G2C2P.hpp:235-270) */
void orc_mpm_post_g2c2p(const orc_mpm_params *p, size_t n, float *pos, const float *vel, const float *Bm, float *Fm) {
  const float dx = p->dx, dx_inv = 1.0f / dx, dt = p->dt;
  for (size_t i = 0; i < n; ++i) {
    float C[9], Dinv[3];
    for (int d = 0; d < 3; ++d) Dinv[d] = dinv_axis(pos[3 * i + d], dx, dx_inv);
    for (int d = 0; d < 9; ++d) C[d] = Bm[9 * i + d] * Dinv[d / 3];
    if (p->model == 4) {
      Fm[9 * i] = (1 + (C[0] + C[4] + C[8]) * dt) * Fm[9 * i];
    } else {
      float tmp[9], oldF[9], F[9];
      memcpy(oldF, Fm + 9 * i, 36);
      for (int d = 0; d < 9; ++d) tmp[d] = C[d] * dt + ((d & 0x3) ? 0.f : 1.f);
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
          F[r + 3 * c] = tmp[r] * oldF[3 * c] + tmp[r + 3] * oldF[3 * c + 1] + tmp[r + 6] * oldF[3 * c + 2];
      memcpy(Fm + 9 * i, F, 36);
    }
    for (int d = 0; d < 3; ++d) pos[3 * i + d] += vel[3 * i + d] * dt;
  }
}
