// Does an MFMA-fed wave share a SIMD with VALU-bound waves for free?  (r05 design question: the P2G half of the fused step as a
// per-cell contraction W^T Q on v_mfma_f32_32x32x2_f32 while the producer waves run the G2P / SVD arithmetic on the VALU.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_covalu_bench.hip -o /tmp/mfma_bench && /tmp/mfma_bench
// 512-thread workgroups, two per CU: waves 0-3 run NV x 16 independent v_fma_f32, waves 4-7 run NM x {4 ds_read_b32, 3 VALU, 1 MFMA}.
// Also checks the operand / result layout of the 32x32x2 form (asymmetric A and B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 4) void k(float *out, int nv, int nm) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) sm[i] = 1.0f / (1 + (i & 255));
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w < 4) {
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = lane * 0.001f + j;
    const float s = 1.0001f, t = 0.0001f;
    for (int it = 0; it < nv; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = fmaf(a[j], s, t);
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r == 123.456f) out[threadIdx.x] = r;
  } else {
    f32x16 acc = {0};
    const float *pa = sm + (lane & 31) * 257 % 4096, *pb = sm + 4096 + (lane & 31);
    int e = lane >> 5;
    for (int it = 0; it < nm; ++it) {
      const float a = pa[e & 255];
      const float b = pb[(e & 127) * 3] * pb[(e & 127) * 5 + 1] * pb[(e & 127) * 7 + 2];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      e += 2;
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += acc[j];
    if (r == 123.456f) out[threadIdx.x] = r;
  }
}

__global__ void layout(const float *A, const float *B, float *D) {  // A[32][2], B[2][32] row-major -> D[32][32]
  const int l = threadIdx.x;
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

static float run(float *out, int nv, int nm) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<<<512, 512>>>(out, nv / 10, nm / 10);
  hipEventRecord(a);
  k<<<512, 512>>>(out, nv, nm);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float *out;
  hipMalloc(&out, 4096);
  {
    std::vector<float> A(64), B(64), D(1024), R(1024, 0.f);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37f; B[i] = 2 - i * 0.11f; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        const float ref = fmaf(A[i * 2 + 1], B[32 + j], A[i * 2] * B[j]);
        err = fmax(err, fabs(ref - D[i * 32 + j]));
      }
    printf("layout check: max |D - A B| = %g (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row=(r&3)+8(r>>2)+4(l>>5), col=l&31)\n", err);
  }
  const int NV = 20000, NM = 4000;
  const float tv = run(out, NV, 0), tm = run(out, 0, NM), tb = run(out, NV, NM);
  // per SIMD: 2 VALU waves x NV x 16 instructions; 2 MFMA waves x NM MFMAs
  printf("VALU only : %.3f ms  (%.2f ns per v_fma per SIMD)\n", tv, tv * 1e6 / (2.0 * NV * 16));
  printf("MFMA only : %.3f ms  (%.1f ns per MFMA-iteration per SIMD)\n", tm, tm * 1e6 / (2.0 * NM));
  printf("both      : %.3f ms  (sum %.3f, max %.3f)\n", tb, tv + tm, tv > tm ? tv : tm);
  for (int nm : {1000, 2000, 8000}) {
    const float t2 = run(out, NV, nm), t1 = run(out, 0, nm);
    printf("NV %d NM %d: both %.3f ms, mfma-only %.3f, valu-only %.3f\n", NV, nm, t2, t1, tv);
  }
  return 0;
}
