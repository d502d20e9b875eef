// How fast does a SIMD of gfx950 issue REAL per-lane code -- the product's 3x3 SVD + Drucker-Prager return mapping (svd3_core, stress_sand of
// zpc_amd/csrc/mpm_device.hpp), registers only, no memory -- against the number of resident waves?  tools/valu_issue_bench.hip: independent
// v_fma_f32 issue every 2.2 cycles per SIMD once two waves are resident (one wave alone: 4.6).  The fused step averages 4.1 cycles per VALU
// instruction per SIMD with four resident waves; this bench tells how much of that is the instruction mix itself.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I zpc_amd/csrc tools/svd_issue_bench.hip -o tools/bin/svd_issue_bench
#include "../zpc_amd/csrc/mpm_device.hpp"
#include <cstdio>
using namespace zsr;

__global__ __launch_bounds__(1024) void k(float *out, int n, unsigned long long *ticks, Material m) {
  const int lane = threadIdx.x;
  float F[9], PF[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) F[d] = (d % 4 == 0 ? 1.f : 0.f) + 0.01f * (float)((lane * 7 + d * 13) % 17 - 8);
  float lj = 0.f;
  const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
    float Fl[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) Fl[d] = F[d];
    stress_sand<false>(m, lj, Fl, PF);
#pragma unroll
    for (int d = 0; d < 9; ++d) F[d] = fmaf(1e-9f, PF[d], F[d]);  // (the next iteration depends on this one)
  }
  const unsigned long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
  float r = lj;
#pragma unroll
  for (int d = 0; d < 9; ++d) r += F[d];
  if (r == 123.456f) out[threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ticks[0] = w1 - w0;
    ticks[1] = c1 - c0;
  }
}

static void run(float *out, unsigned long long *ticks, int wavesPerSimd, const Material &m, int valuPerIter) {
  const int n = 2000;
  const int threads = wavesPerSimd >= 4 ? 1024 : wavesPerSimd * 256;
  const int blocks = 256 * (wavesPerSimd > 4 ? wavesPerSimd / 4 : 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<<<blocks, threads>>>(out, n / 10, ticks, m);
  hipEventRecord(e0);
  k<<<blocks, threads>>>(out, n, ticks, m);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
  const double ns = h[0] * 10.0, clk = h[1] / ns, instr = (double)n * valuPerIter;
  std::printf("%d waves/SIMD: kernel %.3f ms = %.2f ns = %.2f cycles per SIMD VALU instruction (%d per iteration, from the ISA); wave 0: %.2f cycles per own instruction (%.2f GHz)\n",
              wavesPerSimd, ms, ms * 1e6 / (instr * wavesPerSimd), ms * 1e6 / (instr * wavesPerSimd) * clk, valuPerIter, ns / instr * clk, clk);
}
int main(int argc, char **argv) {
  const int valuPerIter = argc > 1 ? std::atoi(argv[1]) : 750;
  float *out;
  unsigned long long *ticks;
  hipMalloc(&out, 8192);
  hipMalloc(&ticks, 64);
  Material m{};
  m.volume = 1e-9f; m.mu = 4e4f; m.lam = 6e4f; m.cohesion = 0.f; m.beta = 1.f; m.yieldSurface = 0.3f; m.volCorrection = 1;
  m.smu = 2.f * m.mu; m.dpCoef = (3.f * m.lam + m.smu) / m.smu; m.expCohesion = 1.f;
  for (int w : {1, 2, 4, 8}) run(out, ticks, w, m, valuPerIter);
  return 0;
}
