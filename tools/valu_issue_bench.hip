// VALU issue rate of one SIMD of gfx950 against the number of resident waves and the independent instructions a wave offers.
// r05: the fused step was costed at "4 cycles per VALU instruction"; tools/pk_bench.hip showed a SIMD with four ready waves issuing
// an independent v_fma_f32 every ~2 cycles.  This bench separates the two readings: rate(waves per SIMD, ILP per wave).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/valu_issue_bench.hip -o tools/bin/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP> __global__ __launch_bounds__(1024) void k(float *out, int n, unsigned long long *ticks) {
  const int lane = threadIdx.x;
  float a[ILP];
#pragma unroll
  for (int j = 0; j < ILP; ++j) a[j] = lane * 0.001f + j;
  const float s = 1.0001f + lane * 1e-9f, t = 0.0001f * lane;
  const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / ILP; ++r) {
#pragma unroll
      for (int j = 0; j < ILP; ++j) a[j] = fmaf(a[j], s, t);
    }
  }
  const unsigned long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < ILP; ++j) r += a[j];
  if (r == 123.456f) out[threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ticks[0] = w1 - w0;
    ticks[1] = c1 - c0;
  }
}

template <int ILP> static void run(float *out, unsigned long long *ticks, int wavesPerSimd) {
  const int n = 5000;
  // one workgroup per CU (256 of them), 4 x wavesPerSimd waves each; 8 per SIMD = two workgroups of 16 waves
  const int threads = wavesPerSimd >= 4 ? 1024 : wavesPerSimd * 256;
  const int blocks = 256 * (wavesPerSimd > 4 ? wavesPerSimd / 4 : 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<ILP><<<blocks, threads>>>(out, n / 10, ticks);
  hipEventRecord(e0);
  k<ILP><<<blocks, threads>>>(out, n, ticks);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
  const double ns = h[0] * 10.0, instr = n * 64.0, clk = h[1] / ns;
  std::printf("ILP %2d, %d waves/SIMD: kernel %.3f ms = %.2f ns = %.2f cycles per SIMD instruction; wave 0 alone %.3f ms, %.2f cycles per own instruction (%.2f GHz)\n",
              ILP, wavesPerSimd, ms, ms * 1e6 / (instr * wavesPerSimd), ms * 1e6 / (instr * wavesPerSimd) * clk, ns * 1e-6, ns / instr * clk, clk);
}
int main() {
  float *out;
  unsigned long long *ticks;
  hipMalloc(&out, 8192);
  hipMalloc(&ticks, 64);
  for (int w : {1, 2, 4, 8}) {
    run<1>(out, ticks, w);
    run<2>(out, ticks, w);
    run<4>(out, ticks, w);
    run<16>(out, ticks, w);
  }
  return 0;
}
