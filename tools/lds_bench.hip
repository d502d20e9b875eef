// LDS micro-benchmark: throughput of ds_add_f32 / ds_add_u32 / ds_read+ds_write RMW / ds_write on gfx950.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, int stride) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float *p = sm + w * 2048 + lane * stride;
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float *q = p + j * 64 * (stride > 1 ? 0 : 1) + (stride > 1 ? j : 0);
      if (MODE == 0) atomicAdd(q, v);
      else if (MODE == 1) atomicAdd((unsigned *)q, (unsigned)lane);
      else if (MODE == 2) { float t = *(volatile float *)q; *(volatile float *)q = t + v; }
      else if (MODE == 3) *(volatile float *)q = v;
      else if (MODE == 4) { v += *(volatile float *)q; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sm[5] + v;
}
template <int MODE> void run(const char *name, int stride) {
  float *out; hipMalloc(&out, 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, blocks = 256 * 4;
  k<MODE><<<blocks, 256>>>(out, 10, stride);
  hipEventRecord(a);
  k<MODE><<<blocks, 256>>>(out, iters, stride);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * 4 * iters * 16;           // wave-instructions
  double per_cu = winstr / 256;                              // per CU
  printf("%-28s stride %d: %.3f ms, %.1f ns per wave-instr per CU (~%.1f cycles @2.4GHz)\n", name, stride, ms, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
}
int main() {
  for (int stride : {1, 2, 33}) {
    run<0>("ds_add_f32", stride);
    run<1>("ds_add_u32", stride);
    run<2>("ds_read+add+ds_write", stride);
    run<3>("ds_write_b32", stride);
    run<4>("ds_read_b32", stride);
  }
  return 0;
}
