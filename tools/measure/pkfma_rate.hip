// measurement only: issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (one workgroup per CU slot, 4 waves per SIMD)
// build: hipcc --offload-arch=gfx950 -O3 -o pkfma_rate tools/measure/pkfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float s) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
  const f2 b = f2{s, s * 0.5f};
  const f2 c = f2{0.25f, 0.125f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MODE == 0) {
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      } else {
        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i].y) : "v"(b.y), "v"(c.y));
      }
    }
  }
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  float *out;
  const int blocks = 256 * 4, iters = 20000;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f);
      else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double fl = 2.0 * 16 * (double)iters * blocks * 256;
      printf("%s: %.3f ms, %.1f TFLOP/s (16 f32 fma per lane per iteration)\n", mode == 0 ? "v_pk_fma_f32 x8 " : "v_fma_f32 x16  ", ms, fl / ms * 1e-9);
    }
  }
  return 0;
}
