// VALU issue-rate micro-benchmark on gfx950: cycles per wave-instruction for v_fma_f32 / v_pk_fma_f32 / v_mul+v_add /
// v_cndmask at 1..8 waves per SIMD, independent vs dependent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f, c = 0.5f;
  float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent fma chains
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      }
    } else if (MODE == 1) {  // one dependent chain
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                     : "+v"(a0) : "v"(b), "v"(c));
      }
    } else if (MODE == 2) {  // 4 independent packed fma chains (8 fma results per 4 instr)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
      }
    } else if (MODE == 3) {  // mul/add/cndmask mix, independent
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                     "v_cndmask_b32 %4, %4, %5, vcc\n v_mov_b32 %5, %6\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, int wavesPerSimd) {
  float *out; hipMalloc(&out, 256 * 8 * 4 * 64 * 4 * 4);
  const int iters = 4000;
  int blocks = 256 * wavesPerSimd;  // 256-thread blocks: 4 waves = 1 per SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(out, 10);
  hipEventRecord(a);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)wavesPerSimd * iters * 64;
  printf("%-30s waves/SIMD=%d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wavesPerSimd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32 x8 independent", w);
    run<1>("v_fma_f32 dependent chain", w);
    run<2>("v_pk_fma_f32 x4 independent", w);
    run<3>("mul/add/cndmask/mov mix", w);
  }
}
