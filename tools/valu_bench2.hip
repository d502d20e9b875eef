// VALU issue-rate micro-benchmark #2 (gfx950): SHADER cycles (s_memtime) per wave-instruction for several opcodes at 1..8 waves per
// SIMD -- which f32 opcodes issue at the 2-cycle (SIMD-32) rate and which at 4?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE> __global__ void k(float *out, unsigned long long *cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f, c = 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 2) { REP8(asm volatile("v_add_f32 %0, %0, %9\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %9\n v_add_f32 %3, %3, %9\n v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 3) { REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 4) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
    if (MODE == 5) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 6) { REP8(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_floor_f32 %2, %2\n v_max_f32 %3, %3, %9\n v_sub_f32 %4, %8, %4\n v_and_b32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_lshlrev_b32 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 7) { REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 9) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]\n v_cndmask_b32_e64 %5, %5, %6, s[20:21]\n v_cndmask_b32_e64 %6, %6, %7, s[20:21]\n v_cndmask_b32_e64 %7, %7, %0, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");) }
    if (MODE == 10) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %6, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
    if (MODE == 11) { REP8(asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
    if (MODE == 12) { REP8(asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 13) { REP8(asm volatile("v_floor_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_f32_i32 %1, %1\n v_floor_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_f32_i32 %4, %4\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (MODE == 8) { REP8(asm volatile("v_fma_f32 %0, %1, %8, %9\n v_fma_f32 %1, %2, %8, %9\n v_fma_f32 %2, %3, %8, %9\n v_fma_f32 %3, %4, %8, %9\n v_fma_f32 %4, %5, %8, %9\n v_fma_f32 %5, %6, %8, %9\n v_fma_f32 %6, %7, %8, %9\n v_fma_f32 %7, %0, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char *name) {
  float *out; hipMalloc(&out, 256 * 8 * 4 * 64 * 4 * 4);
  unsigned long long *cyc; hipMalloc(&cyc, 8);
  const int iters = 2000;
  printf("%-34s", name);
  for (int w : {1, 2, 3, 4, 8}) {
    int blocks = 256 * w;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, cyc, 10);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double instr_per_simd = (double)w * iters * 64;
    printf("  w%d: %.2f ns/instr/SIMD, wave0 %.1f cyc/instr", w, ms * 1e6 / instr_per_simd, (double)c / ((double)iters * 64));
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_fma_f32 (in-place chains x8)");
  run<8>("v_fma_f32 (rotating operands)");
  run<1>("v_mul_f32");
  run<2>("v_add_f32");
  run<3>("v_fmac_f32");
  run<4>("v_cndmask_b32");
  run<5>("v_mov_b32");
  run<6>("sub/floor/max/and/add_u32/lshl mix");
  run<7>("rsq/rcp/log/sqrt (transcendental)");
  run<9>("v_cndmask_b32 e64, sgpr pair cond");
  run<10>("v_cmp_lt_f32 + v_cndmask (vcc)");
  run<11>("v_cndmask independent dst (vcc)");
  run<12>("v_max_f32/v_min_f32");
  run<13>("v_floor/v_cvt_i32_f32/v_cvt_f32_i32");
}
