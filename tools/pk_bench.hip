// Issue rate of packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against the scalar forms on gfx950, at the occupancy
// of the fused step (512-thread workgroups, 4 waves per SIMD).  r05: the fused step's time is VALU instructions x 4 cycles, so a
// packed instruction that issues in the same 4 cycles halves the cost of whatever can be paired.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_bench.hip -o /tmp/pk_bench && /tmp/pk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(512, 4) void k(float *out, int n) {
  const int lane = threadIdx.x;
  if constexpr (MODE == 0) {  // 16 independent v_fma_f32
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = lane * 0.001f + j;
    const float s = 1.0001f, t = 0.0001f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = fmaf(a[j], s, t);
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r == 123.456f) out[threadIdx.x] = r;
  } else if constexpr (MODE == 1) {  // 16 independent v_pk_fma_f32 (32 fmas)
    f2 a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = f2{lane * 0.001f + j, lane * 0.002f + j};
    const f2 s = {1.0001f, 1.0002f}, t = {0.0001f, 0.0002f};
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_elementwise_fma(a[j], s, t);
    }
    f2 r = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r.x + r.y == 123.456f) out[threadIdx.x] = r.x;
  } else if constexpr (MODE == 2) {  // rotation by swizzle: [x, y] <- c [x, y] + s [y, -x]: does the swap / negation fold into op_sel / neg?
    f2 a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = f2{lane * 0.001f + j, lane * 0.002f + j};
    const float c = 0.9999f, s = 0.01f + lane * 1e-6f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f2 sw = {a[j].y, -a[j].x};
        a[j] = __builtin_elementwise_fma(f2{s, s}, sw, f2{c, c} * a[j]);
      }
    }
    f2 r = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r.x + r.y == 123.456f) out[threadIdx.x] = r.x;
  } else if constexpr (MODE == 3) {  // the same rotation in scalar code: 4 instructions per pair
    float x[16], y[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { x[j] = lane * 0.001f + j; y[j] = lane * 0.002f + j; }
    const float c = 0.9999f, s = 0.01f + lane * 1e-6f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float nx = fmaf(s, y[j], c * x[j]), ny = fmaf(-s, x[j], c * y[j]);
        x[j] = nx;
        y[j] = ny;
      }
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += x[j] + y[j];
    if (r == 123.456f) out[threadIdx.x] = r;
  } else if constexpr (MODE == 4) {  // v_pk_fma_f32 with wave-uniform multiplier and addend (SGPR pairs): 2 VGPR dwords read per lane
    f2 a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = f2{lane * 0.001f + j, lane * 0.002f + j};
    const f2 s = {1.0001f + n * 1e-9f, 1.0002f + n * 1e-9f}, t = {0.0001f * n, 0.0002f * n};
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_elementwise_fma(a[j], s, t);
    }
    f2 r = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r.x + r.y == 123.456f) out[threadIdx.x] = r.x;
  } else if constexpr (MODE == 5) {  // v_pk_mul_f32, two VGPR pairs
    f2 a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = f2{1.f + lane * 0.001f + j, 1.f + lane * 0.002f + j};
    const f2 s = {1.0001f + lane * 1e-9f, 0.9998f + lane * 1e-9f};
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = a[j] * s;
    }
    f2 r = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r.x + r.y == 123.456f) out[threadIdx.x] = r.x;
  } else if constexpr (MODE == 6) {  // v_fma_f32 with three distinct VGPR sources
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = lane * 0.001f + j;
    const float s = 1.0001f + lane * 1e-9f, t = 0.0001f * lane;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = fmaf(a[j], s, t);
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r == 123.456f) out[threadIdx.x] = r;
  } else {  // MODE 7: the shader clock: s_memtime ticks of 20000 x 16 dependent-free fmas against the 100 MHz wall clock
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = lane * 0.001f + j;
    const float s = 1.0001f + lane * 1e-9f, t = 0.0001f * lane;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = fmaf(a[j], s, t);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    if (r == 123.456f) out[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      reinterpret_cast<unsigned long long *>(out)[64] = c1 - c0;
      reinterpret_cast<unsigned long long *>(out)[65] = w1 - w0;
    }
  }
}

template <int MODE> static float run(float *out, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<MODE><<<1024, 512>>>(out, n / 10);
  hipEventRecord(a);
  k<MODE><<<1024, 512>>>(out, n);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float *out;
  hipMalloc(&out, 8192);
  const int n = 20000;
  // 1024 workgroups x 8 waves on 256 CUs x 4 SIMDs: 8 waves per SIMD in two residency rounds of 4
  const double waveInstrPerSimd = 8.0 * n * 16;
  const float t0 = run<0>(out, n), t1 = run<1>(out, n), t2 = run<2>(out, n), t3 = run<3>(out, n);
  std::printf("v_fma_f32     : %.3f ms  -> %.2f ns per wave-instruction per SIMD\n", t0, t0 * 1e6 / waveInstrPerSimd);
  std::printf("v_pk_fma_f32  : %.3f ms  -> %.2f ns per wave-instruction per SIMD (2 fmas each)\n", t1, t1 * 1e6 / waveInstrPerSimd);
  const float t4 = run<4>(out, n), t5 = run<5>(out, n), t6 = run<6>(out, n), t7 = run<7>(out, n);
  std::printf("v_pk_fma_f32 v,s,s : %.3f ms -> %.2f ns\n", t4, t4 * 1e6 / waveInstrPerSimd);
  std::printf("v_pk_mul_f32 v,v   : %.3f ms -> %.2f ns\n", t5, t5 * 1e6 / waveInstrPerSimd);
  std::printf("v_fma_f32 v,v,v    : %.3f ms -> %.2f ns\n", t6, t6 * 1e6 / waveInstrPerSimd);
  unsigned long long cw[2];
  hipMemcpy(cw, reinterpret_cast<unsigned long long *>(out) + 64, 16, hipMemcpyDeviceToHost);
  std::printf("mode 7: %.3f ms; wave 0: %llu s_memtime ticks, %llu wall ticks (100 MHz) -> %.3f ticks per ns; %.2f ticks per wave-instruction\n", t7,
              cw[0], cw[1], cw[0] / (cw[1] * 10.0), cw[0] / (n * 16.0));
  std::printf("rotation, packed (see ISA for the instruction count): %.3f ms\n", t2);
  std::printf("rotation, scalar (4 per pair)                        : %.3f ms\n", t3);
  return 0;
}
