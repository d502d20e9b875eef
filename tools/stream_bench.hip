// HBM streaming ceilings on the box: read-only sum, copy A->B, in-place scale, for 16 B/lane and 4 B/lane accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy16(const float4 *a, float4 *b, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) b[i] = a[i]; }
__global__ void scale16(float4 *a, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) { float4 v = a[i]; v.x *= 1.0001f; v.y *= 1.0001f; v.z *= 1.0001f; v.w *= 1.0001f; a[i] = v; } }
__global__ void copy4(const float *a, float *b, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) b[i] = a[i]; }
__global__ void scale4(float *a, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) a[i] *= 1.0001f; }
__global__ void write16(float4 *b, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t s = (size_t)gridDim.x * blockDim.x; float4 v = {1, 2, 3, 4}; for (; i < n; i += s) b[i] = v; }
__global__ void read16(const float4 *a, float *out, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t s = (size_t)gridDim.x * blockDim.x; float acc = 0; for (; i < n; i += s) { float4 v = a[i]; acc += v.x + v.y + v.z + v.w; } if (acc == 1.234f) out[0] = acc; }
template <class F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 5; }
int main() {
  size_t bytes = (size_t)6 << 30; float *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  size_t n16 = bytes / 16, n4 = bytes / 4;
  for (int grid : {2048, 8192, 65536}) {
    float t;
    t = timeit([&] { read16<<<grid, 256>>>((float4 *)a, b, n16); }); printf("grid %6d read16    %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
    t = timeit([&] { write16<<<grid, 256>>>((float4 *)b, n16); }); printf("grid %6d write16   %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
    t = timeit([&] { copy16<<<grid, 256>>>((float4 *)a, (float4 *)b, n16); }); printf("grid %6d copy16    %.3f ms  %.2f TB/s (R+W)\n", grid, t, 2.0 * bytes / t / 1e9);
    t = timeit([&] { scale16<<<grid, 256>>>((float4 *)a, n16); }); printf("grid %6d scale16   %.3f ms  %.2f TB/s (R+W in place)\n", grid, t, 2.0 * bytes / t / 1e9);
    t = timeit([&] { copy4<<<grid, 256>>>(a, b, n4); }); printf("grid %6d copy4     %.3f ms  %.2f TB/s (R+W)\n", grid, t, 2.0 * bytes / t / 1e9);
    t = timeit([&] { scale4<<<grid, 256>>>(a, n4); }); printf("grid %6d scale4    %.3f ms  %.2f TB/s (R+W in place)\n", grid, t, 2.0 * bytes / t / 1e9);
  }
}
