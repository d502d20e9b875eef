// How fast can the P2G read pattern go with NO compute?  For every 64-particle tile of a TileVector<f32,64> with C channels,
// read `nread` of the channels (rows of 256 B) and fold them into one value per wave.
// hipcc --offload-arch=gfx950 -O3 -o tools/tile_read_bench.bin tools/tile_read_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NREAD, int UNROLL_TILES>
__global__ __launch_bounds__(256) void read_kernel(const float *buf, size_t tiles, int C, const int *chn, float *out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  int ch[NREAD];
#pragma unroll
  for (int k = 0; k < NREAD; ++k) ch[k] = chn[k];
  for (size_t t = wave * UNROLL_TILES; t < tiles; t += nw * UNROLL_TILES) {
    float v[UNROLL_TILES][NREAD];
#pragma unroll
    for (int u = 0; u < UNROLL_TILES; ++u) {
      const float *b = buf + (t + u) * (size_t)C * 64 + lane;
#pragma unroll
      for (int k = 0; k < NREAD; ++k) v[u][k] = (t + u < tiles) ? b[(size_t)ch[k] * 64] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNROLL_TILES; ++u)
#pragma unroll
      for (int k = 0; k < NREAD; ++k) acc += v[u][k];
  }
  if (acc == 123.456f) out[wave] = acc;
}
int main() {
  const size_t n = 64ull << 20;  // 67.1M particles
  const int C = 35;
  const size_t tiles = n / 64;
  float *buf, *out;
  hipMalloc(&buf, tiles * C * 64 * sizeof(float));
  hipMemset(buf, 0, tiles * C * 64 * sizeof(float));
  hipMalloc(&out, 1 << 24);
  int *chn;
  hipMalloc(&chn, 64 * sizeof(int));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, std::vector<int> chs, auto kern, int grid) {
    hipMemcpy(chn, chs.data(), chs.size() * sizeof(int), hipMemcpyHostToDevice);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const float *)buf, tiles, C, (const int *)chn, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double bytes = (double)chs.size() * 4 * n;
    printf("%-44s grid %6d  %7.3f ms  %7.1f GB/s\n", name, grid, best, bytes / best / 1e6);
  };
  std::vector<int> p2g;  // m, x, v, C (0..15) + stress (26..34): the 25 channels the cached-stress P2G reads
  for (int k = 0; k < 16; ++k) p2g.push_back(k);
  for (int k = 26; k < 35; ++k) p2g.push_back(k);
  std::vector<int> all;
  for (int k = 0; k < 35; ++k) all.push_back(k);
  std::vector<int> first25;
  for (int k = 0; k < 25; ++k) first25.push_back(k);
  for (int grid : {2048, 8192, 32768}) {
    run("25 P2G channels of 35, 1 tile in flight/wave", p2g, read_kernel<25, 1>, grid);
    run("25 P2G channels of 35, 2 tiles in flight/wave", p2g, read_kernel<25, 2>, grid);
    run("first 25 channels of 35, 2 tiles/wave", first25, read_kernel<25, 2>, grid);
    run("all 35 channels, 1 tile/wave", all, read_kernel<35, 1>, grid);
  }
  return 0;
}
