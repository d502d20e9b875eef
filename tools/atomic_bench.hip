// How fast does the chip take float atomics in the pattern of the fused step's arena flush?  (r05: the step issues ~380 M lane-atomics --
// 212 M from the per-bin arena flush, 170 M from the movers' list -- next to 14 GB of traffic; is the atomic rate the bound?)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o tools/atomic_bench.bin
// One wave = one "bin": 216 arena nodes x 7 channels into an 8^3-cell block grid [block][channel][cell] (3584 floats per block), lanes =
// consecutive nodes (z fastest, rows of 6).  MODE 0: unsafeAtomicAdd (no return), 1: plain store, 2: atomics but every block visited by 8
// waves in a row (the bins of a block, as the block kernel does).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(float *grid, int nblocks, int nbins) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= nbins) return;
  const int blk = MODE == 2 ? (w >> 3) % nblocks : (int)(((unsigned)w * 2654435761u) % (unsigned)nblocks);
  const int ox = MODE == 2 ? ((w >> 2) & 1) * 4 : 0, oy = MODE == 2 ? ((w >> 1) & 1) * 4 : 0, oz = MODE == 2 ? (w & 1) * 4 : 0;
  for (int n = lane; n < 216; n += 64) {
    const int x = n / 36 + ox, y = (n / 6) % 6 + oy, z = n % 6 + oz;
    const int nb = blk + ((x >> 3) * 4 + (y >> 3) * 2 + (z >> 3)) * 7;   // a neighbour block for the apron nodes
    float *g = grid + (size_t)(nb % nblocks) * 3584 + ((x & 7) * 8 + (y & 7)) * 8 + (z & 7);
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) {
      if (MODE == 1) g[ch * 512] = 1.0f;
      else unsafeAtomicAdd(g + ch * 512, 1.0f);
    }
  }
}
template <int MODE> void run(const char *name, float *grid, int nblocks, int nbins) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<(nbins + 3) / 4, 256>>>(grid, nblocks, nbins);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) k<MODE><<<(nbins + 3) / 4, 256>>>(grid, nblocks, nbins);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double ops = (double)nbins * 216 * 7;
  printf("%-34s %.3f ms for %.0f M lane-ops = %.1f G/s\n", name, ms, ops / 1e6, ops / ms / 1e6);
}
int main() {
  const int nblocks = 27200, nbins = 140481;
  float *grid; hipMalloc(&grid, (size_t)nblocks * 3584 * 4); hipMemset(grid, 0, (size_t)nblocks * 3584 * 4);
  run<0>("atomics, bins in random order", grid, nblocks, nbins);
  run<2>("atomics, 8 bins of a block together", grid, nblocks, nbins);
  run<1>("plain stores, random order", grid, nblocks, nbins);
  return 0;
}
