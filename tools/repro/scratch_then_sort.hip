// Does zs::sort of 16-byte struct keys depend on what an earlier kernel left in private (scratch) memory?
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/repro/scratch_then_sort.hip -L zpc_amd/lib -lzsrocm -Wl,-rpath,'$ORIGIN' -o zpc_amd/lib/scratch_then_sort
//   zpc_amd/lib/scratch_then_sort <fill value, hex> [n] [words of private array in the first kernel: 4 | 12 | 24 | 96] [sync between: 0 | 1]
// r04: the first kernel's private array is now a parameter -- the two failing builds of r03 used LESS scratch per lane (48-88 B) than the
// struct-key merge kernels that follow (176 B), i.e. the queue's scratch allocation had to GROW between the two; the r03 version of this
// tool dirtied 384 B per lane, more than anything behind it, and could not see that case.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "zensim_rocm/zs_rocm.hpp"

using namespace zs;

// fills a private array that has to live in scratch (dynamic indexing), and reads it back so that nothing is optimised away
template <int W> __global__ __launch_bounds__(1024) void dirty_scratch(unsigned fill, int rot, unsigned *sink) {
  volatile unsigned a[W];
  for (int i = 0; i < W; ++i) a[(i + rot + threadIdx.x) % W] = fill + i;
  unsigned s = 0;
  for (int i = 0; i < W; ++i) s += a[(i * 7 + rot) % W];
  if (s == 12345u) sink[0] = s;
}

struct Key {
  double d;
  int tag;
  int pad;
  __host__ __device__ bool operator<(const Key &o) const { return d < o.d; }
};

int main(int argc, char **argv) {
  const unsigned fill = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 16) : 0u;
  const int n = argc > 2 ? atoi(argv[2]) : 5000;
  auto pol = rocm_exec();
  unsigned *sink;
  (void)hipMalloc((void **)&sink, 4);
  const int words = argc > 3 ? atoi(argv[3]) : 96, syncBetween = argc > 4 ? atoi(argv[4]) : 1;
  hipStream_t st = (hipStream_t)pol.getStream();
  if (words == 4) hipLaunchKernelGGL(dirty_scratch<4>, dim3(1024), dim3(1024), 0, st, fill, 3, sink);
  else if (words == 12) hipLaunchKernelGGL(dirty_scratch<12>, dim3(1024), dim3(1024), 0, st, fill, 3, sink);
  else if (words == 24) hipLaunchKernelGGL(dirty_scratch<24>, dim3(1024), dim3(1024), 0, st, fill, 3, sink);
  else if (words == 96) hipLaunchKernelGGL(dirty_scratch<96>, dim3(1024), dim3(1024), 0, st, fill, 3, sink);
  if (syncBetween) (void)hipDeviceSynchronize();
  std::vector<Key> hs(n);
  unsigned s = 12345u;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    hs[i] = Key{(double)(((int)(s >> 8) % 1000 - 500) % 17), i, 0};
  }
  Key *ds;
  (void)hipMalloc((void **)&ds, sizeof(Key) * hs.size());
  (void)hipMemcpy(ds, hs.data(), sizeof(Key) * hs.size(), hipMemcpyHostToDevice);
  sort(pol, ds, ds + hs.size());
  std::stable_sort(hs.begin(), hs.end());
  std::vector<Key> rs(hs.size());
  (void)hipMemcpy(rs.data(), ds, sizeof(Key) * hs.size(), hipMemcpyDeviceToHost);
  int bad = 0, first = -1;
  for (int i = 0; i < n; ++i)
    if (rs[i].tag != hs[i].tag || rs[i].d != hs[i].d) {
      if (first < 0) first = i;
      ++bad;
    }
  std::printf("fill %08x n %d first-kernel private words %d sync %d: %d mismatches", fill, n, words, syncBetween, bad);
  if (bad) std::printf(" (first at %d: got d %g tag %d pad %d, want d %g tag %d)", first, rs[first].d, rs[first].tag, rs[first].pad, hs[first].d, hs[first].tag);
  std::printf("\n");
  return bad ? 1 : 0;
}
