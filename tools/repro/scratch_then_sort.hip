// Does zs::sort of 16-byte struct keys depend on what an earlier kernel left in private (scratch) memory?
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/repro/scratch_then_sort.hip -L zpc_amd/lib -lzsrocm -Wl,-rpath,'$ORIGIN' -o zpc_amd/lib/scratch_then_sort
//   zpc_amd/lib/scratch_then_sort <fill value, hex> [n]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "zensim_rocm/zs_rocm.hpp"

using namespace zs;

// fills a private array that has to live in scratch (dynamic indexing), and reads it back so that nothing is optimised away
__global__ __launch_bounds__(1024) void dirty_scratch(unsigned fill, int rot, unsigned *sink) {
  volatile unsigned a[96];
  for (int i = 0; i < 96; ++i) a[(i + rot + threadIdx.x) % 96] = fill + i;
  unsigned s = 0;
  for (int i = 0; i < 96; ++i) s += a[(i * 7 + rot) % 96];
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
  hipLaunchKernelGGL(dirty_scratch, dim3(1024), dim3(1024), 0, (hipStream_t)pol.getStream(), fill, 3, sink);
  (void)hipDeviceSynchronize();
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
  std::printf("fill %08x n %d: %d mismatches", fill, n, bad);
  if (bad) std::printf(" (first at %d: got d %g tag %d pad %d, want d %g tag %d)", first, rs[first].d, rs[first].tag, rs[first].pad, hs[first].d, hs[first].tag);
  std::printf("\n");
  return bad ? 1 : 0;
}
