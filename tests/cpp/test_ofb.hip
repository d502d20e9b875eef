// ZS_ENABLE_OFB_ACCESS_CHECK (container/Vector.hpp:471-480, TileVector.hpp:738-767): with the option on, an out-of-range access through a
// view prints the reference's message and yields a reference to the top of the address space instead of a neighbour's element.  Only the
// addresses are taken here (dereferencing the sentinel faults by design).
#define ZS_ENABLE_OFB_ACCESS_CHECK 1
#include "zensim_rocm/zs_rocm.hpp"
#include <cstdio>

__global__ void ofb_kernel(zs::VectorView<int> v, zs::TileVectorView<float, 32> t, int *out) {
  out[0] = (&v[5] == &zs::detail::ofb_sentinel<int>());              // 5 of [0, 4)
  out[1] = (&v[1] == v._p + 1);                                      // in range: untouched
  out[2] = (&t(3, 0) == &zs::detail::ofb_sentinel<float>());         // channel 3 of [0, 2)
  out[3] = (&t(1, 40) == &zs::detail::ofb_sentinel<float>());        // element 40 of [0, 40)
  out[4] = (&t(1, 33) == t._p + (1 * 2 + 1) * 32 + 1);               // in range: the layout formula
  out[5] = (&t(0, 1, 32) == &zs::detail::ofb_sentinel<float>());     // lane 32 of [0, 32)
}

int main() {
  using namespace zs;
  Vector<int> v(4, memsrc_e::device);
  TileVector<float, 32> tv({{"a", 1}, {"b", 1}}, 40, memsrc_e::device);
  Vector<int> out(6, memsrc_e::um);
  constexpr auto space = execspace_e::rocm;
  hipLaunchKernelGGL(ofb_kernel, dim3(1), dim3(1), 0, 0, view<space>(v), static_cast<TileVectorView<float, 32>>(view<space>(tv)), out.data());
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  int bad = 0;
  for (int k = 0; k < 6; ++k)
    if (out.data()[k] != 1) {
      std::printf("ofb check %d failed\n", k);
      ++bad;
    }
  std::printf("ofb access checks: %d failures\n", bad);
  return bad ? 1 : 0;
}
