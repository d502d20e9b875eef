// mpm_fused8.hip -- the fused G2P2G kernels for 8^3-cell grid blocks (explicit instantiation of g2p2g_launch_side<8>)
#include "mpm_fused_impl.hpp"

namespace zsr {
template void g2p2g_launch_side<8>(Launch &, const MpmDev &, const ParticlesDev &, const BhtDev &, const FusedArgs &);
}

#ifdef ZS_PROBE  // measurement-only build: read and clear the phase stamps of g2p2g_rs_kernel
extern "C" void zs_rocm_debug_probe(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zsr::g_probe), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(zsr::g_probe), z, sizeof(z));
  }
}
#endif
