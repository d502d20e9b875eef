// mpm_fused4.hip -- the fused G2P2G kernels for 4^3-cell grid blocks (explicit instantiation of g2p2g_launch_side<4>)
#include "mpm_fused_impl.hpp"

namespace zsr {
template void g2p2g_launch_side<4>(Launch &, const MpmDev &, const ParticlesDev &, const BhtDev &, const FusedArgs &);
}
