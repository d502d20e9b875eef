#!/bin/bash
# r06: stand-alone G2P, lane = particle (g2p_packed_kernel) against lane = cell (g2p_binned_kernel); -DZS_G2P_AB build (tools/ab_build.sh g2pab "-DZS_G2P_AB" mpm_g2p.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
P="timeout 300 python bench.py --no-cpu-baseline --no-at-rest --compact --unfused --drift 0,0,0 --steps 12 --warmup 8"
run() {
  $P 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d.get('roofline',{})
print('$1', 'g2p ms %.4f frac %.4f  p2g ms %.4f  step %.3f' % (r['g2p']['launch_ms'], r['g2p']['achieved']/8000.0, r.get('launch_ms',-1), d.get('ms_per_step',-1)))"
}
export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_g2pab.so
for rep in 1 2 3 4; do ZS_ROCM_G2P_PACKED=0 run "lane=cell    "; ZS_ROCM_G2P_PACKED=1 run "lane=particle"; done
unset ZS_ROCM_LIB
run "product      "
