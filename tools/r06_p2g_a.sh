#!/bin/bash
# r06, first look: stamps of p2g_wide_kernel (head / stream / flush) and record prefetch depth 2, for 1 / 2 / 4 bins per workgroup
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06a; mkdir -p $O
P2G="python bench.py --no-cpu-baseline --no-at-rest --compact --unfused --drift 0,0,0 --steps 8 --warmup 2"
run() {  # run <lib name or ""> <group>
  n=$1; g=$2
  if [ -n "$n" ]; then export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$n.so; else unset ZS_ROCM_LIB; fi
  ZS_ROCM_P2G_GROUP=$g $P2G 2>$O/err_${n:-product}_$g.txt | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d.get('roofline',{})
print('${n:-product} G=$g', 'p2g launch ms %.4f frac %.4f step %.3f' % (r.get('launch_ms',-1), r.get('frac',-1), d.get('ms_per_step',-1)))"
  grep "p2g probe" $O/err_${n:-product}_$g.txt
}
for g in 2 1 4; do run "" $g; done
export ZS_ROCM_PROBE=1
for g in 2 1 4; do run p2gprobe $g; done
unset ZS_ROCM_PROBE
for g in 2 1 4; do run p2gd2 $g; run p2gd2 $g; done
