#!/bin/bash
# r06: p2g_tile_kernel A/B: product build, measurement builds $EXTRA (1 / 2 / 4 bins per workgroup through ZS_P2G_AB builds), stamps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_mpm_gpu.py -x -q -m gpu 2>&1 | tail -3
P2G="python bench.py --no-cpu-baseline --no-at-rest --compact --unfused --drift 0,0,0 --steps 8 --warmup 2"
run() {  # run <lib name or ""> <group> <tag>
  n=$1; g=$2; tag=$3
  if [ -n "$n" ]; then export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$n.so; else unset ZS_ROCM_LIB; fi
  ZS_ROCM_P2G_GROUP=$g $P2G 2>$O/err_${n:-product}_${g}_$tag.txt | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d.get('roofline',{})
print('${n:-product} G=$g $tag', 'p2g launch ms %.4f frac %.4f step %.3f' % (r.get('launch_ms',-1), r.get('frac',-1), d.get('ms_per_step',-1)))"
  grep "p2g probe" $O/err_${n:-product}_${g}_$tag.txt
}
run "" 2 a; run "" 2 b
for n in $EXTRA; do for g in ${GROUPS_AB:-2}; do run $n $g a; run $n $g b; done; done
export ZS_ROCM_PROBE=1
for g in 2 4; do run p2gprobe $g a; done
