#!/bin/bash
# r06: tile-stream P2G (p2g_tile_kernel) against the per-round kernel; parity tests first
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_mpm_gpu.py -x -q -m gpu 2>&1 | tail -5
P2G="python bench.py --no-cpu-baseline --no-at-rest --compact --unfused --drift 0,0,0 --steps 8 --warmup 2"
run() {  # run <lib name or ""> <group> <tile 0/1>
  n=$1; g=$2; tl=$3
  if [ -n "$n" ]; then export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$n.so; else unset ZS_ROCM_LIB; fi
  ZS_ROCM_P2G_TILE=$tl ZS_ROCM_P2G_GROUP=$g $P2G 2>$O/err_${n:-product}_${g}_$tl.txt | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d.get('roofline',{})
print('${n:-product} G=$g tile=$tl', 'p2g launch ms %.4f frac %.4f step %.3f' % (r.get('launch_ms',-1), r.get('frac',-1), d.get('ms_per_step',-1)))"
  grep "p2g probe" $O/err_${n:-product}_${g}_$tl.txt
}
run "" 2 0
for g in 2 1 4; do run "" $g 1; run "" $g 1; done
for g in 2 1 4; do run p2gnb2 $g 1; done
for g in 2 1; do run p2gnb4 $g 1; done
export ZS_ROCM_PROBE=1
for g in 2 1 4; do run p2gprobe $g 1; done
