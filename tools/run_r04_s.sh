#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-28s ms/step %.3f launch_ms %.3f frac %.3f" % (sys.argv[1], d["ms_per_step"], r["launch_ms"], r["frac"]))'
B="python bench.py --no-at-rest --no-cpu-baseline"
ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_sparse4.so timeout 600 python -m pytest tests/test_mpm_gpu.py -x -q -m gpu -k slotted 2>&1 | tail -2
for rep in 1 2; do for lib in "" sparse2 sparse4 sparse8; do
  if [ -n "$lib" ]; then export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$lib.so; else unset ZS_ROCM_LIB; fi
  timeout 200 $B 2>/dev/null | python -c "$pick" "fused_${lib:-product}"
  timeout 200 $B --steps 60 --warmup 60 2>/dev/null | python -c "$pick" "fused120_${lib:-product}"
done; done
