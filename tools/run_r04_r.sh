#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-28s ms/step %.3f launch_ms %.3f frac %.3f g2p_ms" % (sys.argv[1], d["ms_per_step"], r["launch_ms"], r["frac"]), r.get("g2p",{}).get("launch_ms"))'
B="python bench.py --no-at-rest --no-cpu-baseline --compact --unfused --drift 0,0,0"
ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_early.so timeout 600 python -m pytest tests/test_mpm_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for lib in "" early; do for g in 1 2 4; do
  if [ -n "$lib" ]; then export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$lib.so; else unset ZS_ROCM_LIB; fi
  ZS_ROCM_P2G_GROUP=$g timeout 200 $B 2>/dev/null | python -c "$pick" "p2g_${lib:-product}_g$g"
done; done; done
