#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-28s ms/step %.3f launch_ms %.3f frac %.3f g2p_ms" % (sys.argv[1], d["ms_per_step"], r["launch_ms"], r["frac"]), r.get("g2p",{}).get("launch_ms"))'
B="python bench.py --no-at-rest --no-cpu-baseline"
for lib in "" ploadnt plsnt "" ploadnt plsnt; do
  if [ -n "$lib" ]; then export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$lib.so; else unset ZS_ROCM_LIB; fi
  timeout 200 $B 2>/dev/null | python -c "$pick" "fused_${lib:-product}"
  timeout 200 $B --drift 0,0,0 2>/dev/null | python -c "$pick" "fused_rest_${lib:-product}"
  [ "$lib" != plsnt ] && timeout 200 $B --compact --unfused --drift 0,0,0 2>/dev/null | python -c "$pick" "unfused_${lib:-product}"
done
