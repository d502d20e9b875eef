#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-28s ms/step %.3f launch_ms %.3f frac %.3f g2p_ms" % (sys.argv[1], d["ms_per_step"], r["launch_ms"], r["frac"]), r.get("g2p",{}).get("launch_ms"))'
B="python bench.py --no-at-rest --no-cpu-baseline"
for lib in "" pstorent "" pstorent; do
  if [ -n "$lib" ]; then export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$lib.so; else unset ZS_ROCM_LIB; fi
  timeout 200 $B 2>/dev/null | python -c "$pick" "fused_${lib:-product}"
  timeout 200 $B --drift 0,0,0 2>/dev/null | python -c "$pick" "fused_rest_${lib:-product}"
  timeout 200 $B --compact --unfused --drift 0,0,0 2>/dev/null | python -c "$pick" "unfused_${lib:-product}"
done
unset ZS_ROCM_LIB
timeout 300 python tools/bench_prims.py --only prims 2>&1 | grep -E "exclusive_scan|reduce"
timeout 600 python -m pytest tests/test_primitives_gpu.py -x -q -m gpu 2>&1 | tail -2
