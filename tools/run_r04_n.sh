#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_primitives_gpu.py tests/test_containers_gpu.py tests/test_lbvh_gpu.py tests/test_mpm_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/bench_prims.py 2>&1 | grep -E "exclusive_scan|TileVector|LBvh|config 5"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-28s ms/step %.3f launch_ms %.3f frac %.3f g2p_ms" % (sys.argv[1], d["ms_per_step"], r["launch_ms"], r["frac"]), r.get("g2p",{}).get("launch_ms"))'
timeout 200 python bench.py --no-at-rest --no-cpu-baseline --compact --unfused --drift 0,0,0 2>/dev/null | python -c "$pick" unfused
