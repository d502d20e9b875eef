#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_lbvh_gpu.py -x -q -m gpu 2>&1 | tail -2
for m in s u; do echo "== queries: $m"; ZS_ROCM_LBVH_QUERY=$m timeout 300 python tools/bench_prims.py --only lbvh 2>&1 | grep -E "iter_neighbors"; done
