#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i; rm -rf $O; mkdir -p $O
cd $R
for m in 1 l; do ZS_ROCM_LBVH_SELF=$m timeout 300 python tools/lbvh_fill_probe.py > $O/fill_probe_$m.txt 2>&1; echo "== mode $m"; cat $O/fill_probe_$m.txt; done
timeout 600 python -m pytest tests/test_lbvh_gpu.py -x -q -m gpu > $O/t_lbvh.log 2>&1; echo "lbvh tests rc=$?"; tail -5 $O/t_lbvh.log
ZS_ROCM_LBVH_SELF=1 timeout 600 python tools/bench_prims.py --only lbvh > $O/prims.txt 2>&1; grep -i "iter_neighbors\|self" $O/prims.txt
ZS_ROCM_LBVH_QUERY=l ZS_ROCM_LBVH_SELF=1 timeout 600 python tools/bench_prims.py --only lbvh > $O/prims_ql.txt 2>&1; grep -i "iter_neighbors" $O/prims_ql.txt
