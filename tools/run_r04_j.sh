#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_lbvh_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "lbvh or self or config5 or iter" > $O/t_lbvh.log 2>&1; echo "lbvh tests rc=$?"; tail -3 $O/t_lbvh.log
timeout 300 python tools/lbvh_fill_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/fill_probe.txt
timeout 600 python tools/bench_prims.py --only lbvh 2>&1 | grep -i "lbvh" | tee $O/prims.txt
