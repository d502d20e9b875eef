#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_lbvh_gpu.py tests/test_fullsize_gpu.py tests/test_cpp_face_gpu.py -x -q -m gpu -k "lbvh or self or config5 or iter or refit or build or cpp_face" 2>&1 | tail -3
timeout 300 python tools/bench_prims.py --only lbvh 2>&1 | grep -E "LBvh|config 5"
