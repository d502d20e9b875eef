#!/bin/bash
# full GPU suite + the r04 refresh (profiles)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r04k
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04k/gputest.log 2>&1; echo "gputest rc=$?" | tee gpurun_out/r04k/summary.txt; tail -3 gpurun_out/r04k/gputest.log
bash tools/refresh_r04.sh
