#!/bin/bash
# rocprofv3 kernel trace of tools/bench_prims.py; per-kernel table -> gpurun_out/prof_prims/stats.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/prof_prims; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O -o prims -- python $R/tools/bench_prims.py > $O/bench.txt 2> $O/stderr.txt
db=$(find $O -name '*.db' | head -1)
python $R/tools/rocpd_stats.py "$db" $O/stats.md | grep zsr | head -30
find $O -name '*.db' -size +20M -delete
