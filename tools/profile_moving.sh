#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench run (slotted storage, moving column); summary -> gpurun_out/prof_<tag>/
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-at-rest > $R/gpurun_out/prof_$tag/bench.json 2> $R/gpurun_out/prof_$tag/stderr.txt
f=$(find $R/gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
head -14 "$f" | cut -c1-200
find $R/gpurun_out/prof_$tag -name '*kernel_trace.csv' -size +20M -delete
