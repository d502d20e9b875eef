#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench run; summaries land in gpurun_out/prof_<tag>/
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$tag/bench.json 2> $R/gpurun_out/prof_$tag/stderr.txt
find $R/gpurun_out/prof_$tag -name '*stats*' | head
f=$(find $R/gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
head -25 "$f"
# keep the big trace out of the merge-back
find $R/gpurun_out/prof_$tag -name '*kernel_trace.csv' -size +20M -delete
