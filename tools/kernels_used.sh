#!/bin/bash
# which kernels (and their mean duration) does a bench invocation run?  usage: tools/kernels_used.sh "<bench args>" [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/kused; rm -rf $O; mkdir -p $O
args=$1; shift
env "$@" rocprofv3 --kernel-trace -d $O -o k -- python $R/bench.py $args > $O/bench.json 2> $O/stderr.txt
db=$(find $O -name '*.db' | head -1)
python $R/tools/rocpd_stats.py "$db" | grep zsr | head -12
find $O -name '*.db' -size +20M -delete
