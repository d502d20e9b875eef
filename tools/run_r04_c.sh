#!/bin/bash
# r04 baseline pass on a fresh box: full GPU suite, driver-shaped bench line, rank proxy, kernel-trace stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/summary.txt
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/summary.txt
B="python bench.py --no-at-rest --no-cpu-baseline"
timeout 600 $B --steps 40 --warmup 5 --cells 64,256,64 > $O/eighth_plain.json 2> $O/eighth_plain.err; echo "eighth rc=$?" >> $O/summary.txt
timeout 600 $B --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 > $O/proxy8.json 2> $O/proxy8.err; echo "proxy8 rc=$?" >> $O/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_moving -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-at-rest > $O/stats_moving_bench.json 2> $O/stats_moving_stderr.txt
db=$(find $O/stats_moving -name '*.db' | head -1)
python $R/tools/rocpd_stats.py "$db" $O/kernel_stats_moving.md > /dev/null
rm -rf $O/stats_moving
cd $R
tail -3 $O/gputest.log
cat $O/summary.txt
tail -c 600 $O/bench_n1.json
