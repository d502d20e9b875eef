#!/bin/bash
# rocprofv3 kernel-trace + stats of a bench run: tools/profile_r03.sh <tag> [bench.py args...]; summary in gpurun_out/prof_<tag>/
tag=${1:-r03}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-at-rest "$@" > $R/gpurun_out/prof_$tag/bench.json 2> $R/gpurun_out/prof_$tag/stderr.txt
f=$(find $R/gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %5s avg %10.1f us total %8.2f ms %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
find $R/gpurun_out/prof_$tag -name '*kernel_trace.csv' -size +20M -delete
