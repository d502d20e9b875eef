#!/bin/bash
# rocprofv3 kernel trace of 200 radix sorts: tools/sort_trace.sh <n> <uniform|sentinel> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
n=$1; k=$2; shift; shift
for e in "$@"; do export "$e"; done
rm -rf /tmp/p_$k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$k -o s -- python $R/tools/sort_one.py $n $k > /dev/null 2>&1
f=$(find /tmp/p_$k -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$n $k $*" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print("%s | %-60s calls %5s avg %8.2f us min %8.2f" % (sys.argv[2], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
