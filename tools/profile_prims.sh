#!/bin/bash
# rocprofv3 kernel-trace + stats of the primitive benchmarks at one size; summary -> gpurun_out/prof_prims_<n>/
n=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/prof_prims_$n
mkdir -p $D
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $R/tools/bench_prims.py --only prims --sizes $n > $D/out.txt 2> $D/stderr.txt
f=$(find $D -name '*kernel_stats.csv' | head -1)
head -16 "$f" | cut -c1-160
cat $D/out.txt
# the time line of one radix_sort call: start / end of consecutive kernels
t=$(find $D -name '*kernel_trace.csv' | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 14 kernels whose name contains radix
idx = [i for i, r in enumerate(rows) if "radix" in r["Kernel_Name"]]
sel = idx[-16:]
t0 = int(rows[sel[0]]["Start_Timestamp"])
for i in sel:
    r = rows[i]
    print("%8.1f us  +%6.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
PY
find $D -name '*kernel_trace.csv' -size +20M -delete
