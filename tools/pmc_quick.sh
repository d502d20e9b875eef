#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/r03/pmcq; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "g2p2g_slot_kernel" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-at-rest "$@" > $out/bench.json 2> $out/stderr.txt
python3 - $out <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-24s last3 avg %.5g   first %.5g  n=%d" % (c, sum(v[-3:]) / 3, v[0], len(v)))
PY
find $out -name '*.csv' -delete
