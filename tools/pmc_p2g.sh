#!/bin/bash
# quick PMC pass over the stand-alone P2G kernel (compact storage, column at rest, cached stress)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/r03/pmcp; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "p2g_wide_kernel" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out -o pmc -- python $R/bench.py --compact --unfused --drift 0,0,0 --steps 8 --warmup 2 --no-cpu-baseline --no-at-rest > $out/bench.json 2> $out/stderr.txt
python3 - $out <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-24s avg %.5g  n=%d" % (c, sum(v) / len(v), len(v)))
PY
find $out -name '*.csv' -delete
