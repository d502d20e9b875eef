#!/bin/bash
# PMC on the slotted step's kernels after N steps of motion (last launches only): VALU / busy / thread-cycles; summary to stdout
tag=${1:-x}; steps=${2:-40}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
# counter groups: $PMC_GROUPS (semicolon separated) or the default two
DEF="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
IFS=';' read -ra GROUPS_ <<< "${PMC_GROUPS:-$DEF}"
for grp in "${GROUPS_[@]}"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  out=$R/gpurun_out/pmc_$tag/$name
  mkdir -p $out
  timeout 900 rocprofv3 --kernel-trace --kernel-include-regex "g2p2g_slot" --pmc $grp --output-format csv -d $out -o pmc -- python $R/bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-at-rest $PMC_BENCH_FLAGS > $out/bench.json 2> $out/stderr.txt
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f: sys.exit(0)
rows = list(csv.DictReader(open(f)))
# the last 3 dispatches of each kernel
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = "slot" if "g2p2g_slot" in r["Kernel_Name"] else "mover"
    acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k in acc:
    for c in acc[k]:
        v = sorted(acc[k][c])
        # a dispatch reports one row per XCD/SE: sum rows of the same dispatch
        per = collections.defaultdict(float)
        for d, x in v: per[d] += x
        ds = sorted(per)[-3:]
        print("%-6s %-26s last3 mean=%.6g" % (k, c, sum(per[d] for d in ds) / len(ds)))
PY
  find $out -name '*.csv' -size +8M -delete
done
