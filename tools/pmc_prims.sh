#!/bin/bash
# PMC passes for the secondary kernels (tools/bench_prims.py): one counter group per run, kernel trace only.
# usage: tools/pmc_prims.sh <tag> <only: prims|tv|bht|lbvh> "<kernel regex>"
tag=${1:-r01}; only=${2:-bht}; re=${3:-bht_insert}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  out=$R/gpurun_out/pmcp_$tag/$name
  mkdir -p $out
  rocprofv3 --kernel-trace --kernel-include-regex "$re" --pmc $grp --output-format csv -d $out -o pmc -- python $R/tools/bench_prims.py --only $only > $out/bench.txt 2> $out/stderr.txt
  f=$(find $out -name '*counter_collection.csv' | head -1)
  echo "== $grp"
  python3 - "$f" <<'PY'
import csv, sys, collections, re
f = sys.argv[1]
if not f: sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r.get("Kernel_Name", ""))[-60:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    for c in acc[k]:
        v = acc[k][c]
        print("%-60s %-22s n=%d mean=%.6g max=%.6g" % (k, c, len(v), sum(v) / len(v), max(v)))
PY
  find $out -name '*.csv' -size +8M -delete
done
