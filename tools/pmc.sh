#!/bin/bash
# PMC passes (one counter group per run, no extra trace domains) for the bench's hot kernels.
# usage: tools/pmc.sh <tag> "<bench args>"
tag=${1:-r01}; shift
args=${1:---steps 3 --warmup 1 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  out=$R/gpurun_out/pmc_$tag/$name
  mkdir -p $out
  rocprofv3 --kernel-trace --kernel-include-regex "binned_kernel|binned_split|wide_kernel|tv_scale" --pmc $grp --output-format csv -d $out -o pmc -- python $R/bench.py $args > $out/bench.json 2> $out/stderr.txt
  f=$(find $out -name '*counter_collection.csv' | head -1)
  echo "== $grp -> $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f: sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")
    short = None
    for tag in ("g2p2g_binned", "p2g_wide", "p2g_binned_split", "p2g_binned", "g2p_binned", "tv_scale"):
        if tag in k:
            short = tag
            break
    if short:
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    for c in acc[k]:
        v = acc[k][c]
        print("%-18s %-24s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
  find $out -name '*.csv' -size +8M -delete
done
