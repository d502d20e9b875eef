#!/bin/bash
# PMC passes on the fused step's main kernel (whatever variant ZS_ROCM_G2P2G_VARIANT / ZS_ROCM_LIB select); summary to stdout
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  out=$R/gpurun_out/pmc_$tag/$name
  mkdir -p $out
  timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "g2p2g_(binned|rs|persist)_kernel" --pmc $grp --output-format csv -d $out -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/stderr.txt
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f: sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in acc:
    v = acc[c]
    print("%-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
  find $out -name '*.csv' -size +8M -delete
done
