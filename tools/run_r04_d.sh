#!/bin/bash
# r04 launch-order A/B: XCD-chunked bin mapping (product build) vs plain blockIdx (ablate/libzsrocm_noxcd.so), each with the partition's
# blocks numbered in insertion order / lexicographically / along the Morton curve; stand-alone P2G and the default (slotted, moving) step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
P2G="python $R/bench.py --compact --unfused --drift 0,0,0 --no-at-rest --no-cpu-baseline --steps 10 --warmup 3"
F="python $R/bench.py --no-at-rest --no-cpu-baseline"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-28s ms/step %.3f launch_ms %.3f frac %.3f" % (sys.argv[1], d["ms_per_step"], r["launch_ms"], r["frac"]), r.get("g2p",{}).get("launch_ms"))'
for lib in main noxcd; do
  for order in ins lex morton; do
    export ZS_ROCM_LIB=""; [ $lib = noxcd ] && export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_noxcd.so
    [ -z "$ZS_ROCM_LIB" ] && unset ZS_ROCM_LIB
    unset ZS_ROCM_CANONICAL_PARTITION; [ $order = lex ] && export ZS_ROCM_CANONICAL_PARTITION=1; [ $order = morton ] && export ZS_ROCM_CANONICAL_PARTITION=morton
    timeout 300 $P2G 2> $O/p2g_${lib}_${order}.err | tee $O/p2g_${lib}_${order}.json | python -c "$pick" p2g_${lib}_${order} >> $O/summary.txt
    timeout 300 $F 2> $O/fused_${lib}_${order}.err | tee $O/fused_${lib}_${order}.json | python -c "$pick" fused_${lib}_${order} >> $O/summary.txt
  done
done
unset ZS_ROCM_LIB
export ZS_ROCM_CANONICAL_PARTITION=morton
timeout 900 python -m pytest tests/test_mpm_gpu.py tests/test_containers_gpu.py -x -q -m gpu > $O/t_mpm_morton.log 2>&1; echo "mpm tests (morton) rc=$?" >> $O/summary.txt
# PMC traffic of the stand-alone P2G under the morton + chunked mapping
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$O/pmcp_$c; mkdir -p $out
  timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "p2g_wide_kernel" --pmc $c --output-format csv -d $out -o pmc -- $P2G --steps 4 --warmup 1 > /dev/null 2> $out/stderr.txt
  python3 - $out $c >> $O/summary.txt <<'PY'
import csv, glob, os, sys
v = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2]: v.append(float(r["Counter_Value"]))
print("p2g_wide morton+chunked %s avg %.6g KB over %d launches (x2 for FETCH on gfx950)" % (sys.argv[2], sum(v) / max(len(v), 1), len(v)))
PY
  find $out -name '*.csv' -delete
done
cd $R
cat $O/summary.txt
