#!/bin/bash
# r04: symmetric cached stress (6 floats) + grouped flush of the wide P2G (ZS_ROCM_P2G_GROUP = 1 | 2 | 4) + cache-policy bits of its loads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
P2G="python $R/bench.py --compact --unfused --drift 0,0,0 --no-at-rest --no-cpu-baseline --steps 10 --warmup 3"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-28s ms/step %.3f launch_ms %.3f frac %.3f g2p_ms" % (sys.argv[1], d["ms_per_step"], r["launch_ms"], r["frac"]), r.get("g2p",{}).get("launch_ms"))'
timeout 900 python -m pytest tests/test_mpm_gpu.py tests/test_c2_gpu.py tests/test_dist_gpu.py -x -q -m gpu > $O/t_mpm.log 2>&1; rc=$?; echo "mpm tests rc=$rc" >> $O/summary.txt
if [ $rc != 0 ]; then tail -30 $O/t_mpm.log; exit 1; fi
for g in 1 2 4; do
  for rep in 1 2; do
    ZS_ROCM_P2G_GROUP=$g timeout 120 $P2G 2> $O/p2g_g$g.err | tee $O/p2g_g$g.json | python -c "$pick" p2g_group$g >> $O/summary.txt
  done
done
for lib in nt sc; do
  for g in 1 4; do
    ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$lib.so ZS_ROCM_P2G_GROUP=$g timeout 120 $P2G 2> $O/p2g_${lib}_g$g.err | tee $O/p2g_${lib}_g$g.json | python -c "$pick" p2g_${lib}_group$g >> $O/summary.txt
  done
done
cd /tmp
for g in 1 4; do
for c in FETCH_SIZE WRITE_SIZE; do
  out=$O/pmcp_${c}_g$g; mkdir -p $out
  ZS_ROCM_P2G_GROUP=$g timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "p2g_wide_kernel" --pmc $c --output-format csv -d $out -o pmc -- $P2G --steps 4 --warmup 1 > /dev/null 2> $out/stderr.txt
  python3 - $out $c $g >> $O/summary.txt <<'PY'
import csv, glob, os, sys
v = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2]: v.append(float(r["Counter_Value"]))
print("p2g_wide group %s %s avg %.6g KB over %d launches (x2 for FETCH on gfx950)" % (sys.argv[3], sys.argv[2], sum(v) / max(len(v), 1), len(v)))
PY
  find $out -name '*.csv' -delete
done
done
cd $R
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu > $O/t_fullsize.log 2>&1; echo "fullsize tests rc=$?" >> $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/t_mpm.log $O/t_fullsize.log
