#!/bin/bash
# r04: LBvh self-collision wave walk with node pairs (default) vs single nodes; kernel breakdown of config 5
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lbvh_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "lbvh or self or config5" > $O/t_lbvh.log 2>&1; rc=$?; echo "lbvh tests rc=$rc" >> $O/summary.txt
if [ $rc != 0 ]; then tail -30 $O/t_lbvh.log; exit 1; fi
for m in p s; do
  ZS_ROCM_LBVH_SELF=$m timeout 600 python tools/bench_prims.py --only lbvh --json $O/prims_lbvh_$m.json > $O/prims_lbvh_$m.txt 2>&1
  python - $O/prims_lbvh_$m.json $m >> $O/summary.txt <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    print("mode %s  %-80s %.3f ms" % (sys.argv[2], r["name"], r["ms"]))
PY
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python $R/tools/bench_prims.py --only lbvh > /dev/null 2> $O/stats_stderr.txt
db=$(find $O/stats -name '*.db' | head -1)
python $R/tools/rocpd_stats.py "$db" $O/kernel_stats_lbvh.md > /dev/null
rm -rf $O/stats
cd $R
cat $O/summary.txt; grep -v "at::native" $O/kernel_stats_lbvh.md | head -40
