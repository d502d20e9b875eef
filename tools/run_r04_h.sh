#!/bin/bash
# r04: LBvh self-collision wave walk with C walks per wave; scratch-trap repro build (noinline bucket function in the finish kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
for m in 4 1 2 8 l; do
  ZS_ROCM_LBVH_SELF=$m timeout 600 python -m pytest tests/test_lbvh_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "lbvh or self or config5" > $O/t_lbvh_$m.log 2>&1; rc=$?; echo "lbvh tests mode $m rc=$rc" >> $O/summary.txt
  if [ $rc != 0 ]; then tail -30 $O/t_lbvh_$m.log; continue; fi
  ZS_ROCM_LBVH_SELF=$m timeout 600 python tools/bench_prims.py --only lbvh --json $O/prims_lbvh_$m.json > $O/prims_lbvh_$m.txt 2>&1
  python - $O/prims_lbvh_$m.json $m >> $O/summary.txt <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if "self" in r["name"] or "config 5" in r["name"]: print("mode %s  %-80s %.3f ms" % (sys.argv[2], r["name"], r["ms"]))
PY
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python $R/tools/bench_prims.py --only lbvh > /dev/null 2> $O/stats_stderr.txt
db=$(find $O/stats -name '*.db' | head -1)
python $R/tools/rocpd_stats.py "$db" $O/kernel_stats_lbvh.md > /dev/null
rm -rf $O/stats
cd $R
# scratch trap: the product test program against a library whose finish kernel calls a real (noinline) function with a stack
for i in 1 2 3; do
  LD_PRELOAD=$R/zpc_amd/lib/ablate/libzsrocm_rsnoinline.so timeout 300 zpc_amd/lib/test_cpp_face > $O/cpp_face_noinline_$i.txt 2>&1; echo "test_cpp_face with noinline finish kernel, run $i: rc=$? $(tail -1 $O/cpp_face_noinline_$i.txt)" >> $O/summary.txt
done
timeout 300 zpc_amd/lib/test_cpp_face > $O/cpp_face_product.txt 2>&1; echo "test_cpp_face product: rc=$? $(tail -1 $O/cpp_face_product.txt)" >> $O/summary.txt
cat $O/summary.txt; grep -v "at::native" $O/kernel_stats_lbvh.md | head -12
