#!/bin/bash
# r04: LBvh self-collision broadphase, one walk per wave (scalar / vector node fetch) vs one walk per leaf
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
for m in s v l; do
  ZS_ROCM_LBVH_SELF=$m timeout 600 python -m pytest tests/test_lbvh_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "lbvh or self or config5" > $O/t_lbvh_$m.log 2>&1; echo "lbvh tests mode $m rc=$?" >> $O/summary.txt
  ZS_ROCM_LBVH_SELF=$m timeout 600 python tools/bench_prims.py --only lbvh --json $O/prims_lbvh_$m.json > $O/prims_lbvh_$m.txt 2>&1
  python - $O/prims_lbvh_$m.json $m >> $O/summary.txt <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    print("mode %s  %-80s %.3f ms" % (sys.argv[2], r["name"], r["ms"]))
PY
done
# scratch trap: a first kernel with LESS private memory per lane than the struct-key merge kernels behind it (the allocation has to grow)
for n in 5000 100000; do for w in 4 12 24 96; do for sy in 1 0; do for fill in 0 ffffffff 7fc00000; do
  timeout 60 zpc_amd/lib/scratch_then_sort $fill $n $w $sy >> $O/scratch_repro.txt 2>&1
done; done; done; done
echo "scratch repro: $(grep -c ' 0 mismatches' $O/scratch_repro.txt) clean runs, $(grep -vc ' 0 mismatches' $O/scratch_repro.txt) others" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/t_lbvh_s.log; grep -v ' 0 mismatches' $O/scratch_repro.txt | head
