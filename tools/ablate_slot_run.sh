#!/bin/bash
# run the default bench (no secondary runs) on every measurement build in zpc_amd/lib/ablate/libzsrocm_slot_*.so: ms/step and fused launch ms
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/ablate_slot
for so in $R/zpc_amd/lib/ablate/libzsrocm_slot_*.so; do
  name=$(basename $so .so | sed 's/libzsrocm_slot_//')
  ZS_ROCM_PROBE=1 ZS_BENCH_ABLATION=1 ZS_ROCM_LIB=$so python $R/bench.py --no-at-rest --no-cpu-baseline "$@" > $R/gpurun_out/ablate_slot/$name.json 2> $R/gpurun_out/ablate_slot/$name.err
  python3 - "$R/gpurun_out/ablate_slot/$name" "$name" <<'PY'
import json, sys
base, name = sys.argv[1], sys.argv[2]
try:
    j = json.loads([x for x in open(base + ".json") if x.startswith("{")][-1])
    print("%-28s ms/step %.3f  launch %.3f" % (name, j["ms_per_step"], j["roofline"]["launch_ms"]))
except Exception as e:
    print("%-28s FAILED %s" % (name, open(base + ".err").read()[-600:]))
for l in open(base + ".err"):
    if l.startswith("slot probe"):
        print("   " + l.strip().replace("  ", "\n      "))
PY
done
