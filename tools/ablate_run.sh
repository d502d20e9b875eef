#!/bin/bash
# runs the default bench against the measurement builds of tools/ablate.sh; one JSON line each into gpurun_out/ablate/
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/ablate
cd $R
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ablate/base.json 2> gpurun_out/ablate/base.err
for f in zpc_amd/lib/ablate/libzsrocm_*.so; do
  n=$(basename $f .so | sed 's/libzsrocm_//')
  ZS_ROCM_PROBE=1 ZS_ROCM_LIB=$R/$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ablate/$n.json 2> gpurun_out/ablate/$n.err
done
for f in gpurun_out/ablate/*.json; do echo "$(basename $f): $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -h probe ${f%.json}.err)"; done
