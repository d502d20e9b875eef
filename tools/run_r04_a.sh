#!/bin/bash
# r04 first GPU pass: conservation tests, quick bench, 480-step closed-loop run
mkdir -p gpurun_out/r04a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mpm_gpu.py -x -q -m gpu -k "slotted" > gpurun_out/r04a/t_slotted.log 2>&1; echo "slotted rc=$?" >> gpurun_out/r04a/summary.txt
timeout 600 python -m pytest tests/test_primitives_gpu.py -x -q -m gpu -k "in_place or small_input" > gpurun_out/r04a/t_prims.log 2>&1; echo "prims rc=$?" >> gpurun_out/r04a/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-at-rest --no-cpu-baseline > gpurun_out/r04a/bench20.json 2> gpurun_out/r04a/bench20.err; echo "bench20 rc=$?" >> gpurun_out/r04a/summary.txt
timeout 900 python bench.py --steps 480 --warmup 0 --no-at-rest --no-cpu-baseline --checksum > gpurun_out/r04a/bench480.json 2> gpurun_out/r04a/bench480.err; echo "bench480 rc=$?" >> gpurun_out/r04a/summary.txt
tail -3 gpurun_out/r04a/t_slotted.log gpurun_out/r04a/t_prims.log
cat gpurun_out/r04a/summary.txt
