#!/bin/bash
# r06: bht bulk build, 16 M keys: r05 build (agent-scope probes) / cached first look / cooperative (tile) probe
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_bht_gpu.py tests/test_cpp_face_gpu.py tests/test_hashtable_gpu.py "tests/test_fullsize_gpu.py::test_config2_bht_16m_keys" -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
echo "== r05 probes";        ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_bhtprev.so python tools/bench_prims.py --only bht 2>&1 | grep "bht<"
echo "== cached first look"; python tools/bench_prims.py --only bht 2>&1 | grep "bht<"
echo "== tile probe";        ZS_ROCM_BHT_TILE=1 ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_bhtab.so python tools/bench_prims.py --only bht 2>&1 | grep "bht<"
done
