#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for lib in scannt2 sortnt1 sortnt2 sortnt3; do
  export ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$lib.so
  echo "== lib ${lib:-product}"
  for rep in 1 2; do timeout 300 python tools/bench_prims.py --only prims 2>&1 | grep -E "radix_sort|exclusive_scan" ; done
done
