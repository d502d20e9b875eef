#!/bin/bash
# One GPU call that regenerates profiles/r01_c2_kernel_stats.md and profiles/r01_pmc_c2.md (gather-style transfers, tools/bench_c2.py)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/c2prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python $R/tools/bench_c2.py > $O/out.json 2> $O/err.txt
db=$(find $O/stats -name "*.db" | head -1); python $R/tools/rocpd_stats.py "$db" $O/table.md > /dev/null; find $O/stats -name "*.db" -delete
cd $R
{ echo "# r01: gather-style transfers (P2C2G / G2C2P), tools/bench_c2.py under rocprofv3 --kernel-trace --stats"; echo
  echo "16 777 216 particles (128^3 cells x 8), 256^3 grid, 8^3 blocks (5832), FixedCorotated; first run in lattice order, second with the"
  echo "particle storage order shuffled (min / max columns = the two runs).  HIP-event times of whole calls (out.json of the same run;"
  echo "the P2C2G / G2C2P calls run on the partition-numbered buckets):"; echo; echo '```'; cat $O/out.json; echo '```'; echo
  grep -v "at::native\|rocprim\|rocclr\|elementwise\|scan_kernel\|tv_from\|sparsity\|build_neighbors\|grid_update_kernel\|Memset\|Memcpy\|p2g_global" $O/table.md; } > profiles/r01_c2_kernel_stats.md
bash tools/pmc_c2.sh > /dev/null 2>&1
cp gpurun_out/pmc_c2/summary.md profiles/r01_pmc_c2.md
cp profiles/r01_c2_kernel_stats.md profiles/r01_pmc_c2.md gpurun_out/
head -c 700 $O/out.json
