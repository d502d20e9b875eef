#!/bin/bash
# One GPU call that regenerates everything under profiles/ that the bench line refers to (run at the end of a round):
#   kernel-trace stats of the default bench command, PMC passes of the fused and the unfused step, the N=1 bench lines.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats_stderr.txt
db=$(find $O/stats -name '*.db' | head -1)
python $R/tools/rocpd_stats.py "$db" $O/kernel_stats_table.md > /dev/null
find $O/stats -name '*.db' -size +20M -delete
cd $R
bash tools/pmc.sh rf "--steps 3 --warmup 1 --no-cpu-baseline" > $O/pmc_fused.txt 2>&1
rm -f $O/pmc_fused.md; python tools/pmc_summary.py gpurun_out/pmc_rf $O/pmc_fused.md --json g2p2g_binned $O/pmc_g2p2g.json 67108864 8 sand
bash tools/pmc.sh ru "--steps 3 --warmup 1 --no-cpu-baseline --unfused" > $O/pmc_unfused.txt 2>&1
rm -f $O/pmc_unfused.md; python tools/pmc_summary.py gpurun_out/pmc_ru $O/pmc_unfused.md --json p2g_wide $O/pmc_p2g.json 67108864 8 sand
cp $O/pmc_g2p2g.json $O/pmc_p2g.json profiles/ 2>/dev/null   # so that the bench lines below carry the fresh traffic numbers
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --unfused --no-cpu-baseline > $O/bench_n1_unfused.json 2> /dev/null
tail -c 600 $O/bench_n1.json
