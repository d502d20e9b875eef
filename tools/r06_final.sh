# the round's closing GPU call: every measurement profiles/r06_* quotes, then the bench line with the traffic of THIS build, then the GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/refresh_r06.sh > gpurun_out/refresh_r06.log 2>&1
O=$R/gpurun_out/r06
cp $O/pmc_p2g.json $R/profiles/pmc_p2g.json   # so that the bench line below carries the traffic of THIS build (code-hash checked)
cp $O/pmc_g2p2g.json $R/profiles/pmc_g2p2g.json
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --no-cpu-baseline --no-at-rest --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 --range-schedule in-turn 2>/dev/null | grep '^{' > $O/proxy8_in_turn.json
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06_gputest.log 2>&1
cat gpurun_out/r06_gputest.log
tail -c 1500 $O/bench_n1.json
