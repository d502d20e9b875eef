R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ONLY=p2g bash tools/refresh_r06.sh > gpurun_out/refresh_r06_p2g.log 2>&1
O=$R/gpurun_out/r06
B="python $R/bench.py --no-cpu-baseline --no-at-rest"
cp $O/pmc_p2g.json $R/profiles/pmc_p2g.json   # so that the bench line below carries the traffic of THIS build (code-hash checked)
cp $O/pmc_g2p2g.json $R/profiles/pmc_g2p2g.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
$B --drift 0,0,0 > $O/bench_n1_at_rest.json 2>/dev/null
$B --compact --drift 0,0,0 > $O/bench_n1_compact_at_rest.json 2>/dev/null
$B --compact --unfused --drift 0,0,0 > $O/bench_n1_unfused_at_rest.json 2>/dev/null
$B --steps 40 --warmup 5 --cells 64,256,64 > $O/eighth_plain.json 2>/dev/null
$B --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 2>/dev/null | grep '^{' > $O/proxy8.json
python tools/bench_prims.py --json $O/prims.json > $O/prims.txt 2>&1
tail -c 1500 $O/bench_n1.json
