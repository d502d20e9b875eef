# the rank proxy (one rank's eighth of the 64 Mi column, 8-rank schedule on one GPU): the three range schedules of zs_rocm_mpm_step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-at-rest --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8"
one() { timeout 300 $B --range-schedule $1 2>gpurun_out/err_rs.txt | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d['rank_breakdown']['max_over_ranks']
print('%-12s ms/step %.4f  ' % ('$1', d['ms_per_step']) + '  '.join('%s %.3f' % (k.replace('_ms',''), v) for k, v in r.items() if v is not None))" || tail -5 gpurun_out/err_rs.txt; }
timeout 900 python -m pytest tests/test_dist_gpu.py -q -x -k "range_schedules" 2>&1 | tail -3
for rep in 1 2 3; do
for s in in-turn side-by-side one-launch; do one $s; done
done
timeout 300 python bench.py --no-cpu-baseline --no-at-rest --steps 40 --warmup 5 --cells 64,256,64 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('no proxy (one range, no exchange) ms/step %.4f' % d['ms_per_step'])"
