cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_mpm_gpu.py tests/test_dist_gpu.py -q -x -k "slot or range_schedules or repartition" 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-at-rest 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('headline ms/step %.4f' % d['ms_per_step'])"; done
for i in 1 2; do python bench.py --no-cpu-baseline --no-at-rest --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('proxy ms/step %.4f' % d['ms_per_step'], d['config']['step_schedule'])"; done
python bench.py --no-cpu-baseline --no-at-rest --steps 40 --warmup 5 --cells 64,256,64 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('eighth, no exchange ms/step %.4f' % d['ms_per_step'])"
