# numbering of the blocks that hold particles, A/B (measurement only; ZS_ROCM_HOLDER_ORDER, zpc_amd/mpm.py build_partition)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
U="python bench.py --no-cpu-baseline --no-at-rest --compact --unfused --drift 0,0,0 --steps 12 --warmup 8"
S="python bench.py --no-cpu-baseline --no-at-rest --steps 20 --warmup 5"
unf() { ZS_ROCM_HOLDER_ORDER=$1 timeout 200 $U 2>gpurun_out/err_order2.txt | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d['roofline']
print('unfused [$1] p2g ms %.4f g2p ms %.4f' % (r['launch_ms'], r['g2p']['launch_ms']))" || tail -5 gpurun_out/err_order2.txt; }
stp() { ZS_ROCM_HOLDER_ORDER=$1 timeout 200 $S 2>gpurun_out/err_order2.txt | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1])
print('slotted moving [$1] ms/step %.4f' % d['ms_per_step'])" || tail -5 gpurun_out/err_order2.txt; }
ORD=${ORD:-"012 120 m 012:2,2,2 012:4,4,4 012:1,4,4 012:1,2,2 120:2,2,2 120:4,4,4 120:4,1,4 012:1,1,2 012:1,1,4 012:1,8,8 012:2,8,8"}
if [ -z "$SLOTTED_ONLY" ]; then
for rep in 1 2; do
for o in $ORD; do unf $o; done
done
fi
for rep in $(seq 1 ${REPS:-1}); do
for o in $ORD; do stp $o; done
done
