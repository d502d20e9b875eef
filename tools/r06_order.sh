# block numbering A/B: the same binary, the partition's blocks numbered four ways (ZS_ROCM_CANONICAL_PARTITION overrides build_partition's order)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
U="python bench.py --no-cpu-baseline --no-at-rest --compact --unfused --drift 0,0,0 --steps 12 --warmup 8"
S="python bench.py --no-cpu-baseline --no-at-rest --steps 20 --warmup 5"
C="python bench.py --no-cpu-baseline --no-at-rest --compact --drift 0,0,0 --steps 10 --warmup 3"
unf() { ZS_ROCM_CANONICAL_PARTITION=$1 $U 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d['roofline']
print('unfused [$1] p2g ms %.4f g2p ms %.4f' % (r['launch_ms'], r['g2p']['launch_ms']))"; }
stp() { ZS_ROCM_CANONICAL_PARTITION=$1 $2 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1])
print('$3 [$1] ms/step %.4f' % d['ms_per_step'])"; }
for rep in 1 2; do
for o in insertion holders_lex lex; do unf $o; done
done
for rep in 1 2; do
for o in insertion holders_lex lex; do stp $o "$S" "slotted moving"; done
done
for o in insertion holders_lex lex; do stp $o "$C" "compact fused at rest"; done
