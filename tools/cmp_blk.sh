cd $GRAFT_REPO_ROOT
for n in q2 blk; do
ZS_ROCM_LIB=$PWD/zpc_amd/lib/ablate/libzsrocm_$n.so timeout 300 python bench.py --no-at-rest --no-cpu-baseline --checksum --slot-stats 2>&1 | tail -1 | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
if not l: print('$n FAILED'); sys.exit()
d = json.loads(l[-1]); print('$n', d['ms_per_step'], d['roofline']['launch_ms'], ['%.7e' % x for x in d['checksum'][:7]], d['config']['slot_record_rank0']['movers_sent'], d['config']['slot_record_rank0']['periods_with_flag'])"
done
