#!/bin/bash
# p2g_ab.sh NAME...: HIP-event time of the stand-alone P2G launch (compact storage, column at rest, cached stress) on the measurement
# builds zpc_amd/lib/ablate/libzsrocm_NAME.so (tools/ab_build.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for n in "$@"; do for rep in 1 2; do
ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$n.so python bench.py --no-cpu-baseline --no-at-rest --compact --unfused --drift 0,0,0 --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d.get('roofline',{})
print('$n', 'p2g launch ms %.4f frac %.4f step %.3f' % (r.get('launch_ms',-1), r.get('frac',-1), d.get('ms_per_step',-1)))"
done; done
