#!/bin/bash
# ab_run.sh NAME...: the default bench (moving column) on the measurement builds zpc_amd/lib/ablate/libzsrocm_NAME.so (tools/ab_build.sh); extra bench flags in $ABFLAGS
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for n in "$@"; do
  for rep in 1 2; do
    ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$n.so python bench.py --no-at-rest --no-cpu-baseline $ABFLAGS 2>/tmp/ab_err.txt | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
print('%-16s ms_per_step %.3f launch_ms %.3f movers %.0f' % ('$n', d.get('ms_per_step', -1), d.get('roofline', {}).get('launch_ms', -1), (d.get('config', {}).get('movers_per_step_rank0') or -1)), d.get('slot_stats', ''))"
    [ -s /tmp/ab_err.txt ] && grep -v amdgpu.ids /tmp/ab_err.txt | tail -3
  done
done
