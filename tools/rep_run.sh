#!/bin/bash
# rep_run.sh NAME N: N runs of the default bench on a measurement build; prints ms/step or the tail of stderr when a run fails
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in $(seq 1 $2); do
  ZS_ROCM_LIB=$R/zpc_amd/lib/ablate/libzsrocm_$1.so python bench.py --no-at-rest --no-cpu-baseline --steps 40 $ABFLAGS > /tmp/o.txt 2> /tmp/e.txt
  if grep -q '^{' /tmp/o.txt; then python -c "
import json; d=json.loads([l for l in open('/tmp/o.txt') if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['config']['slot_record_rank0']['periods_with_flag'])"; else echo "$1 FAILED"; tail -5 /tmp/e.txt; fi
done
