#!/bin/bash
# tools-style A/B of two builds of the library on the default bench: ab.sh <alt .so>
for i in 1 2; do
  python bench.py --no-at-rest --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('base   ', d['ms_per_step'], d['roofline']['launch_ms'])"
  ZS_ROCM_LIB=$1 python bench.py --no-at-rest --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('alt    ', d['ms_per_step'], d['roofline']['launch_ms'])"
done
