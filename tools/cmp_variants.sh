#!/bin/bash
# checksum channels (m, x3, v3) of the bench variants on a small column
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "" "--compact" "--compact --unfused"; do
  python bench.py "$@" --no-cpu-baseline --checksum --no-at-rest $v 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['checksum']; print('%-22s' % '$v', ' '.join('%.6e' % x for x in c[:7]))"
done
