#!/bin/bash
# samples rocm-smi power / sclk while the bench runs a long window (is the fused step power- or issue-limited?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 1 0; do
  echo "=== variant $v"
  ZS_ROCM_G2P2G_VARIANT=$v python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --rebin-check 0 --cells 128,512,128 > /tmp/b_$v.json 2>/tmp/b_$v.err &
  pid=$!
  while kill -0 $pid 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; done | sort | uniq -c | sort -k1,1nr | head -12
  grep -o '"ms_per_step": [0-9.]*' /tmp/b_$v.json; tail -2 /tmp/b_$v.err
done
