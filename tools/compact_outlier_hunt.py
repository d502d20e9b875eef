#!/usr/bin/env python3
"""Repeat the 24-step compact-storage run (timing-driven re-bin controller) and report, per run, how many particles carry a velocity-gradient
entry far outside the distribution, and where they sit (hunting the rare deviation of the compact path seen in GPUTEST_r02)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
extra = sys.argv[2:]
for k in range(reps):
    env = dict(os.environ, ZS_BENCH_OUTLIERS="4", ZS_BENCH_TRACE_NODES="0.05")
    storage = [] if "--slotted" in extra else ["--compact"]   # (--slotted: the default storage of the bench instead)
    args = [x for x in extra if x != "--slotted"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "0", "--no-cpu-baseline", "--checksum", "--no-at-rest"]
                       + storage + (args if (args or not storage) else ["--rebin-check", "2"]), env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stderr.splitlines() if l.startswith("[outliers]")]
    n = [int(l.split(":")[1].split()[0]) for l in line if "particles" in l]
    print(k, n[0] if n else r.stderr[-300:], flush=True)
    if (n and n[0] != 2) or k < 2:
        for l in r.stderr.splitlines():
            if l.startswith("[trace") or l.startswith("[repeat-final]"):
                print("   ", l[:1500], flush=True)
