#!/usr/bin/env python3
"""Repeat the 24-step compact-storage run with bench.py's ZS_BENCH_AUDIT hook until one run deviates; print what the audit found
(is the deviating step reproducible from its own input state?  which particles / nodes differ from the unfused kernels?)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
extra = sys.argv[2:] or ["--rebin-at", "6,8,12,14,16,18,22,24"]
hits = 0
for k in range(reps):
    env = dict(os.environ, ZS_BENCH_TRACE_NODES="0.05", ZS_BENCH_AUDIT="3", ZS_BENCH_AUDIT_EVERY="1", ZS_BENCH_CHECK_REBIN="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "0", "--no-cpu-baseline", "--checksum", "--no-at-rest",
                        "--compact"] + extra, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stderr.splitlines() if l.startswith("[audit]")]
    extra_lines = [l for l in r.stderr.splitlines() if l.startswith("[audit-every]") or (l.startswith("[rebin-check]") and ("False" in l or "{}" not in l))]
    if lines or extra_lines or k == 0:
        for l in r.stderr.splitlines():
            if l.startswith("[audit-every]") or l.startswith("[rebin-check]"):
                print("   ", l[:900], flush=True)
    print(k, "deviated" if lines else "clean", flush=True)
    for l in lines:
        print("   ", l[:2000], flush=True)
    if lines or extra_lines:
        hits += 1
        if hits >= 2:
            break
