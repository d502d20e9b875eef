#!/usr/bin/env python3
"""Does the state after 24 steps of the compact-storage fused step depend on WHEN the particles were re-binned?  (It must not: bins are a
storage order.)  Forced schedules (bench.py --rebin-at) against the slotted run."""
import sys

import numpy as np

from state_noise_probe import NAMES, bench, ratios

base = ["--steps", "24", "--warmup", "0", "--no-cpu-baseline", "--checksum", "--no-at-rest"]
n = 67_108_864
ref = bench(base)["checksum"]
for sched in sys.argv[1:]:
    j = bench(base + ["--compact", "--rebin-at", sched])
    rs, rq = ratios(ref, j["checksum"], n)
    i, q = int(np.argmax(rs)), int(np.argmax(rq))
    print("rebin-at %-40s rebins %2d  max sum %-6s %.2e   max sq %-6s %.2e" % (sched, j["config"]["rebins"], NAMES[i], rs[i], NAMES[q], rq[q]), flush=True)
    if rq[q] > 1e-5:
        print("   sq  ratios: " + " ".join("%s=%.1e" % (NAMES[c], rq[c]) for c in range(len(NAMES)) if rq[c] > 1e-6), flush=True)
