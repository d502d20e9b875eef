#!/usr/bin/env python3
"""Repeat the 24-step slotted / compact experiment of tests/test_fullsize_gpu.py many times and report, per run, the largest deviation
of its channel sums / sums of squares from the FIRST slotted run (hunting an intermittent difference).  python tools/state_noise_hunt.py reps"""
import sys

import numpy as np

from state_noise_probe import NAMES, bench, ratios


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    base = ["--steps", "24", "--warmup", "0", "--no-cpu-baseline", "--checksum", "--no-at-rest"]
    n = 67_108_864
    ref = None
    for k in range(reps):
        for tag, extra in (("slot", []), ("comp", ["--compact", "--rebin-check", "2"])):
            j = bench(base + extra)
            cs = j["checksum"]
            if ref is None:
                ref = cs
                continue
            rs, rq = ratios(ref, cs, n)
            i, q = int(np.argmax(rs)), int(np.argmax(rq))
            flag = "  <<<<<< OUTLIER" if rs[i] > 1e-5 or rq[q] > 1e-5 else ""
            print("%2d %s  max sum %-6s %.2e   max sq %-6s %.2e  hip_error %s rebins %s%s" % (k, tag, NAMES[i], rs[i], NAMES[q], rq[q], j["hip_error"], j["config"]["rebins"], flag), flush=True)
            if flag:
                print("   sum ratios: " + " ".join("%s=%.1e" % (NAMES[c], rs[c]) for c in range(len(NAMES)) if rs[c] > 1e-6))
                print("   sq  ratios: " + " ".join("%s=%.1e" % (NAMES[c], rq[c]) for c in range(len(NAMES)) if rq[c] > 1e-6), flush=True)


if __name__ == "__main__":
    main()
