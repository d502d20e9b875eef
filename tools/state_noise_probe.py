#!/usr/bin/env python3
"""How far do two runs of the SAME 24-step moving-column experiment differ, channel by channel?  (The P2G sums are float atomics: their
order changes from run to run; the Drucker-Prager return mapping is not smooth.)  Prints |a - b| over the tolerance terms that
tests/test_fullsize_gpu.py::_same_state uses, for slotted vs slotted, slotted vs compact + re-bins, compact vs compact.
    python tools/state_noise_probe.py [reps]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = (["m"] + ["x%d" % k for k in range(3)] + ["v%d" % k for k in range(3)] + ["C%d" % k for k in range(9)] + ["F%d" % k for k in range(9)]
         + ["logJp"] + ["PFt%d" % k for k in range(6)])


def bench(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


def ratios(a, b, npart):
    a, b = np.array(a), np.array(b)
    nch = len(a) // 2
    scale = np.sqrt(npart * np.maximum(a[nch:], 1e-30))
    return np.abs(a[:nch] - b[:nch]) / scale, np.abs(a[nch:] - b[nch:]) / np.maximum(np.abs(a[nch:]), 1e-300)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = sys.argv[2] if len(sys.argv) > 2 else "24"
    base = ["--steps", steps, "--warmup", "0", "--no-cpu-baseline", "--checksum", "--no-at-rest"]
    S = [bench(base)["checksum"] for _ in range(reps)]
    Cc = [bench(base + ["--compact", "--rebin-check", "2"])["checksum"] for _ in range(reps)]
    n = 67_108_864
    pairs = [("slot-slot", S[0], S[1]), ("comp-comp", Cc[0], Cc[1])] + [("slot-comp%d" % k, S[k], Cc[k]) for k in range(reps)]
    print("%-8s" % "chan" + "".join(" %22s" % p[0] for p in pairs))
    rr = [ratios(p[1], p[2], n) for p in pairs]
    for k, nm in enumerate(NAMES):
        print("%-8s" % nm + "".join("  sum %8.2e sq %8.2e" % (r[0][k], r[1][k]) for r in rr))
    print("values (slotted run 0): sum / sumsq per channel")
    a = np.array(S[0])
    for k, nm in enumerate(NAMES):
        print("%-8s %14.6e %14.6e" % (nm, a[k], a[len(NAMES) + k]))


if __name__ == "__main__":
    main()
