#!/usr/bin/env python3
"""Small-input radix sort (split + finish, primitives.hip): correctness over sizes x key distributions x bit windows against torch.sort(stable),
then timings.  ZS_ROCM_SORT_SMALL=0 in the environment times the ordinary LSD passes at the same sizes.   python tools/sort_small_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpc_amd as zs  # noqa: E402

pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(11)


def keys(kind, n):
    if kind == "uniform":
        return torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    if kind == "c1":          # BASELINE config 1: ints in [-2^30, 2^30) -- half of the top window's digits occur
        return torch.randint(-2 ** 30, 2 ** 30, (n,), dtype=torch.int32, device="cuda", generator=g)
    if kind == "narrow":      # one top-digit bucket -> the fallback passes
        return torch.randint(0, 70000, (n,), dtype=torch.int32, device="cuda", generator=g)
    if kind == "sentinel":    # 20 % of the keys are INT_MAX
        a = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
        a[torch.rand(n, device="cuda", generator=g) < 0.2] = 2 ** 31 - 1
        return a
    if kind == "equal":
        return torch.full((n,), -7, dtype=torch.int32, device="cuda")
    if kind == "dups":        # 256 distinct values spread over the whole range
        return (torch.randint(-128, 128, (n,), dtype=torch.int32, device="cuda", generator=g) * (1 << 24)) + 5
    if kind == "morton":      # 30-bit codes under the default 32-bit window
        return torch.randint(0, 2 ** 30, (n,), dtype=torch.int32, device="cuda", generator=g)
    if kind == "sorted":
        return torch.sort(torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32))[0]
    if kind == "outlier":     # a narrow range and a few keys just above it (tiles disagree about the top bit)
        a = torch.randint(0, 70000, (n,), dtype=torch.int32, device="cuda", generator=g)
        a[n // 3] = 70000 * 5
        a[n - 1] = 70000 * 3
        return a
    if kind == "far":         # ... and one far above it
        a = torch.randint(0, 70000, (n,), dtype=torch.int32, device="cuda", generator=g)
        a[n // 2] = 2 ** 30 + 12345
        return a
    if kind == "two":
        return torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda", generator=g) * 1000 - 3
    raise ValueError(kind)


bad = 0
for n in (1, 63, 511, 8192, 8193, 100_000, 777_777, 1_000_000, 1_572_864, 2_048_000, 2_048_001):
    for kind in ("uniform", "c1", "narrow", "sentinel", "equal", "dups", "morton", "sorted", "outlier", "far", "two"):
        for (sb, eb) in ((0, 32), (4, 27), (0, 9), (16, 32)):
            a = keys(kind, n)
            v = torch.arange(n, dtype=torch.int32, device="cuda")
            ko, vo, k1 = torch.empty_like(a), torch.empty_like(v), torch.empty_like(a)
            zs.radix_sort_pair(pol, a, v, ko, vo, sbit=sb, ebit=eb)
            zs.radix_sort(pol, a, k1, sbit=sb, ebit=eb) if (sb, eb) != (0, 32) else zs.radix_sort(pol, a, k1)
            w = (a.to(torch.int64) + 2 ** 31) >> sb & ((1 << (eb - sb)) - 1)   # the window of the flipped key
            _, idx = torch.sort(w, stable=True)
            ok = torch.equal(vo, idx.to(torch.int32)) and torch.equal(ko, a[idx]) and torch.equal(k1, a[idx])
            if not ok:
                bad += 1
                print("MISMATCH n", n, kind, sb, eb)
# in place (output aliasing the input)
a = keys("uniform", 1_000_000)
ref, _ = torch.sort(a, stable=True)
zs.radix_sort(pol, a, a)
bad += 0 if torch.equal(a, ref) else 1
assert zs.lib().zs_rocm_last_error(0) == 0
print("small sort check:", bad, "mismatches")


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000.0


print("mode ZS_ROCM_SORT_SMALL=%s; us per sort: keys / pairs" % os.environ.get("ZS_ROCM_SORT_SMALL", "1"))
for n in (10_000, 100_000, 250_000, 500_000, 1_000_000, 1_500_000, 2_000_000):
    row = []
    for kind in ("uniform", "c1", "narrow", "morton", "sorted", "sentinel", "dups", "outlier"):
        a = keys(kind, n)
        v = torch.arange(n, dtype=torch.int32, device="cuda")
        ko, vo = torch.empty_like(a), torch.empty_like(v)
        row.append("%s %.0f/%.0f" % (kind, timeit(lambda: zs.radix_sort(pol, a, ko)), timeit(lambda: zs.radix_sort_pair(pol, a, v, ko, vo))))
    print("n %8d   " % n + "   ".join(row))
sys.exit(1 if bad else 0)
