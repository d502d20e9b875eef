#!/usr/bin/env python3
"""Randomised stress of the small-input radix sort path (primitives.hip, "split + finish"): sizes 1 .. 2.1 M, eleven key distributions,
random bit windows, keys and pairs, against torch.sort(stable=True).   python tools/sort_small_stress.py [iterations] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpc_amd as zs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(seed)
cg = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=cg).item())


def keys(kind, n):
    full = lambda: torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    if kind == 0:
        return full()
    if kind == 1:   # a range of random width at a random offset
        w = 1 << ri(1, 32)
        lo = ri(-2 ** 31, 2 ** 31 - w)
        return (torch.randint(0, w, (n,), dtype=torch.int64, device="cuda", generator=g) + lo).to(torch.int32)
    if kind == 2:   # a sentinel in a random share of the keys
        a = full()
        a[torch.rand(n, device="cuda", generator=g) < ri(1, 90) / 100.0] = [2 ** 31 - 1, -2 ** 31, 0, -1][ri(0, 4)]
        return a
    if kind == 3:
        return torch.sort(full())[0]
    if kind == 4:
        return torch.sort(full(), descending=True)[0]
    if kind == 5:   # few distinct values spread over the range
        m = ri(1, 40)
        vals = torch.randint(-2 ** 31, 2 ** 31 - 1, (m,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
        return vals[torch.randint(0, m, (n,), device="cuda", generator=g)]
    if kind == 6:   # narrow range plus a few outliers at random distances
        a = torch.randint(0, 1 << ri(1, 20), (n,), dtype=torch.int32, device="cuda", generator=g)
        for _ in range(ri(1, 4)):
            a[ri(0, n)] = ri(0, 2 ** 31 - 1)
        return a
    if kind == 7:   # sorted blocks: tiles see different ranges
        return torch.sort(full().view(-1)[: n // 8 * 8].view(8, -1), dim=1)[0].reshape(-1) if n >= 8 else full()
    if kind == 8:
        return torch.full((n,), ri(-2 ** 31, 2 ** 31 - 1), dtype=torch.int32, device="cuda")
    if kind == 9:   # two clusters far apart
        a = torch.randint(0, 1 << ri(1, 16), (n,), dtype=torch.int32, device="cuda", generator=g)
        a[torch.rand(n, device="cuda", generator=g) < 0.5] += ri(1 << 20, 1 << 30)
        return a
    return torch.randint(-2 ** 30, 2 ** 30, (n,), dtype=torch.int32, device="cuda", generator=g)


bad = 0
for it in range(iters):
    n = ri(1, 2_100_000) if it % 5 else ri(1, 20_000)
    kind = ri(0, 11)
    a = keys(kind, n)
    n = a.numel()
    sb = ri(0, 24) if it % 3 == 0 else 0
    eb = ri(sb + 1, 33) if it % 3 == 0 else 32
    v = torch.arange(n, dtype=torch.int32, device="cuda")
    ko, vo, k1 = torch.empty_like(a), torch.empty_like(v), torch.empty_like(a)
    zs.radix_sort_pair(pol, a, v, ko, vo, sbit=sb, ebit=eb)
    zs.radix_sort(pol, a, k1, sbit=sb, ebit=eb)
    w = (a.to(torch.int64) + 2 ** 31) >> sb & ((1 << (eb - sb)) - 1)
    _, idx = torch.sort(w, stable=True)
    if not (torch.equal(vo, idx.to(torch.int32)) and torch.equal(ko, a[idx]) and torch.equal(k1, a[idx])):
        bad += 1
        print("MISMATCH iteration", it, "n", n, "kind", kind, "window", sb, eb, flush=True)
assert zs.lib().zs_rocm_last_error(0) == 0
print("small sort stress: %d iterations (seed %d), %d mismatches" % (iters, seed, bad))
sys.exit(1 if bad else 0)
