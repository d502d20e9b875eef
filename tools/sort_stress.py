#!/usr/bin/env python3
"""Randomised stress of radix_sort / radix_sort_pair (sizes 1 .. 5 M, ragged tiles, narrow and wide key ranges, bit windows) against
torch.sort(stable=True); meant to shake out ordering / look-back races that the fixed-size tests would not hit.   python tools/sort_stress.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpc_amd as zs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(7)
bad = 0
for it in range(iters):
    n = int(torch.randint(1, 5_000_000 if it % 4 else 40_000, (1,)).item())
    span = [2, 255, 70_000, 2 ** 30][it % 4]
    a = torch.randint(-span, span, (n,), dtype=torch.int32, device="cuda", generator=g)
    v = torch.arange(n, dtype=torch.int32, device="cuda")
    out, vo = torch.empty_like(a), torch.empty_like(v)
    zs.radix_sort_pair(pol, a, v, out, vo)
    ref, idx = torch.sort(a, stable=True)
    ok = torch.equal(out, ref) and torch.equal(vo, idx.to(torch.int32))
    out2 = torch.empty_like(a)
    zs.radix_sort(pol, a, out2)
    ok = ok and torch.equal(out2, ref)
    if not ok:
        bad += 1
        print("MISMATCH at iteration", it, "n", n, "span", span)
assert zs.lib().zs_rocm_last_error(0) == 0
print("sort stress:", iters, "iterations,", bad, "mismatches")
sys.exit(1 if bad else 0)
