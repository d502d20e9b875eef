#!/usr/bin/env python3
"""Two streams sorting concurrently, both in the small path's in-launch LSD mode (narrow range plus near outliers, 2 M keys = 245 tiles
each: together more workgroups than the device has CUs), next to a third stream that keeps the CUs busy.  A scheme that needs all of a
sort's workgroups resident at once can hang here; tickets cannot.   python tools/sort_two_streams.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpc_amd as zs  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 2_000_000
g = torch.Generator(device="cuda").manual_seed(5)
streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
pols = [zs.rocm_exec().sync(False).external_stream(s.cuda_stream) for s in streams[:2]]
keys, outs, refs = [], [], []
for i in range(2):
    a = torch.randint(0, 70000, (n,), dtype=torch.int32, device="cuda", generator=g)
    a[n // 3], a[n - 1] = 70000 * 5, 70000 * 3
    keys.append(a)
    outs.append(torch.empty_like(a))
    refs.append(torch.sort(a)[0])
busy = torch.randn(4096, 4096, device="cuda")
torch.cuda.synchronize()
bad = 0
for r in range(rounds):
    with torch.cuda.stream(streams[2]):
        for _ in range(4):
            busy = torch.tanh(busy) * 1.0001
    for i in range(2):
        zs.radix_sort(pols[i], keys[i], outs[i])
    if r % 10 == 9:
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(outs[i], refs[i]) else 1 for i in range(2))
torch.cuda.synchronize()
bad += sum(0 if torch.equal(outs[i], refs[i]) else 1 for i in range(2))
assert zs.lib().zs_rocm_last_error(0) == 0
print("two-stream sort: %d rounds, %d mismatches" % (rounds, bad))
sys.exit(1 if bad else 0)
