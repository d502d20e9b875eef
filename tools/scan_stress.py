import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpc_amd as zs
pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
g = np.random.default_rng(3)
bad = 0
for it in range(300):
    n = int(g.integers(1, 3_000_000 if it % 3 else 5000))
    if it % 2:
        a = g.integers(-1000, 1000, n).astype(np.int32); d = torch.from_numpy(a).cuda(); o = torch.empty_like(d)
        zs.exclusive_scan(pol, d, o); ref = (np.cumsum(a, dtype=np.int64) - a).astype(np.int32)
    else:
        a = g.integers(-10**9, 10**9, n).astype(np.int64); d = torch.from_numpy(a).cuda(); o = torch.empty_like(d)
        zs.inclusive_scan(pol, d, o); ref = np.cumsum(a, dtype=np.int64)
    if not np.array_equal(o.cpu().numpy(), ref):
        bad += 1; print("MISMATCH", it, n, a.dtype)
print("scan mix:", bad, "mismatches")
sys.exit(1 if bad else 0)
