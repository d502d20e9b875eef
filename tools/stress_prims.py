import numpy as np, torch, sys
sys.path.insert(0, '.')
import zpc_amd as zs
pol = zs.rocm_exec()
rng = np.random.default_rng(1)
bad = 0
for it in range(300):
    n = int(rng.integers(1, 400_000))
    a = rng.integers(-2**30, 2**30, n, dtype=np.int32)
    d = torch.from_numpy(a).cuda(); out = torch.empty_like(d)
    zs.radix_sort(pol, d, out)
    if not np.array_equal(out.cpu().numpy(), np.sort(a)): bad += 1; print('sort mismatch n', n, 'iter', it)
    zs.exclusive_scan(pol, d, out)
    if not np.array_equal(out.cpu().numpy(), (np.cumsum(a, dtype=np.int64) - a).astype(np.int32)): bad += 1; print('scan mismatch n', n, 'iter', it)
print('stress done, bad =', bad)
