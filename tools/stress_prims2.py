"""onesweep fast path (full tiles of plain arrays) under misaligned base pointers, exact multiples of the tile and pair sorts"""
import numpy as np, torch, sys
sys.path.insert(0, '.')
import zpc_amd as zs
pol = zs.rocm_exec()
rng = np.random.default_rng(7)
bad = 0
for it in range(120):
    tiles = int(rng.integers(1, 40))
    n = tiles * 8192 + int(rng.choice([0, 0, 1, 8191, 4096, int(rng.integers(0, 8192))]))
    off = int(rng.integers(0, 4))
    a = rng.integers(-2**31, 2**31 - 1, n + off, dtype=np.int64).astype(np.int32)
    d = torch.from_numpy(a).cuda()[off:]
    out = torch.empty(n + off, dtype=torch.int32, device="cuda")[off:]
    zs.radix_sort(pol, d, out)
    if not np.array_equal(out.cpu().numpy(), np.sort(a[off:])): bad += 1; print('sort mismatch', n, off)
    v = torch.arange(n + off, dtype=torch.int32, device="cuda")[off:]
    vo = torch.empty(n + off, dtype=torch.int32, device="cuda")[off:]
    sb, eb = (0, 32) if it % 3 else (4, 20)
    zs.radix_sort_pair(pol, d, v, out, vo, sbit=sb, ebit=eb)
    key = ((a[off:].astype(np.int64) ^ (1 << 31 if True else 0)) & 0xffffffff)  # sign-flipped unsigned order
    dig = (key >> sb) & ((1 << (eb - sb)) - 1)
    order = np.argsort(dig, kind="stable")
    if not np.array_equal(vo.cpu().numpy(), (order + off).astype(np.int32)): bad += 1; print('pair mismatch', n, off, sb, eb)
print('stress2 done, bad =', bad)
