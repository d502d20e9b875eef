"""config 5 pieces timed one by one: count pass, scan, fill pass; distribution of the per-leaf hit counts"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zpc_amd as zs
from zpc_amd.containers import LBvh
from zpc_amd.primitives import exclusive_scan
pol = zs.rocm_exec().sync(False)
g = torch.Generator(device="cuda"); g.manual_seed(7)
n, side = 10_000_000, 3163
uv = torch.rand(n, 2, device="cuda", generator=g)
ctr = torch.stack([uv[:, 0], uv[:, 1], 0.5 + 0.2 * torch.sin(6.28 * uv[:, 0]) * torch.cos(6.28 * uv[:, 1])], dim=1)
ext = (1.0 / side) * (0.5 + torch.rand(n, 3, device="cuda", generator=g))
bvs = torch.cat([ctr - ext, ctr + ext], dim=1).contiguous()
bvh = LBvh(); bvh.build(pol, bvs); pol.syncCtx()
counts = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
def t(f, reps=3):
    f(); torch.cuda.synchronize(); pol.syncCtx()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    pol.syncCtx(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
L = zs.lib()
print("count ms", t(lambda: L.zs_rocm_lbvh_self_query_count(pol.handle, bvh._h, counts.data_ptr())))
c = counts[:n]
print("mean %.2f max %d  >16: %.4f %%  >32: %.4f %%  waves (64 leaves) with a leaf >16: %.2f %%" % (c.float().mean().item(), c.max().item(), (c > 16).float().mean().item() * 100,
      (c > 32).float().mean().item() * 100, (c[: n // 64 * 64].view(-1, 64) > 16).any(1).float().mean().item() * 100))
offsets = torch.empty_like(counts)
print("scan ms", t(lambda: exclusive_scan(pol, counts, offsets)))
total = int(offsets[n].item())
pairs = torch.empty(total * 2, dtype=torch.int32, device="cuda")
print("fill ms", t(lambda: L.zs_rocm_lbvh_self_query_fill(pol.handle, bvh._h, offsets.data_ptr(), pairs.data_ptr())), "pairs", total)
