import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, zpc_amd as zs
from zpc_amd.containers import LBvh
pol = zs.rocm_exec().sync(False)
g = torch.Generator(device="cuda"); g.manual_seed(7)
n, side = 10_000_000, 3163
uv = torch.rand(n, 2, device="cuda", generator=g)
ctr = torch.stack([uv[:, 0], uv[:, 1], 0.5 + 0.2 * torch.sin(6.28 * uv[:, 0]) * torch.cos(6.28 * uv[:, 1])], dim=1)
ext = (1.0 / side) * (0.5 + torch.rand(n, 3, device="cuda", generator=g))
bvs = torch.cat([ctr - ext, ctr + ext], dim=1).contiguous()
bvh = LBvh(); bvh.build(pol, bvs); pol.syncCtx()
counts = torch.zeros(n + 2, dtype=torch.int32, device="cuda")
zs.lib().zs_rocm_lbvh_self_query_count(pol.handle, bvh._h, counts.data_ptr()); pol.syncCtx(); torch.cuda.synchronize()
c = counts.cpu()
nw = (n + 63) // 64
print("waves %d  local steps/wave %.1f  far steps/wave %.1f  pairs %d" % (nw, c[n].item() / nw, c[n + 1].item() / nw, int(c[:n].sum())))
