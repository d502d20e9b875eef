import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import zpc_amd as zs
from zpc_amd.mpm import MpmTransfer
from util import make_cloud
pol = zs.rocm_exec()
dx, dt = 1.0/64, 1e-3
mass, pos, vel, Cm, F = make_cloud(12, dx, 2, seed=3, vel_scale=0.05)
n = pos.shape[0]
vel += np.array([2.4, -3.2, 1.6], np.float32)   # 0.15, 0.2, 0.1 cell/step
mt = MpmTransfer(pol, n, dx, dt, model=1, side=8, volume=dx**3/8, cache_stress=True)
mt.upload(mass, pos, vel, Cm, F, np.zeros(n, np.float32))
mt.build_partition(n, margin=1)
mt.rebin(); mt.update_stress(); mt.clear_grid(); mt.p2g(); mt.grid_update((0, 0, 0))
mt.slot(K=int(sys.argv[1]) if len(sys.argv) > 1 else 16, outbox_cap=512)
for s in range(12):
    mt.g2p2g(); mt.grid_update((0, 0, 0)); pol.syncCtx()
    m = mt.cell_mask.cpu().numpy().view(np.uint32)
    pc = np.array([bin(int(x)).count("1") for x in m[m != 0]])
    st = mt.slot_status.cpu().numpy()
    print(s, "particles", pc.sum(), "of", n, "max/cell", pc.max(), "status", st[:7], "highest bit", int(np.log2(m.max())))
