import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zpc_amd as zs
from zpc_amd.containers import Bht
pol = zs.rocm_exec()
for n in (1 << 20, 1 << 22, 1 << 24):
    g = np.random.default_rng(12)
    keys = g.integers(0, 256, (n, 3), dtype=np.int32)
    packed = (keys[:, 0].astype(np.int64) << 16) | (keys[:, 1].astype(np.int64) << 8) | keys[:, 2]
    ndist = np.unique(packed).shape[0]
    tab = Bht(3, n)
    dk = torch.from_numpy(keys).cuda()
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert(pol, dk.data_ptr(), n, ret.data_ptr())
    pol.syncCtx()
    r = ret.cpu().numpy()
    v = tab.view()
    succ = np.empty(1, np.int32); C.CDLL("libamdhip64.so").hipMemcpy(succ.ctypes.data_as(C.c_void_p), C.c_void_p(v.success), C.c_size_t(4), 2)
    print(n, "distinct", ndist, "size", tab.size(), "ret>=0", (r >= 0).sum(), "ret==-1", (r == -1).sum(), "other", ((r < -1)).sum(), "success", succ[0], "tableSize", v.tableSize)
    q = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.query(pol, dk.data_ptr(), n, q.data_ptr()); pol.syncCtx()
    print("  queries missing:", (q.cpu().numpy() < 0).sum())
