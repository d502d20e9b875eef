import json, subprocess, sys, os, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_dist_gpu import _run
args = ["--cells", "24,48,24", "--steps", "9", "--warmup", "0", "--no-cpu-baseline", "--checksum", "--drift", "0,7,0", "--migrate-every", "3"]
ref = _run(1, args); out = _run(int(sys.argv[1]), args)
a, b = np.array(ref["checksum"]), np.array(out["checksum"])
nch = len(a)//2
scale = np.sqrt(ref["config"]["particles"] * np.maximum(a[nch:], 1e-30))
r1 = np.abs(a[:nch]-b[:nch])/(scale+1e-30); r2 = np.abs(a[nch:]-b[nch:])/(np.abs(a[nch:])+1e-30)
np.set_printoptions(linewidth=200, precision=3)
print("sum ratio", r1); print("sq ratio", r2); print(out["config"]["migrated_rank0"], out["config"]["particles"], ref["config"]["particles"])
