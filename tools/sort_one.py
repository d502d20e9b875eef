import sys, os, torch
sys.path.insert(0, "/root/repo")
import zpc_amd as zs
pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
a = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
if kind == "c1":   # BASELINE config 1: ints in [-2^30, 2^30)
    a = torch.randint(-2**30, 2**30, (n,), dtype=torch.int32, device="cuda", generator=g)
if kind == "sentinel":
    a[torch.rand(n, device="cuda", generator=g) < 0.2] = 2**31 - 1
out = torch.empty_like(a)
for _ in range(200):
    zs.radix_sort(pol, a, out)
torch.cuda.synchronize()
