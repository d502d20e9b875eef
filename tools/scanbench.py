import sys, torch
sys.path.insert(0, "/root/repo")
import zpc_amd as zs
pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
for n in (16_000_000, 64_000_000, 256_000_000):
    a = torch.randint(-100, 100, (n,), dtype=torch.int32, device="cuda"); out = torch.empty_like(a)
    for _ in range(3): zs.exclusive_scan(pol, a, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): zs.exclusive_scan(pol, a, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ok = bool((out[1:1000] == torch.cumsum(a[:999], 0).int()).all())
    print("scan n=%d %.4f ms %.2f TB/s ok=%s" % (n, ms, 8 * n / ms / 1e9, ok))
