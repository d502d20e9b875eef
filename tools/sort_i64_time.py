import sys, torch
sys.path.insert(0, "/root/repo")
import zpc_amd as zs
pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(3)
def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for n in (100_000, 500_000, 1_000_000):
    a = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device="cuda", generator=g)
    v = torch.arange(n, dtype=torch.int32, device="cuda")
    o, vo = torch.empty_like(a), torch.empty_like(v)
    print("i64 n", n, "keys %.1f us pairs %.1f us" % (timeit(lambda: zs.radix_sort(pol, a, o)), timeit(lambda: zs.radix_sort_pair(pol, a, v, o, vo))))
