"""sort_time.py [n]: HIP-event time of zs.radix_sort / radix_sort_pair alone (measurement builds via ZS_ROCM_LIB: nothing else runs, nothing checks the output)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import zpc_amd as zs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randint(-2**30, 2**30, (n,), dtype=torch.int32, device="cuda", generator=g)
out = torch.empty_like(a); v = torch.arange(n, dtype=torch.int32, device="cuda"); vo = torch.empty_like(v)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps): f()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps
print("n=%d keys %.4f ms  pairs %.4f ms" % (n, timeit(lambda: zs.radix_sort(pol, a, out)), timeit(lambda: zs.radix_sort_pair(pol, a, v, out, vo))))
