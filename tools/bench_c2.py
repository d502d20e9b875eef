"""P2C2G / G2C2P (gather-style transfers) at the config-3 size: 8 M particles (128^3 cells x 8) on a 256^3 grid, blocks of 8^3.
Prints one JSON line per pass with HIP-event times (per-kernel with ZS_C2_PROFILE=1 through the policy's profile switch)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import zpc_amd
from zpc_amd import lib
from zpc_amd.mpm import MpmTransfer

def main(cells=128, grid=256, side=8, model=0, iters=5, shuffle=False):
    dev = torch.device("cuda", 0)
    dx, dt = 1.0 / grid, 1e-4
    lo = [(grid - cells) // 2 // side * side] * 3
    hi = [l + cells for l in lo]
    pol = zpc_amd.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
    aos = bench.generate_particles(lo, hi, dx, 1234, dev, model)
    aos[:, 7:16] *= dx * dx * 0.25
    if shuffle:
        aos = aos[torch.randperm(aos.shape[0], device=dev)].contiguous()
    n = aos.shape[0]
    mt = MpmTransfer(pol, n, dx, dt, model=model, side=side, volume=dx ** 3 / 8, device=dev)
    lib().zs_rocm_tv_from_aos_f32(pol.handle, aos.data_ptr(), n, mt.nchn, mt.L, mt.buf.data_ptr())
    torch.cuda.synchronize()
    del aos
    mt.build_partition(n // 64)
    t0 = time.time(); mt.build_buckets(); torch.cuda.synchronize(); tb = time.time() - t0
    def timed(f):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        f(); torch.cuda.synchronize()
        ev[0].record()
        for _ in range(iters):
            f()
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / iters
    rows = {"n": n, "nblocks": mt.nblocks, "buckets_first_build_ms": tb * 1e3, "shuffled": shuffle}
    t0 = time.time()
    for _ in range(iters):
        mt.build_buckets()
    torch.cuda.synchronize()
    rows["buckets_rebuild_ms"] = (time.time() - t0) * 1e3 / iters   # wall clock: the build reads the bucket count back
    if "--hashed" not in sys.argv:
        mt.build_buckets(dense=True); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            mt.build_buckets(dense=True)
        torch.cuda.synchronize()
        rows["partition_buckets_rebuild_ms"] = (time.time() - t0) * 1e3 / iters   # P2C2G below then runs on these
    def p2c2g(kind):
        mt.clear_grid(); mt.p2c2g(kind)
    for kind, name in ((0, "p2c2g_ms"), (1, "p2c2g_momentum_ms"), (2, "p2c2g_force_ms")):
        rows[name] = timed(lambda: p2c2g(kind))
    mt.clear_grid(); mt.p2c2g(0); mt.grid_update((0.0, -9.8, 0.0))
    mt.params.dt = 0.0   # repeatable: positions and F stay put
    rows["g2c2p_ms"] = timed(mt.g2c2p)
    rows["g2c2p_three_calls_ms"] = timed(lambda: mt.g2c2p(fused=False))
    mt.params.dt = dt
    # the scatter transfers on the same particles, particle order (the reference algorithm) for comparison
    def p2g():
        mt.clear_grid(); mt.p2g(binned=False)
    rows["p2g_particle_order_ms"] = timed(p2g)
    print(json.dumps(rows))

if __name__ == "__main__":
    main(shuffle=False)
    if "--lattice-only" not in sys.argv:
        main(shuffle=True)
