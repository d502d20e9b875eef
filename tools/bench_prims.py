#!/usr/bin/env python3
"""Secondary benchmarks (BASELINE.json configs 1, 2 and the sort stress of config 5): reduce / exclusive_scan /
radix_sort(_pair) throughput, TileVector AoSoA load/store, bht build -- each with its algorithmic bytes and the
achieved fraction of the 8 TB/s HBM roofline.  python tools/bench_prims.py [--json out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpc_amd as zs  # noqa: E402
from zpc_amd.containers import Bht  # noqa: E402

PEAK = 8000.0


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main(only=None, sizes=None, quiet=False, tv_cases=((16_000_000, 32, 25), (64_000_000, 64, 26))):
    """only: subset of {"prims", "tv", "bht", "lbvh"}; sizes: element counts of the primitive rows.  Also called by bench.py
    (secondary.prims of the driver line: the SURVEY 8(d) secondary metrics)."""
    pol = zs.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
    rows = []
    if only is None and "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1].split(",")  # prims,tv,bht,lbvh
    want = lambda k: only is None or k in only

    def add(name, n, unit_bytes, ms):
        gbs = unit_bytes * n / (ms * 1e-3) / 1e9
        rows.append({"name": name, "n": n, "ms": ms, "units_per_s": n / (ms * 1e-3), "algorithmic_bytes_per_unit": unit_bytes,
                     "GBps": gbs, "frac_of_8TBps": gbs / PEAK})
        if not quiet:
            print("%-44s n=%-10d %9.4f ms  %9.2f G/s  %8.1f GB/s  (%.1f%% of 8 TB/s)" % (name, n, ms, n / ms / 1e6, gbs, 100 * gbs / PEAK), flush=True)

    g = torch.Generator(device="cuda").manual_seed(1)
    if sizes is None:
        sizes = tuple(int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(",")) if "--sizes" in sys.argv else (1_000_000, 16_000_000, 64_000_000)
    for n in sizes if want("prims") else ():
        a = torch.randint(-2**30, 2**30, (n,), dtype=torch.int32, device="cuda", generator=g)
        out1 = torch.zeros(1, dtype=torch.int32, device="cuda")
        out = torch.empty_like(a)
        v = torch.arange(n, dtype=torch.int32, device="cuda")
        vo = torch.empty_like(v)
        # repetitions by size: a timed window of a few milliseconds at least (five 40-us sorts between two events measure the events' own
        # ramp as much as the sorts: 45 us against 38 us over 200 repetitions)
        reps = max(5, min(200, 200_000_000 // n))
        add("reduce<i32,plus>", n, 4, timeit(lambda: zs.reduce(pol, a, None, out1), reps=max(20, reps)))
        add("exclusive_scan<i32>", n, 8, timeit(lambda: zs.exclusive_scan(pol, a, out), reps=max(20, reps)))
        add("radix_sort<i32> (32 bit)", n, 32, timeit(lambda: zs.radix_sort(pol, a, out), reps=reps))
        add("radix_sort_pair<i32,i32> (32 bit)", n, 64, timeit(lambda: zs.radix_sort_pair(pol, a, v, out, vo), reps=reps))
        if n <= 16_000_000:
            import math
            passes = 1 + max(0, math.ceil(math.log2(max(1, n / 2048))))  # tile sort + global merge passes, 16 B/pair each
            def ms_pair():
                out.copy_(a); vo.copy_(v)
                zs.merge_sort_pair(pol, out, vo)
            def ms_copy_only():
                out.copy_(a); vo.copy_(v)
            t_all, t_copy = timeit(ms_pair, reps=5), timeit(ms_copy_only, reps=5)
            add("merge_sort_pair<i32,i32> (%d passes)" % passes, n, 16 * passes, t_all - t_copy)
        del a, out, v, vo
    # config 2: TileVector<f32,32>{m:1,x:3,v:3,F:9,C:9} load-all/store-all at 16M
    for n, L, Cn in tv_cases if want("tv") else ():
        tiles = (n + L - 1) // L
        tv = torch.rand(tiles * L * Cn, dtype=torch.float32, device="cuda", generator=g)
        add("TileVector<f32,%d> %d ch load+store" % (L, Cn), n, 8 * Cn, timeit(lambda: zs.lib().zs_rocm_tv_scale_f32(pol.handle, tv.data_ptr(), n, Cn, L, C.c_float(1.0001))))
        del tv
    # config 2: bht build, 16M random particles in [0,1)^3, dx = 1/256: cell keys (~10.6M distinct) and 8^3-block keys
    n = 16_000_000
    pos = torch.rand(n, 3, device="cuda", generator=g) if want("bht") else None
    for name, keys in (("cell keys", torch.floor(pos * 256).to(torch.int32)), ("8^3-block keys", torch.floor(pos * 32).to(torch.int32))) if want("bht") else ():
        keys = keys.contiguous()
        tab = Bht(3, n)

        def build():
            tab.reset(True)
            tab.insert(pol, keys.data_ptr(), n)
        ms = timeit(build, reps=3, warm=1)
        torch.cuda.synchronize()
        distinct = tab.size()
        add("bht<int,3,int,16> build, %s (%d distinct)" % (name, distinct), n, 12 + 32.0 * distinct / n, ms)
        del tab
    del pos
    # config 5: 10 M triangle AABBs of a jittered surface mesh in [0,1)^3 (extent ~ 2 cells of a 3163^2 sheet), LBvh<3,int,f32>
    from zpc_amd.containers import LBvh
    if not want("lbvh"):
        return rows
    n = 10_000_000
    side = 3163
    uv = torch.rand(n, 2, device="cuda", generator=g)
    ctr = torch.stack([uv[:, 0], uv[:, 1], 0.5 + 0.2 * torch.sin(6.28 * uv[:, 0]) * torch.cos(6.28 * uv[:, 1])], dim=1)
    ext = (1.0 / side) * (0.5 + torch.rand(n, 3, device="cuda", generator=g))
    bvs = torch.cat([ctr - ext, ctr + ext], dim=1).contiguous()
    bvh = LBvh()
    # traffic floor per primitive: boxes read by reduce + morton + leaf refit (72) + code/id written (8) + pair sort (64)
    # + node arrays written: (2n-1)/n * (24 box + 12 ints) + 4 leafInds (~76) + trunk temporaries ~6 ints r/w (48)
    add("LBvh build+refit, 10M AABBs", n, 268, timeit(lambda: bvh.build(pol, bvs), reps=3, warm=1))
    add("LBvh refit, 10M AABBs", n, 24 + 2 * 24 + 3 * 4 * 2, timeit(lambda: bvh.refit(pol, bvs), reps=3, warm=1))
    nq = 1_000_000
    counts = torch.zeros(nq, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: zs.lib().zs_rocm_lbvh_query_count(pol.handle, bvh.handle, bvs.data_ptr(), nq, counts.data_ptr()), reps=3, warm=1)
    hits = float(counts.double().mean().item())
    add("LBvh iter_neighbors count, 1M queries (%.1f hits/query)" % hits, nq, 24 + 4, ms)
    # config 5 proper: self-collision broadphase over all 10M leaves (count pass; the fill pass repeats the walk)
    sc = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: zs.lib().zs_rocm_lbvh_self_query_count(pol.handle, bvh.handle, sc.data_ptr()), reps=3, warm=1)
    npairs = float(sc.double().sum().item())
    add("LBvh self-collision broadphase count, 10M leaves (%.2f pairs/leaf)" % (npairs / n), n, 24 + 4, ms)
    # config 5 end to end: build + refit, count pass, exclusive scan, fill pass -> the (i, j) pair list
    def config5():
        bvh.build(pol, bvs)
        return bvh.self_query(pol)
    ms = timeit(config5, reps=3, warm=1)
    add("config 5: LBvh build + self-collision pair list, 10M boxes (%.0f M pairs)" % (npairs / 1e6), n, 268 + 2 * 28 + 8 * npairs / n, ms)
    del bvh, bvs
    return rows


if __name__ == "__main__":
    rows = main()
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
