#!/usr/bin/env python3
"""bench.py -- headline benchmark of the zpc hot path on MI355X.

Metric (BASELINE.json): particle*steps/s of the MPM particle<->grid transfer step (grid reset + P2G + grid
update + G2P) on the 64M-particle sand column (DruckerPrager, 8 particles/cell, dx = 1/512, 512^3 sparse grid),
on 1/2/4/8 GPUs of one node (strong scaling: the 64M particles are split spatially, ghost grid blocks exchanged
over RCCL/xGMI).  Inputs are generated on the device and are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel (binned P2G) algorithmic bytes / launch duration (HIP events on the launch stream)
  "cpu_baseline": the CPU oracle's OpenMP port of the reference P2G+G2P timed on this box's host cores on a
                  bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_ISSUE_PEAK_CYCLES = 2.2     # shader cycles per VALU instruction a SIMD reaches on independent v_fma_f32 with >= 2 resident waves (tools/valu_issue_bench.hip, r05)
# counted floor of the fused step's wave-level VALU instructions per 64 particles, by model index (0 FixedCorotated, 1 DruckerPrager): gather 290 +
# advection / F update / SVD / model (700 | 860) + Q-form staging 150 + P2G accumulate of the four channel sets at full lane occupancy 628
VALU_FLOOR_PER_64 = {0: 290.0 + 700.0 + 150.0 + 628.0, 1: 290.0 + 860.0 + 150.0 + 628.0}
VALU_ISSUE_MIX_CYCLES = 2.9      # the same for the fused step's own SVD + return-mapping instruction mix at 4 waves per SIMD (tools/svd_issue_bench.hip, r05)
P2G_BYTES = {0: 107.0, 1: 115.0}  # algorithmic B/particle: 100 B particle read + 7 B grid (+8 B logJp r/w)
G2P_BYTES = 145.5
YIELD_SURFACE = 0.816496580927726 * 2.0 * 0.5 / (3.0 - 0.5)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=512, help="cells per unit length: dx = 1/grid (512: config 4; 256: config 3)")
    ap.add_argument("--cells", type=str, default="128,512,128",
                    help="sand column extent in cells, 8 particles each (default 128x512x128 = 64 Mi particles; splits evenly on block planes for 2/4/8 ranks)")
    ap.add_argument("--lift", type=int, default=0,
                    help="raise the column by this many cells (rounded to whole blocks).  Default 0: it stands on y = 0 as in BASELINE config 4. "
                         "Next to the coordinate origin the reference's arena weights a particle one cell off when its local position rounds to "
                         "exactly 1.5 (profiles/r03_compact_outliers.md): one foot particle in ~3 %% of the compact-storage runs; comparisons of two runs lift the column")
    ap.add_argument("--model", type=str, default="sand", choices=["sand", "jello"])
    ap.add_argument("--side", type=int, default=8, choices=[4, 8],
                    help="grid block side: 8 = SparseGrid<3,f32,8> blocks (default, the '512^3 sparse grid' of BASELINE.json), 4 = Grids<f32,3,4>")
    ap.add_argument("--lane-width", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--unbinned", action="store_true", help="particle-order path (reference algorithm) instead of the binned path")
    ap.add_argument("--no-cache-stress", action="store_true",
                    help="evaluate the constitutive model inside P2G (reference order) instead of in the tail of the previous G2P")
    ap.add_argument("--unfused", action="store_true",
                    help="separate P2G and G2P kernels per step instead of the fused G2P2G pass (G2P of step n + P2G of step n+1 in "
                         "one kernel; v, C, stress stay on chip)")
    ap.add_argument("--migrate-every", type=int, default=0,
                    help="every K steps: move particles to the rank that owns their cell, rebuild partition / halo lists / bins "
                         "(0 = never; the default bench window moves particles < 0.1 cell)")
    ap.add_argument("--drift", type=str, default="0,-1.0,0",
                    help="uniform velocity added to every particle (m/s).  Default 0,-1,0: the column falls at 1 m/s = 0.051 cell per "
                         "step (dx = 1/512, dt = 1e-4) -- the operating point: 5 %% of the particles change cell every step")
    ap.add_argument("--rebin-check", type=int, default=4,
                    help="fused step: every K..8K steps look at the step times since the last re-bin; once the time lost to "
                         "particles that left their cell (sum of step time - best step time) exceeds the cost of a re-bin, the "
                         "particles are re-binned (local, no communication; only the step's input channels move).  0 = never")
    ap.add_argument("--rebin-at", type=str, default="",
                    help="compact storage: re-bin after exactly these steps (comma list, counted from 1 over warm-up + timed steps) instead of "
                         "letting the timing-driven controller decide -- a reproducible schedule for comparisons")
    ap.add_argument("--floor", action="store_true",
                    help="apply a Separate plane collider 1.5 cells above y = 0 after every grid update "
                         "(ApplyBoundaryConditionOnGridBlocks; off in the headline configuration)")
    ap.add_argument("--decomp", type=str, default="",
                    help="rank grid AxBxC (default: 2x2x2 for 8 ranks, slabs along y, 1xNx1, otherwise)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1, fused step: exchange the ghost blocks after the whole transfer kernel instead of overlapping "
                         "it with the interior blocks")
    ap.add_argument("--comm", type=str, default="native", choices=["native", "torch"],
                    help="native: halo exchange, CFL allreduce and migration through libzsrocm.so's zs_rocm_dist_* (RCCL from C++); "
                         "torch: the same steps through torch.distributed (always used with --backend gloo)")
    ap.add_argument("--no-cfl", action="store_true", help="skip the per-step maxVelSqr allreduce(max) (CFL time-step control)")
    ap.add_argument("--backend", type=str, default="nccl", choices=["nccl", "gloo"],
                    help="gloo: halo buffers staged through host memory -- lets N ranks share ONE GPU to validate the multi-rank path")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (validation only)")
    ap.add_argument("--slot-stats", action="store_true", help="report the occupancy statistics of the slotted storage after the run")
    ap.add_argument("--checksum", action="store_true", help="print global sums of particle state after the run (N-rank vs 1-rank check)")
    ap.add_argument("--halo-channels", type=int, default=4, choices=[4, 7],
                    help="grid channels of a ghost block that travel between ranks: 4 = {m, mv}, all a step reads of a ghost block (grid update: "
                         "v = mv / m + g dt; G2P gathers v -- GridOp.hpp:90-104); 7 = the rhs channels too")
    ap.add_argument("--range-schedule", type=str, default="auto", choices=["auto", "in-turn", "side-by-side", "one-launch"],
                    help="multi-GPU step (zs_rocm_mpm_step.rangeSchedule): in-turn = boundary blocks' launch, then the interior launch, on the compute "
                         "stream; side-by-side = the boundary launch on the exchange stream next to the interior launch; one-launch = one launch "
                         "over all blocks whose boundary workgroups count themselves off, a gate kernel on the exchange stream waits for the "
                         "count (8^3 blocks).  auto: one-launch for 8^3 blocks, in-turn for 4^3")
    ap.add_argument("--check-handover", action="store_true",
                    help="(tests, with --rank-proxy) after every step compare what the exchange stream saw of the shared blocks' mass sums when the exchange "
                         "started (zs_rocm_mpm_step.handoverSnapshot) with the same blocks after the step: config.handover = steps checked / mismatched")
    ap.add_argument("--understate-boundary", type=float, default=0.0,
                    help="(negative control of --check-handover) tell the step that only this share of the boundary blocks is boundary: the exchange "
                         "starts before the rest is done and the check must see it")
    ap.add_argument("--tag-mass", action="store_true",
                    help="every particle's mass carries its number in the global box (tests: per-particle comparison of runs on different rank counts)")
    ap.add_argument("--dump-state", type=str, default="",
                    help="after the run every rank writes its particles (m, x, v, C, F, logJp) to <prefix>.rank<r>.npz")
    ap.add_argument("--block-order", type=str, default="holders_lex", choices=["insertion", "holders_lex", "lex", "morton"],
                    help="numbering of the partition's blocks (MpmTransfer.build_partition): holders_lex = blocks with particles in "
                         "lexicographic key order, apron blocks behind them; insertion = the hash table's race (the reference's)")
    ap.add_argument("--block-axes", type=str, default="auto",
                    help="holders_lex / lex: the block key's components from most to least significant, e.g. 0,2,1 (the last one changes fastest "
                         "along the numbering).  auto: slotted storage -- the column's longest axis last (fused block kernel: -1.3 %%); else 0,1,2")
    ap.add_argument("--compact", action="store_true",
                    help="compact round-robin particle order + re-bin controller (round 1's storage) instead of the slotted storage "
                         "the fused step keeps valid by itself (zpc_amd/csrc/mpm_slotted.hip)")
    ap.add_argument("--slot-rounds", type=int, default=24, help="slotted storage: rounds (particle capacity) per cell, <= 32")
    ap.add_argument("--outbox-cap", type=int, default=128, help="slotted storage: movers one 4^3-cell bin can send per step")
    ap.add_argument("--margin", type=int, default=1,
                    help="slotted storage: extra layers of grid blocks around the occupied ones (room to travel before a re-partition)")
    ap.add_argument("--repartition", type=str, default="closed", choices=["closed", "open"],
                    help="slotted storage, when to re-partition: closed = when the step's own status words ask for it (a particle lives in a "
                         "block next to the partition's edge; polled asynchronously every side / 4 steps); open = r03's schedule derived "
                         "from drift + gravity alone (kept to show what it misses: flags are logged, not fatal)")
    ap.add_argument("--py-step", action="store_true",
                    help="slotted storage: enqueue the step's kernels / exchange from Python call by call (r03) instead of through the ONE "
                         "C-ABI call zs_rocm_mpm_step_slotted")
    ap.add_argument("--rank-proxy", type=int, default=0,
                    help="N > 0 on ONE GPU: run this rank's share of an N-rank job with the N-rank schedule's launches -- boundary range, "
                         "interior range, the ghost-block exchange over RCCL (world 1, the rank itself as its only peer, into a scratch grid) "
                         "on the side stream, CFL allreduce -- to measure the fixed per-step host / launch / exchange cost of a rank")
    ap.add_argument("--no-at-rest", action="store_true",
                    help="skip the secondary measurements (short sub-runs of this script at rest: slotted, compact, unfused stand-alone P2G / G2P; "
                         "and the primitives / bht / TileVector rows of SURVEY 8(d))")
    return ap.parse_args()


def _hash_uniform(gid, stream):
    """Counter-based uniform [0,1) from the GLOBAL particle id (splitmix64-style integer hash, computed in int64 with
    wrap-around): every decomposition of the domain generates bit-identical particles."""
    def s64(v):  # python int -> the int64 with the same low 64 bits
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    x = gid * s64(0x9E3779B97F4A7C15) + s64((stream + 1) * 0x632BE59BD9B4E019)  # int64 wrap-around is intended
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * s64(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * s64(0x94D049BB133111EB)
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    return ((x >> 11) & ((1 << 53) - 1)).double() * (1.0 / (1 << 53))


def _hash_normal(gid, stream):
    u1 = _hash_uniform(gid, 2 * stream).clamp_min(1e-12)
    u2 = _hash_uniform(gid, 2 * stream + 1)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * 3.141592653589793 * u2)


def generate_particles(box_lo, box_hi, dx, seed, device, model, tag_box=None):
    """8 particles per cell on a jittered 2x2x2 sub-lattice, generated on the device in chunks.
    Returns AoS [n, C] float32: m, x3, v3, C9, F9, (logJp).  Values depend only on the global particle id."""
    ext = [box_hi[d] - box_lo[d] for d in range(3)]
    ncell = ext[0] * ext[1] * ext[2]
    n = ncell * 8
    nch = 26 if model == 1 else 25
    aos = torch.empty(n, nch, dtype=torch.float32, device=device)
    h = dx / 2
    chunk = 1 << 21
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        pid = torch.arange(s, e, device=device, dtype=torch.int64)
        cell, sub = pid // 8, pid % 8
        cx = cell // (ext[1] * ext[2]) + box_lo[0]
        cy = (cell // ext[2]) % ext[1] + box_lo[1]
        cz = cell % ext[2] + box_lo[2]
        gid = (((cx + 4096) * 8192 + (cy + 4096)) * 8192 + (cz + 4096)) * 8 + sub + seed * 7919
        sx, sy, sz = sub // 4, (sub // 2) % 2, sub % 2
        aos[s:e, 1] = ((cx * 2 + sx).double() + 0.5 + (_hash_uniform(gid, 0) - 0.5) * 0.8).float() * h
        aos[s:e, 2] = ((cy * 2 + sy).double() + 0.5 + (_hash_uniform(gid, 1) - 0.5) * 0.8).float() * h
        aos[s:e, 3] = ((cz * 2 + sz).double() + 0.5 + (_hash_uniform(gid, 2) - 0.5) * 0.8).float() * h
        aos[s:e, 0] = 1000.0 * dx ** 3 / 8
        if tag_box is not None:  # --tag-mass: the mass carries the particle's number in the GLOBAL box (exact for < 2^22 particles)
            tlo, text = tag_box
            tid = (((cx - tlo[0]) * text[1] + (cy - tlo[1])) * text[2] + (cz - tlo[2])) * 8 + sub
            aos[s:e, 0] = ((1000.0 * dx ** 3 / 8) * (1.0 + tid.double() * 2.0 ** -22)).float()
        for k in range(3):
            aos[s:e, 4 + k] = (0.05 * _hash_normal(gid, 3 + k)).float()
        for k in range(9):
            aos[s:e, 7 + k] = (0.1 * _hash_normal(gid, 6 + k)).float()
            aos[s:e, 16 + k] = (0.01 * _hash_normal(gid, 15 + k) + (1.0 if k % 4 == 0 else 0.0)).float()
        if model == 1:
            aos[s:e, 25] = 0.0
    return aos


def _same_code(j):
    """a PMC traffic figure belongs to this run only if the kernels it was collected on still have the same machine code: the json
    names the object file and a kernel-name regex and carries tools/kernel_hash.py's fingerprint of them (collected by the refresh
    script on the GPU box from the same build)"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_hash
        want = j.get("code_hash")
        return bool(want) and kernel_hash.combined(os.path.join(ROOT, j["code_object"]), j["code_regex"]) == want
    except Exception:
        return False


def cpu_baseline(sample, dx, dt, model, side, vol):
    """OpenMP port of the reference's OmpExecutionPolicy P2G+G2P (oracle/mpm.c) on a bounded sample of the workload."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libzpc_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libzpc_oracle.so"])
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from orc import OracleMpm  # the checker's own wrapper (oracle/orc.py): the cpu_baseline leg is its only use here
    o = C.CDLL(so)
    cores = os.cpu_count() or 1
    nth = max(1, cores - 1)  # omp_exec() = hardware_concurrency() - 1 (omp/execution/ExecutionPolicy.hpp:1192-1194)
    ncell = max(1, sample // 8)
    ey = max(1, ncell // (40 * 40))
    aos = generate_particles((100, 0, 100), (140, ey, 140), dx, 7, "cpu", model).numpy()
    n = aos.shape[0]
    mass, pos, vel = aos[:, 0].copy(), aos[:, 1:4].copy(), aos[:, 4:7].copy()
    Cm, F = aos[:, 7:16].copy(), aos[:, 16:25].copy()
    lj = np.zeros(n, np.float32)
    def steps_per_second(threads, budget_s, max_reps):
        """median over the timed steps after ONE discarded warm-up step; (particle*steps/s, timed steps)"""
        om = OracleMpm(o, model, dx, dt, side, vol, nthreads=threads)
        om.build_partition(pos, max(1024, n // 64))
        p, v, c, f = pos.copy(), vel.copy(), Cm.copy(), F.copy()
        times, spent = [], 0.0
        for rep in range(max_reps + 1):
            om.grid[:] = 0
            t0 = time.perf_counter()
            om.p2g(mass, p, v, c, f, lj)
            om.grid_update((0.0, -9.8, 0.0))
            om.g2p(p, v, c, f)
            dt_s = time.perf_counter() - t0
            if rep:  # (step 0: thread pool start-up, first touch of the grid)
                times.append(dt_s)
                spent += dt_s
            if rep >= 5 and spent >= budget_s:
                break
        return n / float(np.median(times)), len(times)
    val, reps = steps_per_second(nth, 8.0, 20)
    # the port's P2G is a CAS loop on shared grid nodes and stops scaling long before the box's core count: the same
    # sample at a few smaller thread counts, the best of them beside the reference's own default (hardware_concurrency() - 1)
    best_val, best_thr = val, nth
    for thr in (8, 16, 32, 64, 128):
        if thr < nth:
            v2, _ = steps_per_second(thr, 1.5, 5)
            if v2 > best_val:
                best_val, best_thr = v2, thr
    return {"value": val, "unit": "particle*steps/s", "cores": cores, "threads": nth, "kind": "port",
            "best_threads": best_thr, "best_threads_value": best_val,
            "sample": "%d particles of the same sand column, median of %d steps after one discarded warm-up step (P2G + grid update + G2P), "
                      "oracle/mpm.c OpenMP port; best_threads: the same sample at 8 / 16 / 32 / 64 / 128 threads (5 steps each)" % (n, reps)}


def main():
    a = parse()
    a.fused = not (a.unfused or a.unbinned or a.no_cache_stress)
    a.slotted = a.fused and not a.compact and a.lane_width == 64
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU of this node, rendezvous on the loopback address
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if a.same_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        # fail fast and loudly: a rank without a GPU of its own would share device 0 and RCCL would hang or refuse later
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) are visible -- run --gpus N on a node with N GPUs (or --same-device for validation)"
                         % (int(os.environ.get("RANK", "0")), local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    comm_dev = device if a.backend == "nccl" else torch.device("cpu")
    assert world == a.gpus, "launch with --nproc-per-node == --gpus"

    import zpc_amd
    from zpc_amd import lib
    from zpc_amd.mpm import MpmTransfer
    from zpc_amd.dist import cell_box, HaloExchange, NativeComm
    if a.decomp and world > 1:
        zpc_amd.dist.set_split_dims(world, [int(v) for v in a.decomp.split("x")])

    model = 1 if a.model == "sand" else 0
    dx, dt = 1.0 / a.grid, 1e-4
    ext = [int(x) for x in a.cells.split(",")]
    glo = [(a.grid - ext[0]) // 2 // a.side * a.side, a.lift // a.side * a.side, (a.grid - ext[2]) // 2 // a.side * a.side]
    ghi = [glo[d] + ext[d] for d in range(3)]
    lo, hi = cell_box(rank, world, glo, ghi, align=a.side)
    vol = dx ** 3 / 8

    pol = zpc_amd.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
    # the exchange steps run inside libzsrocm.so on RCCL (zs_rocm_dist_*); torch.distributed only launches the ranks, carries the
    # unique id and brackets the timed region
    comm = None
    proxy = a.rank_proxy > 0 and world == 1 and a.slotted
    if proxy:
        comm = NativeComm(0, 1, local_rank, None)
    if world > 1 and a.backend == "nccl" and a.comm == "native":
        try:
            comm = NativeComm(rank, world, local_rank, dist)
        except RuntimeError as e:  # (ncclCommInitRank is collective: it fails on every rank or on none)
            print("[bench] native RCCL communicator unavailable (%s): exchange steps go through torch.distributed" % e, file=sys.stderr)
        if comm is not None:
            cc = lib().zs_rocm_dist_comm_count(comm._h)
            if cc != world:
                raise SystemExit("rank %d: RCCL reports ncclCommCount = %d for --gpus %d" % (rank, cc, world))
            if rank == 0:
                print("[bench] RCCL communicator: ncclCommCount = %d, %d GPU(s) visible" % (cc, torch.cuda.device_count()), file=sys.stderr)
    max_vel = torch.zeros(1, dtype=torch.float32, device=device)
    if a.tag_mass and ext[0] * ext[1] * ext[2] * 8 >= 1 << 22:
        raise SystemExit("--tag-mass numbers the particles in the mantissa of their mass: fewer than 2^22 of them")
    aos = generate_particles(lo, hi, dx, 1234, device, model, tag_box=(glo, ext) if a.tag_mass else None)
    drift_v = [float(x) for x in a.drift.split(",")]
    for k in range(3):
        if drift_v[k] != 0.0:
            aos[:, 4 + k] += drift_v[k]
    n_local = aos.shape[0]
    mt = MpmTransfer(pol, n_local, dx, dt, model=model, side=a.side, volume=vol, lane_width=a.lane_width, device=device,
                     cache_stress=not a.no_cache_stress)
    if mt.cache_stress:
        aos = torch.cat([aos, torch.zeros(n_local, mt.nchn - aos.shape[1], dtype=torch.float32, device=device)], dim=1).contiguous()  # the cached stress channels
    lib().zs_rocm_tv_from_aos_f32(pol.handle, aos.data_ptr(), n_local, mt.nchn, mt.L, mt.buf.data_ptr())
    torch.cuda.synchronize()
    del aos
    # ---- partition, block numbering, bins, halo lists (re-run after every re-partition)
    nc = a.side ** 3
    stage = {}
    # boundary blocks first, their ghost sums travel on a second stream while the interior blocks compute (compact and slotted storage)
    overlap = (world > 1 or proxy) and a.fused and not a.no_overlap
    comm_stream = torch.cuda.Stream(device=device, priority=-1) if overlap else None
    pol_comm = zpc_amd.rocm_exec().sync(False).external_stream(comm_stream.cuda_stream) if overlap else pol
    one_call = a.slotted and not a.py_step and (world == 1 or comm is not None)   # the step behind zs_rocm_mpm_step_slotted
    range_schedule = {"auto": 2 if a.side == 8 else 0, "in-turn": 0, "side-by-side": 1, "one-launch": 2}[a.range_schedule]
    proxy_grid = None
    ev_boundary, ev_comm = torch.cuda.Event(), torch.cuda.Event()
    n_boundary = 0

    def dev_buf(buf):
        if buf.device.type != "cpu":
            return buf
        if buf.data_ptr() not in stage:
            stage[buf.data_ptr()] = torch.empty(buf.numel(), dtype=torch.float32, device=device)
        return stage[buf.data_ptr()]

    def make_pack(p):
        def pack(blocks, nb, buf):
            d = dev_buf(buf)
            lib().zs_rocm_mpm_halo_pack(p.handle, mt.grid.data_ptr(), blocks.data_ptr(), nb, a.side, 0, a.halo_channels, d.data_ptr())
            if d is not buf:
                buf.copy_(d)  # gloo validation path: stage through host memory
        return pack

    def make_unpack(p):
        def unpack_add(blocks, nb, buf):
            d = dev_buf(buf)
            if d is not buf:
                d.copy_(buf)
            lib().zs_rocm_mpm_halo_unpack(p.handle, mt.grid.data_ptr(), blocks.data_ptr(), nb, a.side, 0, a.halo_channels, d.data_ptr(), 2)
        return unpack_add

    pack, unpack_add = make_pack(pol), make_unpack(pol)
    pack_c, unpack_add_c = make_pack(pol_comm), make_unpack(pol_comm)

    def exchange(on_comm_stream=False):
        """ghost-block sums of the grid just written: += every peer's partial sums"""
        if halo is None:
            return
        if comm is not None:
            halo.exchange_native(comm, pol_comm if on_comm_stream else pol, proxy_grid if proxy else mt.grid, a.side, 0, a.halo_channels)
        elif on_comm_stream:
            halo.exchange(pack_c, unpack_add_c)
        else:
            halo.exchange(pack, unpack_add)

    def grid_update():
        """grid momenta -> velocities; the CFL bound max |v|^2 over ALL ranks' nodes lands in max_vel (device)"""
        if a.no_cfl:
            mt.grid_update((0.0, -9.8, 0.0))
            return
        mt._zero(max_vel)
        mt.grid_update((0.0, -9.8, 0.0), max_vel)
        if comm is not None:
            comm.allreduce(pol, max_vel, "max")
        elif dist is not None:
            t = max_vel if a.backend == "nccl" else max_vel.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if t is not max_vel:
                max_vel.copy_(t)

    def partition_and_halo():
        """sparse-grid partition of the local particles; with the overlapped exchange the blocks near a rank boundary are
        numbered first; bins; ghost-block lists"""
        nonlocal n_boundary, proxy_grid
        from zpc_amd.dist import gather_block_keys, near_shared_mask
        if a.block_axes == "auto":
            longest = max(range(3), key=lambda d: ext[d])
            block_axes = ([d for d in range(3) if d != longest] + [longest]) if a.slotted else [0, 1, 2]
        else:
            block_axes = [int(x) for x in a.block_axes.split(",")]
        nb_ = mt.build_partition(max(4096, mt.n // 128), margin=a.margin if a.slotted else 0, order=a.block_order, axes=block_axes)
        all_keys = None
        if world > 1 and (overlap or comm is None):
            all_keys = gather_block_keys(dist, world, mt.active_keys(), comm_dev)
            if overlap:
                # blocks whose launch must precede the exchange: 8^3 blocks hold their 4^3 bins and those bins' exact-path particles
                # (at most one bin away, drift flag) within one block of themselves; 4^3 blocks (block = bin) within two
                # (slotted storage: a bin writes grid nodes at most one block away from its own block -- its stencil arena, and the
                # movers it finishes, which land at most one cell outside the bin)
                n_boundary = mt.reorder_partition(near_shared_mask(all_keys[rank], all_keys, rank, mt.kstride,
                                                                   margin=1 if (a.side == 8 or a.slotted) else 2))
        proxy_blocks = None
        if proxy:
            # the blocks an N-rank job would share with its neighbours: the two outermost block layers of this box on every cut face
            # (slabs along y for N < 8, 2x2x2 for 8: three cut faces) -- numbered first, exchanged with the rank itself over RCCL
            from zpc_amd.dist import NativeHaloPlan
            keys = mt.active_keys()
            kb = keys // mt.kstride
            lo_b, hi_b = kb.min(0), kb.max(0)
            axes = (0, 1, 2) if a.rank_proxy >= 8 else (1,)
            shared = np.zeros(keys.shape[0], bool)
            for d in axes:
                shared |= kb[:, d] >= hi_b[d] - 2
                if a.rank_proxy < 8:
                    shared |= kb[:, d] <= lo_b[d] + 2
            near = shared.copy()
            for d in axes:   # one more layer: launched before the exchange (see near_shared_mask)
                near |= kb[:, d] >= hi_b[d] - 3
                if a.rank_proxy < 8:
                    near |= kb[:, d] <= lo_b[d] + 3
            order = np.concatenate([np.nonzero(near)[0], np.nonzero(~near)[0]])
            n_boundary = mt.reorder_partition(near)
            blocks = np.nonzero(shared[order])[0].astype(np.int32)
            proxy_blocks = blocks
        if not a.unbinned:
            mt.rebin()
        stage.clear()
        h = None
        if proxy:
            from zpc_amd.dist import NativeHaloPlan
            h = NativeHaloPlan.from_lists(comm, a.side, [(0, 0, proxy_blocks.shape[0])], proxy_blocks)
            proxy_grid = torch.zeros_like(mt.grid)
            return nb_, h
        if world > 1 and comm is not None:
            # key all-gather, shared-block lists and exchange buffers inside the library (zs_rocm_dist_halo_plan_*)
            from zpc_amd.dist import NativeHaloPlan
            pol.syncCtx()
            h = NativeHaloPlan(comm, pol, mt.table, mt.nblocks, a.side)
        elif world > 1:
            if all_keys is None:
                all_keys = gather_block_keys(dist, world, mt.active_keys(), comm_dev)
            my_keys = mt.active_keys()

            def lookup(sk):
                d = torch.from_numpy(np.ascontiguousarray(sk)).to(device)
                r = torch.empty(sk.shape[0], dtype=torch.int32, device=device)
                mt.table.query(pol, d.data_ptr(), sk.shape[0], r.data_ptr())
                torch.cuda.synchronize()
                return r.cpu().numpy()

            h = HaloExchange(dist, rank, world, my_keys, lookup, lambda x: torch.from_numpy(x).to(device),
                             lambda m: torch.empty(max(m, 1), dtype=torch.float32, device=comm_dev), a.halo_channels * nc, all_keys=all_keys)
        return nb_, h

    t0 = time.perf_counter()
    nblocks, halo = partition_and_halo()
    torch.cuda.synchronize()
    rebin_ms = (time.perf_counter() - t0) * 1e3
    mt.update_stress()  # constitutive state for the first P2G (later ones get it from the preceding G2P)

    floor = None
    if a.floor:
        from zpc_amd.mpm import make_collider, PLANE, SEPARATE
        floor = make_collider(PLANE, SEPARATE, [0.0, 1.5 * dx, 0.0, 0.0, 1.0, 0.0])

    ev = lambda: torch.cuda.Event(enable_timing=True)
    p2g_ev, g2p_ev = [], []

    def step(timed):
        mt.clear_grid()
        if timed:
            e0, e1 = ev(), ev()
            e0.record()
        mt.p2g(binned=not a.unbinned)
        if timed:
            e1.record()
            p2g_ev.append((e0, e1))
        exchange()
        grid_update()
        if floor is not None:
            mt.apply_boundary(floor)
        if timed:
            e2, e3 = ev(), ev()
            e2.record()
        mt.g2p(binned=not a.unbinned)
        if timed:
            e3.record()
            g2p_ev.append((e2, e3))
        if os.environ.get("ZS_BENCH_GAP_CYCLES"):   # measurement only (profiles/r06_g2p.md): an idle stretch between G2P and the next P2G
            torch.cuda._sleep(int(os.environ["ZS_BENCH_GAP_CYCLES"]))

    fused_ev = []
    from zpc_amd.mpm import HipEvents, StepBreakdown
    hip_events = HipEvents()
    breakdown = StepBreakdown() if (world > 1 or proxy) else None   # where a rank's step time goes (events inside zs_rocm_mpm_step_slotted)

    handover = {"snap": None, "final": None, "steps": 0, "mismatched_steps": 0, "nonzero": 0}   # --check-handover

    ctrl_ev = []  # (start, end) events of the fused launches since the last look of the re-bin controller

    def step_fused(timed, write_all=False, reorder=False):
        # grid holds the velocities of the current step (after grid_update): G2P from it, P2G of the next step into the
        # second (zeroed) grid, which then becomes the current one
        track = timed or (a.rebin_check > 0 and not a.slotted)
        if track:
            e0, e1 = ev(), ev()
            e0.record()
        if one_call:
            # grid reset, both block ranges, exchange on the side stream, re-home / commit, grid update, CFL allreduce: ONE call;
            # the library records the event pair around the transfer kernels itself
            nbd = n_boundary if (overlap and halo is not None) else 0
            check = a.check_handover and nbd and comm is not None and 0 < nbd < mt.nblocks
            if check:
                if a.understate_boundary:   # negative control of the check: the exchange is released when only this share of the boundary blocks is done
                    nbd = max(1, int(nbd * a.understate_boundary))
                if handover["snap"] is None or handover["snap"].numel() != halo.total_blocks * nc:
                    handover["snap"] = torch.zeros(halo.total_blocks * nc, dtype=torch.float32, device=device)
                    handover["final"] = torch.zeros_like(handover["snap"])
            mt.step_slotted((0.0, -9.8, 0.0), None if a.no_cfl else max_vel, write_all=write_all,
                            n_boundary=nbd, comm=comm, plan=halo if comm is not None else None,
                            comm_pol=pol_comm if overlap else None, collider=floor, halo_grid=proxy_grid,
                            events=hip_events.pair() if timed else None,
                            breakdown=breakdown.next() if (timed and breakdown is not None) else None, halo_channels=a.halo_channels,
                            range_schedule=range_schedule, handover_snapshot=handover["snap"] if check else None)
            if check:
                # the mass channel of the shared blocks as the exchange stream saw it when the exchange started against the same blocks after the
                # step: only boundary blocks write those nodes (and the proxy's exchange goes to a scratch grid), so the two are equal BIT FOR BIT
                # exactly when the hand-over let the exchange see complete sums
                torch.cuda.synchronize()
                lib().zs_rocm_mpm_halo_pack(pol.handle, mt.grid.data_ptr(), lib().zs_rocm_dist_halo_plan_block_list(halo._h), halo.total_blocks, a.side, 0, 1,
                                            handover["final"].data_ptr())
                pol.syncCtx()
                torch.cuda.synchronize()
                handover["steps"] += 1
                handover["mismatched_steps"] += int(not torch.equal(handover["snap"], handover["final"]))
                handover["nonzero"] = max(handover["nonzero"], int((handover["final"] != 0).sum().item()))
            return
        if overlap and halo is not None and 0 < n_boundary < mt.nblocks:
            # boundary blocks first; their ghost sums travel on the communication stream while the interior blocks compute
            mt.g2p2g(write_all=write_all, split=n_boundary, between=lambda: ev_boundary.record(), reorder=reorder)
            if track:
                e1.record()
                ctrl_ev.append((e0, e1))
            if timed:
                fused_ev.append((e0, e1))
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev_boundary)
                exchange(on_comm_stream=True)
                ev_comm.record()
            torch.cuda.current_stream().wait_event(ev_comm)
        else:
            mt.g2p2g(write_all=write_all, reorder=reorder)
            if track:
                e1.record()
                ctrl_ev.append((e0, e1))
            if timed:
                fused_ev.append((e0, e1))
            exchange()
        grid_update()
        if floor is not None:
            mt.apply_boundary(floor)
    migrated = 0

    def prime_grid():
        mt.clear_grid()
        mt.p2g()
        exchange()
        grid_update()
        if floor is not None:
            mt.apply_boundary(floor)

    def remap():
        """particles -> owning ranks, new partition / halo lists / bins (full particle state must be in memory)"""
        nonlocal halo, nblocks, migrated
        from zpc_amd.dist import migrate_particles
        moved = (0, 0)
        if inplace_remap:
            # single rank, slotted storage: the partition follows the particles without touching one (new partition from the occupancy
            # words, populated bins move as whole tile rows, the velocity grid is carried over): ~4 ms instead of ~50
            nblocks = mt.repartition_slotted(margin=a.margin, strict=(a.repartition == "closed" and not os.environ.get("ZS_BENCH_ABLATION")))
            return moved
        if mt.slotted:
            # occupied slots -> compact buffer (the step before a re-map stored v, C and the stress as well).  The period's status words
            # are checked and folded into mt.slot_record first, and the particle count must be unchanged: nothing is ever dropped
            mt.unslot(strict=(a.repartition == "closed" and not os.environ.get("ZS_BENCH_ABLATION")))
        if world > 1:
            to_c = (lambda t: t) if a.backend == "nccl" else (lambda t: t.cpu())
            moved = migrate_particles(mt, pol, dist, rank, world, glo, ghi, a.side, to_c, lambda t: t.to(device), comm=comm)
        nblocks, halo = partition_and_halo()
        if a.fused:
            prime_grid()
        if a.slotted:
            mt.slot(K=a.slot_rounds, outbox_cap=a.outbox_cap)
        migrated += moved[0]
        return moved

    # re-partition in place (zs_rocm_mpm_reslot) where particles cannot change owner and no boundary-first block numbering is needed
    inplace_remap = a.slotted and a.fused and world == 1 and not proxy and not os.environ.get("ZS_BENCH_FULL_REMAP")
    if a.fused:
        if a.unbinned or not mt.cache_stress:
            raise SystemExit("--fused needs the binned path with cached stress")
        prime_grid()
        if a.slotted:
            mt.slot(K=a.slot_rounds, outbox_cap=a.outbox_cap)
            if inplace_remap:
                mt.reserve_repartition_buffers()   # (setup, untimed: the second slot buffer a re-partition in place moves the bins into)
        unfused_step = step
        step = lambda timed, write_all=False, reorder=False: step_fused(timed, write_all, reorder)
    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    K = a.migrate_every
    remaps = [0]
    remap_steps = []
    next_remap = None   # --repartition open: step count after which the next re-partition is due
    closed_loop = a.slotted and K == 0 and a.repartition == "closed"
    poll_iv = max(1, a.side // 4)   # the early warning leaves `side` cells of travel; the answer lags one interval, the re-map one more step
    pending_remap = False

    def steps_in_margin(s0):
        """(--repartition open) how many steps from step s0 on the column (initial drift + gravity) stays inside the partition's margin"""
        safe_cells = max(a.margin, 0) * a.side + a.side // 2 - 2
        v0 = max(abs(x) for x in drift_v) + 9.8 * s0 * dt   # (upper bound: gravity added to the largest component)
        k = 0
        while k < 100000 and (v0 * (k + 1) * dt + 0.5 * 9.8 * ((k + 1) * dt) ** 2) / dx <= safe_cells:
            k += 1
        return k
    if K == 0 and a.slotted and a.repartition == "open":
        n_steps = a.warmup + a.steps + 2
        k = steps_in_margin(0)
        if k < n_steps:
            next_remap = max(k, 8)

    def reduce_flags(t):
        if comm is not None:
            comm.allreduce(pol, t, "max")
        elif dist is not None:
            tt = t if a.backend == "nccl" else t.cpu()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if tt is not t:
                t.copy_(tt)
    done = 0

    # re-bin controller: particles that leave their cell make the fused launch slower step by step (LDS queue, exact path); a
    # re-bin resets that at a fixed cost.  Re-bin when the time lost since the last re-bin (sum of launch time - best launch time)
    # reaches the cost of a re-bin: for a linearly growing loss that is the period that minimises the average step time.
    rebins = 0
    check_iv = max(a.rebin_check, 1)
    next_check = check_iv
    best_ms, lost_ms, rebin_cost_ms = None, 0.0, None

    rebin_at = set(int(x) for x in a.rebin_at.split(",")) if a.rebin_at else None
    rebin_steps = []

    def run_steps(count, timed):
        nonlocal done, rebins, check_iv, next_check, best_ms, lost_ms, rebin_cost_ms, next_remap, pending_remap
        for _ in range(count):
            remap_now = (K > 0 and (done + 1) % K == 0) or (next_remap is not None and done + 1 >= next_remap) or pending_remap
            ts = time.perf_counter()
            if a.fused:
                step(timed, remap_now and not inplace_remap)  # the step before a FULL re-map materialises v, C, stress of every particle
            else:
                step(timed)
            if timed:
                host_step_s[0] += time.perf_counter() - ts   # host time inside the step's own call(s): enqueue cost, no poll waits
            done += 1
            if remap_now:
                remap()
                remaps[0] += 1
                remap_steps.append(done)
                pending_remap = False
                if next_remap is not None:
                    next_remap = done + max(steps_in_margin(done), 8)
                ctrl_ev.clear()
                best_ms, lost_ms = None, 0.0
            elif closed_loop and done % poll_iv == 0:
                # the step's own status words ask for the re-partition (word [3]: a particle lives in a block next to the partition's
                # edge; or an overflow flag): asynchronous copy, answer of the previous poll -- no stall, same decision on every rank
                pending_remap = mt.poll_repartition(reduce_flags if world > 1 else None)
            elif a.fused and not a.slotted and rebin_at is not None:
                if done in rebin_at:
                    mt.rebin(inputs_only=True)
                    rebins += 1
                    rebin_steps.append(done)
                ctrl_ev.clear()
            elif a.fused and not a.slotted and a.rebin_check > 0 and done >= next_check:
                ctrl_ev[-1][1].synchronize()  # the look stalls the stream: done less often while nothing is being lost
                for e0, e1 in ctrl_ev:
                    t = e0.elapsed_time(e1)
                    best_ms = t if best_ms is None else min(best_ms, t)
                    lost_ms += max(0.0, t - 1.01 * best_ms)
                ctrl_ev.clear()
                if rebin_cost_ms is None:
                    rebin_cost_ms = 0.6 * best_ms  # first guess: 14 of 32 channels read + written, plus the ordering passes
                if lost_ms >= rebin_cost_ms:
                    # particles only (partition, block numbers and halo lists stay), and only the channels the next fused step
                    # reads: m, x, F, logJp -- v, C and the stress are recomputed from the grid.  (MpmTransfer.g2p2g(reorder=True)
                    # folds the re-bin into the step instead; with today's binning order its scattered reads cost as much.)
                    r0, r1 = ev(), ev()
                    r0.record()
                    mt.rebin(inputs_only=True)
                    r1.record()
                    r1.synchronize()
                    rebin_cost_ms = r0.elapsed_time(r1)
                    rebins += 1
                    rebin_steps.append(done)
                    best_ms, lost_ms = None, 0.0
                    check_iv = a.rebin_check
                elif lost_ms < 0.1 * rebin_cost_ms:
                    check_iv = min(check_iv * 2, 8 * a.rebin_check)
                else:
                    check_iv = a.rebin_check
                next_check = done + check_iv

    host_step_s = [0.0]
    run_steps(a.warmup, False)
    barrier()
    probe = os.environ.get("ZS_ROCM_PROBE") and hasattr(lib(), "zs_rocm_debug_probe")  # measurement builds only (-DZS_PROBE, tools/ab_build.sh)
    slot_probe = os.environ.get("ZS_ROCM_PROBE") and hasattr(lib(), "zs_rocm_slot_probe")  # -DZS_PROBE build of mpm_slotted.hip
    p2g_probe = os.environ.get("ZS_ROCM_PROBE") and hasattr(lib(), "zs_rocm_p2g_probe")  # -DZS_PROBE_P2G build of mpm_p2g.hip
    if probe or slot_probe:
        import ctypes
        pv = (ctypes.c_ulonglong * 32)()
        (lib().zs_rocm_slot_probe if slot_probe else lib().zs_rocm_debug_probe)(pv, 1)
    if p2g_probe:
        import ctypes
        ppv = (ctypes.c_ulonglong * 16)()
        lib().zs_rocm_p2g_probe(ppv, 1)
    t0 = time.perf_counter()
    run_steps(a.steps, True)
    host_enqueue_s = time.perf_counter() - t0   # when the last step was enqueued (== elapsed once the launch queue is full)
    barrier()
    elapsed = time.perf_counter() - t0
    if p2g_probe:
        lib().zs_rocm_p2g_probe(ppv, 0)
        nw = max(int(ppv[8]), 1)
        names = ["life", "head", "stream", "stream: in vmcnt waits", "barrier + arena clear", "27 phases", "post-pass + atomics issued", "atomics drained"]
        print("p2g probe (s_memtime ticks per sampled wave): " + "  ".join("%s=%.0f" % (names[k], ppv[k] / nw) for k in range(8)) +
              "  rounds=%.2f  waves=%d" % (ppv[9] / nw, nw), file=sys.stderr)
    if slot_probe:
        lib().zs_rocm_slot_probe(pv, 0)
        wgs = max(int(pv[11]), 1)
        names = ["wg total", "head", "arena fill+barrier", "prod work", "prod barrier wait", "prod final work", "prod final barrier", "cons work",
                 "cons barrier wait", "cons flush", "tail", "wgs", "list entries", "cons: rounds loop", "cons: atomic list", "cons: loop iterations"]
        print("slot probe (cycles per sampled workgroup; 100 MHz s_memtime ticks x ?): " +
              "  ".join("%s=%.0f" % (names[k], pv[k] / wgs) for k in range(16) if k != 11) + "  wgs=%d" % wgs, file=sys.stderr)
        if any(pv[k] for k in range(16, 32)):  # block kernel: the producer wave's chunk iteration by segment
            seg = ["top of iteration (hand-over, prefetch, tables)", "arena + gather", "advection + F", "movers + stores", "constitutive update",
                   "tail stores", "ring wait", "staging", "vmcnt(0) before the barrier", "top: descriptors + hand-over", "top: record requests",
                   "top: finish_bin / neighbour bins", "top: entry table"]
            print("producer segments (cycles per sampled workgroup): " + "  ".join("%s=%.0f" % (seg[k], pv[16 + k] / wgs) for k in range(13)), file=sys.stderr)
    if probe:
        lib().zs_rocm_debug_probe(pv, 0)
        wgs = max(int(pv[7]), 1)
        print("probe (cycles per workgroup, 100 MHz-agnostic s_memtime ticks): " +
              " ".join("[%d]=%.0f" % (k, pv[k] / wgs / (4 if k in (0, 1, 4, 5, 6) else 8)) for k in range(7)) + " wgs=%d" % wgs, file=sys.stderr)
    if a.fused and (a.checksum or a.dump_state):
        step_fused(False, write_all=True)  # untimed: materialise v, C, stress of every particle for the checksum
        torch.cuda.synchronize()
    err = lib().zs_rocm_last_error(-1)
    drift = int(mt.margin_violated())
    if overlap and drift:
        # an exact-path particle of an interior block may have reached a shared block after its ghost sums were sent
        raise SystemExit("rank %d: particles drifted more than one bin from their bins -- re-bin more often (--migrate-every) "
                         "or run with --no-overlap" % rank)

    movers_per_step = None
    slot_record = None
    if a.slotted:
        try:
            mt.check_slots(strict=(a.repartition == "closed"))   # the last period; earlier ones were folded in by every unslot()
        except RuntimeError as e:
            if not os.environ.get("ZS_BENCH_ABLATION"):  # (measurement builds with parts of the step stubbed lose particles by construction)
                raise SystemExit("rank %d: %s -- raise --slot-rounds / --outbox-cap / --margin" % (rank, e))
        rec = mt.slot_record
        cnt_now = int(lib().zs_rocm_mpm_slot_list(pol.handle, mt.cell_mask.data_ptr(), mt.nbins, mt.K, None))
        if cnt_now != mt.n and not os.environ.get("ZS_BENCH_ABLATION"):
            raise SystemExit("rank %d: the slotted storage holds %d particles, %d went in" % (rank, cnt_now, mt.n))
        movers_per_step = rec["sent"] / max(a.steps + a.warmup, 1)
        slot_record = {"movers_sent": rec["sent"], "movers_rehomed": rec["homed"], "periods": rec["periods"],
                       "periods_with_edge_warning": rec["edge_periods"],
                       "periods_with_flag": {mt.SLOT_FLAG_NAMES[k]: rec["flags"][k] for k in (0, 1, 2, 4)},
                       "particles_in_storage": cnt_now, "log": rec["log"][:16]}
    if a.fused and mt.left_partition():
        # the reference does not check this either (P2G.hpp:109-110), but a benchmark that loses mass is not a benchmark
        raise SystemExit("rank %d: particles left the sparse-grid partition (contributions dropped) -- use --migrate-every K to "
                         "rebuild the partition" % rank)
    n_local = mt.n
    n_total = n_local
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nt = torch.tensor([n_local], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(nt)
        n_total = int(nt.item())
    rank_breakdown = None
    if breakdown is not None and breakdown.steps:
        bs = breakdown.summary()
        names = [k for k, _, _ in StepBreakdown.STRETCHES]
        vals = torch.tensor([bs[k] if bs[k] is not None else 0.0 for k in names], dtype=torch.float64, device=comm_dev)
        vmax, vsum = vals.clone(), vals.clone()
        if dist is not None:
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(vsum)
        rank_breakdown = {"max_over_ranks": {k: float(vmax[i]) for i, k in enumerate(names)},
                          "mean_over_ranks": {k: float(vsum[i]) / max(world, 1) for i, k in enumerate(names)},
                          "note": "HIP events recorded inside zs_rocm_mpm_step_slotted (zs_rocm_mpm_step.evBreakdown), mean over the timed steps: "
                                  "boundary range | interior range + re-home + commit | main stream waiting for the exchange | grid update | CFL "
                                  "allreduce(max); exchange_side_stream_ms = pack + grouped ncclSend/ncclRecv + unpack-add on the side stream "
                                  "(overlaps the interior range)"}
    slot_stats = None
    if a.slotted:
        # occupancy of the slotted storage at the end of the run: rounds a bin's producers walk (highest occupied round + 1, in chunks of
        # four) against the particles it holds
        m = mt.cell_mask.view(mt.nbins, 64).to(torch.int64) & 0xFFFFFFFF
        top = torch.zeros_like(m)
        cnt = torch.zeros_like(m)
        for r in range(mt.K):
            bit = (m >> r) & 1
            top = torch.where(bit > 0, torch.full_like(top, r + 1), top)
            cnt += bit
        rounds = top.max(dim=1).values.float()
        occ = rounds > 0
        per_bin = cnt.sum(dim=1).float()
        slot_stats = {"bins_occupied": int(occ.sum()), "mean_rounds": float(rounds[occ].mean()), "mean_chunk_rounds": float((torch.ceil(rounds[occ] / 4) * 4).mean()),
                      "mean_particles_per_bin_div64": float(per_bin[occ].mean() / 64), "max_rounds": int(rounds.max()),
                      "mean_max_cell_count": float(cnt.max(dim=1).values.float()[occ].mean())}
    if a.dump_state:
        dd = mt.download()
        np.savez(a.dump_state + ".rank%d.npz" % rank, **dd)
    checksum = checksum_trim = None
    if a.checksum:
        # order-independent global sums of the particle state (float64): equal for any number of ranks up to rounding
        cbuf, ctiles = mt.buf, mt.tiles
        if mt.slotted:
            cbuf, _ = mt._compact_copy()
            ctiles = cbuf.numel() // (mt.nchn * mt.L)
        v = cbuf.view(ctiles, mt.nchn, mt.L).double()
        valid = (torch.arange(ctiles * mt.L, device=device) < n_local).view(ctiles, 1, mt.L)
        sums = (v * valid).sum(dim=(0, 2))
        sq = ((v * valid) ** 2).sum(dim=(0, 2))
        cs = torch.cat([sums, sq]).to(comm_dev)
        # ... and the same sums without the particles that carry a velocity-gradient entry beyond 8 rms.  Two kinds exist: the two edge
        # particles of the column that have such entries in every run, and what a hit of the reference arena's rounding case leaves behind
        # (a local position that rounds to exactly 1.5 next to the coordinate origin is weighted one cell off and G2P returns C ~ 4 v / dx:
        # one foot particle in ~3 % of the compact-storage runs with the column on y = 0, 76 particles with |C| > 4 two steps later --
        # profiles/r03_compact_outliers.md; the reference computes the same).  Sum-of-squares comparisons between two runs use the
        # trimmed sums and bound the number of trimmed particles.
        c2 = v[:, 7:16, :] ** 2 * valid
        thr2 = 64.0 * float(c2.sum()) / max(9 * n_local, 1)
        keep = valid & ~((c2 > thr2).any(dim=1, keepdim=True))
        cs_trim = torch.cat([(v * keep).sum(dim=(0, 2)), ((v * keep) ** 2).sum(dim=(0, 2)), (valid & ~keep).sum().double().view(1)]).to(comm_dev)
        # where the trimmed particles are: the highest one (cells above the column's foot) and how many of them sit on the column's
        # side faces (the two edge particles every run trims)
        trimmed = (valid & ~keep)
        ty = torch.where(trimmed[:, 0, :], v[:, 2, :] / dx - glo[1], torch.full_like(v[:, 2, :], -1.0))
        cx, cz = v[:, 1, :] / dx, v[:, 3, :] / dx
        on_side = trimmed[:, 0, :] & ((cx < glo[0] + 1.5) | (cx > ghi[0] - 1.5) | (cz < glo[2] + 1.5) | (cz > ghi[2] - 1.5))
        tinfo = torch.stack([ty.max(), on_side.sum().double()]).to(comm_dev)
        if dist is not None:
            dist.all_reduce(cs)
            dist.all_reduce(cs_trim)
            tmax = tinfo[:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tinfo)
            tinfo[0] = tmax[0]
        checksum = [float(x) for x in cs.cpu()]
        checksum_trim = [float(x) for x in cs_trim.cpu()]
        trim_info = [float(x) for x in tinfo.cpu()]
    p2g_ms = float(np.mean([x.elapsed_time(y) for x, y in p2g_ev])) if p2g_ev else 0.0
    if os.environ.get("ZS_BENCH_PER_STEP") and rank == 0 and p2g_ev:
        print("per-step p2g ms:", " ".join("%.4f" % x.elapsed_time(y) for x, y in p2g_ev), "| g2p ms:", " ".join("%.4f" % x.elapsed_time(y) for x, y in g2p_ev), file=sys.stderr)
    g2p_ms = float(np.mean([x.elapsed_time(y) for x, y in g2p_ev])) if g2p_ev else 0.0
    fused_ms = float(np.mean([x.elapsed_time(y) for x, y in fused_ev])) if fused_ev else 0.0
    if one_call and hip_events.pairs:
        fused_list = hip_events.elapsed_ms()
        fused_ms = float(np.mean(fused_list)) if fused_list else 0.0
    if os.environ.get("ZS_BENCH_PER_STEP") and rank == 0:
        print("per-step fused ms:", " ".join("%.3f" % v for v in (hip_events.elapsed_ms() if one_call else [x.elapsed_time(y) for x, y in fused_ev])), file=sys.stderr)

    if rank == 0:
        value = n_total * a.steps / elapsed
        # cached stress: the algorithmic figure stays SURVEY 8(d)'s 107 B (m, x, v, C + a 3x3 state + 7 B grid); the kernel itself
        # reads the symmetric P F^T (6 floats: 88 + 7 B moved); G2P additionally reads/writes logJp and writes the 6 floats
        p2g_bytes = 107.0 if mt.cache_stress else P2G_BYTES[model]
        g2p_bytes = G2P_BYTES + ((24.0 + (8.0 if model == 1 else 0.0)) if mt.cache_stress else 0.0)
        ach = p2g_bytes * n_local / (max(p2g_ms, 1e-9) * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_p2g.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if (j.get("particles") == n_local and j.get("side") == a.side and j.get("model") == a.model
                        and j.get("cache_stress", False) == mt.cache_stress and _same_code(j)):
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                pass
        workload = ("%s %dx%dx%d cells, 8 particles/cell = %d particles, dx=1/%d (%d^3 sparse grid), %s, %d^3-cell grid blocks "
                    "(bht<int,3,int,16> + TileVector<f32,%d^3> {m,v,rhs}), TileVector<f32,%d> particles; step = grid reset + P2G + grid "
                    "update + G2P%s"
                    % ("MPM sand column" if model else "MPM elastic jello block", ext[0], ext[1], ext[2], n_total, a.grid, a.grid,
                       "DruckerPrager" if model else "FixedCorotated", a.side, a.side, a.lane_width,
                       "" if not a.unbinned else " [particle-order path]"))
        out = {
            "metric": "particle*steps/s (P2G+G2P)", "value": value, "unit": "particle*steps/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "particles": n_total, "grid_blocks_rank0": nblocks, "decomposition": "x".join(map(str, zpc_amd.dist.split_dims(world))),
                       "halo_bytes_per_step_rank0": ((halo.bytes_per_exchange * a.halo_channels // 7 if comm is not None else halo.bytes_per_exchange) if halo and halo.peers else 0),
                       "halo_channels": a.halo_channels, "handover": ({k: handover[k] for k in ("steps", "mismatched_steps", "nonzero")} if a.check_handover else None),
                       "block_order": getattr(mt, "block_order", None), "block_axes": list(getattr(mt, "block_axes", None) or (0, 1, 2)),
                       "step_schedule": (("ranges in turn", "ranges side by side", "one launch + gate")[range_schedule] if (one_call and overlap) else "one range"),
                       "boundary": "plane collider (Separate) at y = 1.5 dx" if a.floor else "none", "rebins": rebins, "halo_overlap": bool(overlap), "exchange": ("rccl via libzsrocm (zs_rocm_dist_*)" if comm is not None else ("torch.distributed/" + a.backend if world > 1 else "none")),
                       "cfl_max_vel_sqr": (None if a.no_cfl else float(max_vel.item())), "boundary_blocks_rank0": n_boundary,
                       "rebin_ms_once": rebin_ms, "migrate_every": K, "repartitions": remaps[0], "migrated_rank0": migrated,
                       "drift_m_per_s": drift_v, "cells_per_step": max(abs(x) for x in drift_v) * dt / dx,
                       "storage": ("slotted: bins x %d rounds x 64 lanes + per-cell occupancy masks; a mover is finished by the workgroup that "
                                   "moves it (new slot by ticket inside its bin; across bins: global atomics + an outbox record of %d per bin, "
                                   "re-homed by a second small kernel) -- no re-bins in the time loop"
                                   % (a.slot_rounds, a.outbox_cap)) if a.slotted else "compact round-robin order + re-bin controller",
                       "movers_per_step_rank0": movers_per_step, "partition_margin_blocks": a.margin if a.slotted else 0,
                       "repartition_trigger": (("closed loop: status word [3] of the slotted step, polled every %d steps" % poll_iv) if closed_loop
                                               else ("every %d steps" % K if K else ("open loop (drift + gravity)" if a.slotted else "none"))),
                       "repartition_steps": remap_steps[:64], "repartition_kind": ("in place (zs_rocm_mpm_reslot: bins move as whole tile rows)" if inplace_remap else "full (unslot, partition, re-bin, prime, slot)"),
                       "slot_record_rank0": slot_record,
                       "step_call": ("one C-ABI call per step (zs_rocm_mpm_step_slotted)" if one_call else "python: one call per kernel / exchange"),
                       "host_step_call_us_per_step": host_step_s[0] / a.steps * 1e6,   # time the host spends inside the step's call(s)
                       # wall clock until the last step was enqueued: includes the closed-loop poll, which waits for the status copy of
                       # two steps earlier (so this follows the device time; the host cost proper is host_step_call_us_per_step)
                       "host_enqueue_us_per_step": host_enqueue_s / a.steps * 1e6},
            "roofline": {"bound": "hbm", "kernel": ("p2g_global_kernel" if a.unbinned else (("p2g_tile_kernel" if a.lane_width == 64 else "p2g_wide_kernel") if mt.cache_stress else ("update_stress_kernel + p2g_tile_kernel" if a.lane_width == 64 else "p2g_binned_kernel"))),
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_particle": p2g_bytes, "particles_per_launch": n_local, "launch_ms": p2g_ms,
                         "constitutive_update": "tail of previous G2P (particles.stress)" if mt.cache_stress else "inside P2G",
                         "g2p": {"achieved": g2p_bytes * n_local / (max(g2p_ms, 1e-9) * 1e-3) / 1e9, "launch_ms": g2p_ms,
                                 "bytes_per_particle": g2p_bytes}},
            "hip_error": err,
        }
        if a.fused:
            # algorithmic bytes of the work one launch does = one G2P + one P2G per particle: SURVEY.md 8(d), 145.5 + 107 (115
            # with logJp) B.  What the fused pass actually has to move is less: m, x, F (, logJp) in (52 / 56 B), x, F (, logJp)
            # out (48 / 52 B), grid A velocities 1.5 B, grid B clear + accumulate 7 B
            fb = 260.5 if model == 1 else 252.5
            fmin = (56.0 + 52.0 if model == 1 else 52.0 + 48.0) + 1.5 + 7.0
            fach = fb * n_local / (fused_ms * 1e-3) / 1e9
            ftraffic = None
            # PMC traffic of the kernels this run used: slotted storage under motion (pmc_g2p2g.json, tools/refresh_r05.sh) or the
            # compact-storage kernel at rest (pmc_g2p2g_compact.json); no figure was collected for the other combinations
            moving = any(abs(x) > 0 for x in drift_v)
            per_bin = a.side != 8   # 4^3 blocks: one workgroup per bin (= block); 8^3 blocks: one workgroup per block
            fkernel = (("g2p2g_slot_kernel" if per_bin else "g2p2g_slotblk_kernel") + " + slot_rehome_kernel + slot_commit_kernel") if a.slotted else "g2p2g_rs_kernel"
            fvalu = None
            pmcf = os.path.join(ROOT, "profiles", "pmc_g2p2g.json" if (a.slotted and moving) else "pmc_g2p2g_compact.json")
            if os.path.exists(pmcf) and (a.slotted == moving):
                try:
                    j = json.load(open(pmcf))
                    if (j.get("particles") == n_local and j.get("side") == a.side and j.get("model") == a.model and j.get("kernel") == fkernel
                            and _same_code(j)):
                        ftraffic = j.get("hbm_bytes_per_launch")   # (a figure collected for another kernel generation is not this run's traffic)
                        if j.get("valu_insts_per_launch"):
                            # the other roofline: VALU issue.  insts = SQ_INSTS_VALU of the step's kernels; cycles_per_inst = shader cycles a
                            # SIMD spent per VALU instruction of the main kernel (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs / SQ_INSTS_VALU);
                            # issue_peak = what a SIMD of this chip issues with >= 2 resident waves: 2.2 cycles per independent v_fma_f32
                            # (tools/valu_issue_bench.hip; one wave alone: 4.6), 2.9 for the instruction mix of this kernel's 3x3 SVD +
                            # return mapping at 4 waves per SIMD (tools/svd_issue_bench.hip: v_cndmask / v_cmp / v_max issue at 4 cycles,
                            # transcendentals at 8); algorithmic = the same kernels' SQ_INSTS_VALU on the column at rest
                            cpi = j.get("valu_cycles_per_inst_per_simd")
                            fvalu = {"insts_per_launch": j["valu_insts_per_launch"],
                                     "cycles_per_inst_per_simd": cpi,
                                     "issue_peak_cycles_per_inst": VALU_ISSUE_PEAK_CYCLES, "issue_mix_cycles_per_inst": VALU_ISSUE_MIX_CYCLES,
                                     "issue_frac": (VALU_ISSUE_PEAK_CYCLES / cpi) if cpi else None,
                                     "issue_frac_of_mix_rate": (VALU_ISSUE_MIX_CYCLES / cpi) if cpi else None,
                                     # a COUNTED floor, not another run of the same kernel: wave-level VALU instructions per 64 particles of the
                                     # stages a fused step cannot do without, every lane busy -- G2P gather by sum factorisation 290, advection + F
                                     # update + 3x3 SVD + return mapping 860 (DruckerPrager; FixedCorotated 700), Q-form staging 150, P2G accumulate
                                     # of the four channel sets at full lane occupancy 628
                                     "algorithmic_insts": VALU_FLOOR_PER_64[model] * n_local / 64.0,
                                     "algorithmic_insts_per_64_particles": VALU_FLOOR_PER_64[model],
                                     "frac": VALU_FLOOR_PER_64[model] * n_local / 64.0 / j["valu_insts_per_launch"],
                                     "insts_at_rest": j.get("valu_insts_at_rest"),   # the same kernels on the column at rest (no movers, no holes)
                                     "ns_per_inst_per_simd": fused_ms * 1e6 * 1024 / j["valu_insts_per_launch"],
                                     "source": j.get("source")}
                except Exception:
                    pass
            out["config"]["workload"] = out["config"]["workload"].replace("step = grid reset + P2G + grid update + G2P",
                                                                          "step = grid reset + fused G2P2G (G2P of step n, P2G of step n+1) + grid update")
            out["roofline"] = {"bound": "hbm", "kernel": fkernel, "achieved": fach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": fach / HBM_PEAK_GBS, "traffic": ftraffic, "bytes_per_particle": fb,
                               "particles_per_launch": n_local, "launch_ms": fused_ms,
                               "fused_min_bytes_per_particle": fmin,
                               "note": "bytes_per_particle = SURVEY 8(d) P2G + G2P; the fused pass keeps v, C and the stress on chip "
                                       "(traffic < algorithmic bytes) and is bound by instruction issue and the waves' own chains, not by HBM "
                                       "(profiles/r03_pmc_g2p2g.md, r03_slot_probe.md); "
                                       "launch_ms = HIP-event time of the fused launches of one step (slotted: main kernel + re-home + commit kernels)"}
            out["roofline"]["hbm_frac"] = fach / HBM_PEAK_GBS
            if fvalu is not None:
                out["roofline"]["valu"] = fvalu
                # the binding roofline is the one the kernel sits closer to: the fraction of the HBM peak its algorithmic bytes reach, against
                # the fraction of the SIMDs' VALU issue rate its instruction stream reaches
                if fvalu.get("issue_frac") and fvalu["issue_frac"] > fach / HBM_PEAK_GBS:
                    out["roofline"]["bound"] = "valu"
        # the measured device-copy ceiling of THIS box beside the nominal peak (1 GiB device-to-device copies, read + write
        # bytes over HIP-event time, after the timed region)
        try:
            ca = torch.empty(1 << 28, dtype=torch.float32, device=device)
            cb = torch.empty_like(ca)
            cb.copy_(ca)
            ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ce[0].record()
            for _ in range(5):
                cb.copy_(ca)
            ce[1].record()
            torch.cuda.synchronize()
            copy_gbs = 2.0 * ca.numel() * 4 * 5 / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9
            del ca, cb
            out["roofline"]["measured_copy"] = copy_gbs
            out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / copy_gbs
            # ... and the read-only stream ceiling (P2G is 80 % reads: the copy ceiling understates what a read-dominated kernel can
            # reach): zs::reduce<i32, plus> over 1 GiB, HIP-event time on the policy's stream
            import zpc_amd as _zs
            ra = torch.zeros(1 << 28, dtype=torch.int32, device=device)
            r1 = torch.zeros(1, dtype=torch.int32, device=device)
            _zs.reduce(pol, ra, None, r1)
            ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ce[0].record()
            for _ in range(5):
                _zs.reduce(pol, ra, None, r1)
            ce[1].record()
            torch.cuda.synchronize()
            read_gbs = ra.numel() * 4 * 5 / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9
            del ra
            out["roofline"]["measured_read"] = read_gbs
            out["roofline"]["frac_of_measured_read"] = out["roofline"]["achieved"] / read_gbs
        except Exception:
            pass
        if checksum is not None:
            out["checksum"] = checksum
            out["checksum_trimmed"] = {"sums": checksum_trim[:-1], "trimmed_particles": int(checksum_trim[-1]),
                                       "trimmed_max_y_cells": trim_info[0], "trimmed_edge_particles": int(trim_info[1])}
        if slot_stats is not None:
            out["slot_stats"] = slot_stats
        if rank_breakdown is not None:
            out["rank_breakdown"] = rank_breakdown
        if comm is not None:
            out["config"]["nccl_comm_count"] = int(lib().zs_rocm_dist_comm_count(comm._h))
            out["config"]["gpus_visible"] = torch.cuda.device_count()
        if world == 1 and a.slotted and not a.no_at_rest and any(abs(x) > 0 for x in drift_v):
            # secondary numbers: the same column at rest (no movers) -- short sub-runs of this script after the timed region
            import subprocess
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--drift", "0,0,0", "--no-at-rest", "--no-cpu-baseline", "--steps", "10",
                       "--warmup", "3", "--grid", str(a.grid), "--cells", a.cells, "--model", a.model, "--side", str(a.side),
                       "--slot-rounds", str(a.slot_rounds), "--outbox-cap", str(a.outbox_cap), "--margin", str(a.margin), "--block-order", a.block_order, "--block-axes", a.block_axes]

                def sub(extra):
                    r = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=600)
                    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                j = sub([])
                out["config"]["at_rest_ms_per_step"] = j["ms_per_step"]
                # ... at rest on compact storage: the workload and storage round 1 quoted its headline on (role-split kernel)
                j2 = sub(["--compact"])
                out["secondary"] = {"at_rest_slotted": {"ms_per_step": j["ms_per_step"], "roofline_frac": j["roofline"]["frac"]},
                                    "at_rest_compact": {"ms_per_step": j2["ms_per_step"], "roofline_frac": j2["roofline"]["frac"],
                                                        "note": "particles at rest, dense binned storage: the r01 headline workload"}}
                # ... and the transfers as separate kernels: the stand-alone P2G is the kernel north_star sets its 0.60 target on
                # (8 untimed steps first: the P2G launch needs ~5 steps to settle -- 1.48, 1.46, 1.44, 1.43, then 1.42 ms on the same box, ZS_BENCH_PER_STEP=1)
                j3 = sub(["--compact", "--unfused", "--warmup", "8", "--steps", "12"])
                r3 = j3["roofline"]
                out["p2g_standalone"] = {"kernel": r3["kernel"], "ms": r3["launch_ms"], "bytes_per_particle": r3["bytes_per_particle"],
                                         "achieved_GBps": r3["achieved"], "frac": r3["frac"], "particles": r3["particles_per_launch"],
                                         "target_frac": 0.60, "steps": j3["steps"], "warmup": j3["warmup"],
                                         "note": "HIP-event time of the P2G launch alone (grid reset and update outside), column at rest, "
                                                 "compact binned storage, cached stress (107 B/particle: SURVEY 8(d))"}
                out["secondary"]["unfused_at_rest"] = {"ms_per_step": j3["ms_per_step"], "p2g_ms": r3["launch_ms"], "p2g_frac": r3["frac"],
                                                       "g2p_ms": r3["g2p"]["launch_ms"],
                                                       "g2p_frac": r3["g2p"]["achieved"] / HBM_PEAK_GBS}
                # what the 0.5x of the stand-alone P2G does and does not contain: (a) the reference-order P2G, constitutive update inside
                # (--no-cache-stress); (b) the unfused step P2G + G2P as a whole, which is where the moved SVD is paid
                j4 = sub(["--compact", "--unfused", "--no-cache-stress", "--warmup", "8", "--steps", "12"])
                r4 = j4["roofline"]
                step_bytes = (r3["bytes_per_particle"] + r3["g2p"]["bytes_per_particle"]) * r3["particles_per_launch"]
                out["p2g_standalone"].update({
                    "reference_order_frac": r4["frac"], "reference_order_ms": r4["launch_ms"],
                    "unfused_step_frac": step_bytes / ((r3["launch_ms"] + r3["g2p"]["launch_ms"]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "unfused_step_ms": r3["launch_ms"] + r3["g2p"]["launch_ms"],
                    "frac_of_measured_read": (r3["achieved"] / out["roofline"]["measured_read"]) if out["roofline"].get("measured_read") else None,
                    "note": "HIP-event time of the P2G launch alone (grid reset and update outside), column at rest, compact binned storage. "
                            "frac: cached stress -- the constitutive update (SVD + return mapping) runs in the tail of the previous G2P, so this is "
                            "P2G without its ALU (107 B/particle: SURVEY 8(d)); reference_order_frac: the same launch with the update inside, as "
                            "P2G.hpp orders it; unfused_step_frac: P2G + G2P launches together (where the moved update is paid); "
                            "frac_of_measured_read: against this box's read-only stream rate instead of the nominal 8 TB/s"})
                # the headline over a long window: steps 100-200 of a 200-step run (fresh storage flatters the first steps)
                js = sub(["--drift", ",".join(str(x) for x in drift_v), "--steps", "100", "--warmup", "100"])
                out["secondary"]["sustained"] = {"ms_per_step": js["ms_per_step"], "roofline_frac": js["roofline"]["frac"], "steps": "100-200 of a 200-step run",
                                                 "repartitions": js["config"].get("repartitions"),
                                                 "mean_rounds": (js.get("slot_stats") or {}).get("mean_rounds")}
                # BASELINE config 3: MLS-MPM elastic jello, 8 M particles, 256^3 sparse grid
                jc = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-at-rest", "--no-cpu-baseline", "--cells", "100,100,100", "--model", "jello",
                                     "--grid", "256"], capture_output=True, text=True, timeout=600)
                jc = json.loads([l for l in jc.stdout.splitlines() if l.startswith("{")][-1])
                out["secondary"]["config3_jello_8M"] = {"ms_per_step": jc["ms_per_step"], "value": jc["value"], "roofline_frac": jc["roofline"]["frac"],
                                                        "workload": jc["config"]["workload"].split(";")[0]}
            except Exception as e:
                out["config"]["at_rest_ms_per_step"] = None
                print("at-rest runs failed: %r" % (e,), file=sys.stderr)
        if world == 1 and not a.no_at_rest:
            # SURVEY 8(d) secondary metrics (BASELINE configs 1, 2 and 5): reduce / exclusive_scan / radix_sort(_pair) at 1 M and 64 M ints,
            # bht build over 16 M random particles, TileVector<f32,32> 25-channel load + store at 16 M, LBvh build / refit / queries /
            # self-collision broadphase at 10 M boxes -- each {ms, units_per_s, frac}
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_prims
                rows = bench_prims.main(only=["prims", "tv", "bht", "lbvh"], sizes=(1_000_000, 64_000_000), quiet=True, tv_cases=((16_000_000, 32, 25),))
                out.setdefault("secondary", {})["prims"] = [
                    {"name": r["name"], "n": r["n"], "ms": r["ms"], "units_per_s": r["units_per_s"],
                     "bytes_per_unit": r["algorithmic_bytes_per_unit"], "frac": r["frac_of_8TBps"]} for r in rows]
            except Exception as e:
                print("secondary primitives failed: %r" % (e,), file=sys.stderr)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample, dx, dt, model, a.side, vol)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
