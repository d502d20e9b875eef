"""Multi-rank path on ONE GPU: bench.py with 2 and 4 ranks sharing cuda:0 (gloo transport, halo buffers staged through
host memory) must reproduce the single-rank particle state after several steps -- this exercises the domain
decomposition, per-rank partitions, shared-block discovery, halo pack / exchange / unpack-add and the strong-scaling
bookkeeping; only the RCCL transport itself is not covered (no multi-GPU box in the test pool)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the compact storage of round 1 (re-bin controller, overlapped exchange); the slotted default has its own test below
# (--lift 8 everywhere: two runs are compared, and on y = 0 the reference arena's rounding case -- profiles/r03_compact_outliers.md -- can
# hit one run and not the other)
ARGS = ["--cells", "24,48,24", "--steps", "4", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--checksum", "--compact", "--drift", "0,0,0"]


def _run(n, ARGS=ARGS):
    if "--cells" in ARGS[2:]:  # a later --cells overrides the default one
        ARGS = ARGS[2:]
    env = dict(os.environ)
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--same-device"] + ARGS
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("n,extra", [(2, []), (4, ["--decomp", "2x2x1"]), (8, ["--decomp", "2x2x2"]), (8, ["--cells", "24,128,24"]),
                                     (2, ["--no-overlap"]), (4, ["--side", "4", "--cells", "24,64,24"]),
                                     (2, ["--cells", "16,192,16", "--side", "4"]), (2, ["--cells", "16,192,16"])])
def test_n_ranks_reproduce_single_rank(n, extra):
    """default N > 1 step: boundary blocks first, ghost exchange on a second stream overlapped with the interior blocks"""
    ref = _run(1, ARGS + extra)
    out = _run(n, ARGS + extra)
    assert out["n_gpus"] == n and out["config"]["particles"] == ref["config"]["particles"]
    assert out["config"]["halo_bytes_per_step_rank0"] > 0
    assert out["config"]["halo_overlap"] == ("--no-overlap" not in extra)
    if out["config"]["halo_overlap"]:
        assert 0 < out["config"]["boundary_blocks_rank0"] <= out["config"]["grid_blocks_rank0"]
    if "16,192,16" in extra:  # tall column: most blocks are interior, so the split launch + second stream really run
        assert out["config"]["boundary_blocks_rank0"] < out["config"]["grid_blocks_rank0"] // 2
    a, b = np.array(ref["checksum"]), np.array(out["checksum"])
    # sums and sums of squares of every particle channel (m, x, v, C, F, logJp) after 4 steps; particles are generated
    # from their global id, so every decomposition starts from the same state and must reach the same state up to float
    # summation order in P2G (halo partial sums are added in a different order)
    nch = len(a) // 2
    npart = ref["config"]["particles"]
    # channel sums (they cancel for v, C: scale by the RMS magnitude), channel sums of squares (relative)
    scale = np.sqrt(npart * np.maximum(a[nch:], 1e-30))
    assert (np.abs(a[:nch] - b[:nch]) <= 1e-6 * scale + 1e-12).all(), np.abs(a[:nch] - b[:nch]) / scale
    assert (np.abs(a[nch:] - b[nch:]) <= 1e-5 * np.abs(a[nch:]) + 1e-12).all()
    assert out["hip_error"] == 0


def test_eight_ranks_2x2x2_match_single_rank_particle_by_particle(tmp_path):
    """2 x 2 x 2 decomposition of the default (slotted, moving) step against the single-rank run, PER PARTICLE: every particle carries its
    number in its mass (--tag-mass), every rank dumps its particles (--dump-state), and the eight dumps together must be the single-rank
    state -- same particles (nobody lost, nobody twice), same position / velocity / deformation up to the summation order of the float
    atomics.  (Channel sums, which the other tests compare, would let a permutation or two cancelling errors pass.)"""
    args = ["--cells", "32,48,32", "--steps", "6", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--tag-mass", "--no-at-rest"]
    _run(1, args + ["--dump-state", str(tmp_path / "one")])
    _run(8, args + ["--decomp", "2x2x2", "--dump-state", str(tmp_path / "eight")])
    ref = dict(np.load(str(tmp_path / "one") + ".rank0.npz"))
    parts = [dict(np.load(str(tmp_path / "eight") + ".rank%d.npz" % r)) for r in range(8)]
    assert all(p["m"].shape[0] > 0 for p in parts)  # every rank owns a share
    got = {k: np.concatenate([p[k] for p in parts]) for k in ref}
    assert got["m"].shape[0] == ref["m"].shape[0]
    o1, o8 = np.argsort(ref["m"], kind="stable"), np.argsort(got["m"], kind="stable")
    assert np.array_equal(ref["m"][o1], got["m"][o8]) and np.unique(ref["m"]).shape[0] == ref["m"].shape[0]   # the same set of distinct particles
    vs = np.abs(ref["v"]).max()
    for k, tol in (("x", 2e-6), ("v", 2e-4 * vs), ("F", 2e-5), ("logJp", 2e-5)):
        if k in ref:
            assert np.abs(ref[k][o1] - got[k][o8]).max() <= tol, (k, np.abs(ref[k][o1] - got[k][o8]).max())


@pytest.mark.parametrize("n,extra", [(2, []), (4, []), (2, ["--unfused"])])
def test_particle_migration_between_ranks(n, extra):
    """The column drifts upwards at 0.35 cell per step; every 3 steps particles are handed to the rank that now owns their
    cell (owner classification, radix partition by destination, all_to_all of AoS rows, AoSoA rebuild), partitions, halo
    lists and bins are rebuilt.  After 9 steps the N-rank state equals the single-rank state, which only re-partitions."""
    args = ["--cells", "24,48,24", "--steps", "9", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--checksum", "--compact", "--drift", "0,7,0",
            "--migrate-every", "3"] + extra
    ref = _run(1, args)
    out = _run(n, args)
    assert out["n_gpus"] == n and out["config"]["particles"] == ref["config"]["particles"]  # nothing lost, nothing duplicated
    assert out["config"]["migrated_rank0"] > 0
    a, b = np.array(ref["checksum"]), np.array(out["checksum"])
    nch = len(a) // 2
    npart = ref["config"]["particles"]
    scale = np.sqrt(npart * np.maximum(a[nch:], 1e-30))
    # 9 steps with three re-partitions: P2G sums are formed in a different order on every decomposition (observed 3e-6 / 1.5e-5)
    assert (np.abs(a[:nch] - b[:nch]) <= 2e-5 * scale + 1e-12).all(), np.abs(a[:nch] - b[:nch]) / scale
    assert (np.abs(a[nch:] - b[nch:]) <= 1e-4 * np.abs(a[nch:]) + 1e-12).all()
    assert out["hip_error"] == 0 and ref["hip_error"] == 0


def test_overlap_drift_guard_refuses_stale_bins():
    """Without re-binning, particles flying 3 cells per step leave the margin that keeps interior blocks away from the shared
    blocks: the overlapped step must refuse the run (device drift flag of zs_rocm_mpm_g2p2g_range) instead of silently
    dropping ghost contributions.  (A run like this needs --migrate-every anyway: the particles also leave the partition.)"""
    args = ["--cells", "16,192,16", "--side", "4", "--steps", "4", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--checksum", "--compact", "--drift", "0,60,0"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device"] + args
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0 and b"drifted more than one bin" in r.stderr


def test_rccl_calls_of_the_multi_gpu_path_on_one_gpu():
    """tools/nccl_selftest.py: the torch.distributed calls of the halo exchange / migration / bench bookkeeping on the nccl (= RCCL)
    backend with world_size 1 and self as the only peer, the exchange issued on the side stream exactly as bench.py does."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_selftest.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"nccl selftest ok" in r.stdout, r.stderr.decode()[-2000:]


def test_long_run_with_rebins_is_rank_independent():
    """40 steps of a drifting tall column (0.05 cell per step, 2 cells in total -- inside the partition's margin, so no
    re-partition is needed: particles keep changing cells, the re-bin controller of bench.py fires at rank-dependent moments) on 1
    and on 2 ranks with the overlapped exchange: same final particle state."""
    args = ["--cells", "16,192,16", "--side", "4", "--steps", "40", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--checksum", "--compact", "--drift", "0,1,0",
            "--rebin-check", "2"]
    ref = _run(1, args)
    out = _run(2, args)
    assert out["config"]["particles"] == ref["config"]["particles"] and out["hip_error"] == 0 and ref["hip_error"] == 0
    a, b = np.array(ref["checksum"]), np.array(out["checksum"])
    nch = len(a) // 2
    scale = np.sqrt(ref["config"]["particles"] * np.maximum(a[nch:], 1e-30))
    assert (np.abs(a[:nch] - b[:nch]) <= 5e-5 * scale + 1e-12).all(), np.abs(a[:nch] - b[:nch]) / scale
    assert (np.abs(a[nch:] - b[nch:]) <= 2e-4 * np.abs(a[nch:]) + 1e-12).all()


@pytest.mark.parametrize("n,extra", [(2, []), (4, ["--decomp", "2x2x1"]), (2, ["--side", "4", "--cells", "16,96,16"]),
                                     (2, ["--drift", "3,-4,2", "--outbox-cap", "512"]), (2, ["--no-overlap"]),
                                     # 8 ranks in the default 2 x 2 x 2 grid, the cloud drifting obliquely through the corner all eight
                                     # boxes share: movers cross faces, edges and the corner between ranks
                                     (8, ["--cells", "32,64,32", "--drift", "3,-4,2", "--outbox-cap", "512"]),
                                     (8, ["--cells", "24,128,24", "--decomp", "1x8x1"])])
def test_slotted_default_n_ranks_reproduce_single_rank(n, extra):
    """the bench's default storage (slotted: the step keeps its own order, movers travel through outboxes) on N ranks sharing one GPU:
    the column falls at 0.05 cell per step (default) or drifts obliquely at up to 0.2 cell per step across the rank boundaries, no
    re-bin and no re-partition anywhere; after 12 steps the N-rank particle state equals the single-rank one."""
    args = ["--cells", "24,48,24", "--steps", "12", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--checksum", "--no-at-rest"] + extra
    ref = _run(1, args)
    out = _run(n, args)
    assert out["n_gpus"] == n and out["config"]["particles"] == ref["config"]["particles"]
    assert "slotted" in out["config"]["storage"] and out["config"]["rebins"] == 0
    assert ref["config"]["movers_per_step_rank0"] > 0
    # the exchange of the ghost-block sums is overlapped with the interior blocks on slotted storage too (boundary blocks first)
    assert out["config"]["halo_overlap"] == ("--no-overlap" not in extra) and out["config"]["halo_bytes_per_step_rank0"] > 0
    if n == 8 and "--decomp" not in extra:
        assert out["config"]["decomposition"] == "2x2x2"
    a, b = np.array(ref["checksum"]), np.array(out["checksum"])
    nch = len(a) // 2
    scale = np.sqrt(ref["config"]["particles"] * np.maximum(a[nch:], 1e-30))
    assert (np.abs(a[:nch] - b[:nch]) <= 2e-5 * scale + 1e-12).all(), np.abs(a[:nch] - b[:nch]) / scale
    assert (np.abs(a[nch:] - b[nch:]) <= 1e-4 * np.abs(a[nch:]) + 1e-12).all()
    assert out["hip_error"] == 0 and ref["hip_error"] == 0


@pytest.mark.parametrize("extra", [[], ["--drift", "3,-4,2", "--outbox-cap", "512"]])
def test_range_schedules_of_the_step_call_agree(extra):
    """zs_rocm_mpm_step.rangeSchedule: the boundary blocks' range and the interior range in turn | side by side on two streams | ONE launch
    whose boundary workgroups count themselves off for a gate kernel on the exchange stream (the default for 8^3 blocks).  On one GPU
    through bench.py's rank proxy (the 8-rank schedule with RCCL at world size 1, exchange into a scratch grid): the same particle state
    after 12 steps as the plain single-rank step, movers crossing between the two ranges included."""
    args = ["--cells", "24,48,24", "--steps", "12", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--checksum", "--no-at-rest"] + extra
    ref = _run(1, args)
    assert ref["config"]["movers_per_step_rank0"] > 0 and ref["config"]["step_schedule"] == "one range"
    for sched, name in (("in-turn", "ranges in turn"), ("side-by-side", "ranges side by side"), ("one-launch", "one launch + gate"), ("auto", "one launch + gate")):
        out = _run(1, args + ["--rank-proxy", "8", "--range-schedule", sched, "--check-handover"])
        assert out["hip_error"] == 0 and out["config"]["halo_overlap"] and 0 < out["config"]["boundary_blocks_rank0"] < out["config"]["grid_blocks_rank0"]
        assert out["config"]["step_schedule"] == name
        # the hand-over itself: what the exchange stream saw of the shared blocks' mass sums when the exchange started equals, bit for bit, the
        # same blocks after the step (zs_rocm_mpm_step.handoverSnapshot) -- in every step: the exchange never runs ahead of a boundary block
        ho = out["config"]["handover"]
        assert ho["steps"] >= 12 and ho["mismatched_steps"] == 0 and ho["nonzero"] > 1000, (sched, ho)
        a, b = np.array(ref["checksum"]), np.array(out["checksum"])
        nch = len(a) // 2
        scale = np.sqrt(ref["config"]["particles"] * np.maximum(a[nch:], 1e-30))
        assert (np.abs(a[:nch] - b[:nch]) <= 2e-5 * scale + 1e-12).all(), (sched, np.abs(a[:nch] - b[:nch]) / scale)
        assert (np.abs(a[nch:] - b[nch:]) <= 1e-4 * np.abs(a[nch:]) + 1e-12).all(), sched
        bdn = out["rank_breakdown"]["max_over_ranks"]
        assert bdn["both_ranges_ms"] > 0 and bdn["boundary_range_ms"] > 0 and bdn["exchange_side_stream_ms"] > 0
    if not extra:
        # ... and on a box of 1024 blocks with particles (two rounds of the chip's 512 workgroup slots: boundary blocks really finish at different
        # times), with the negative control: told that only a tenth of the boundary blocks is boundary, a schedule releases the exchange too
        # early -- and the check sees it
        big = ["--cells", "64,128,64", "--steps", "6", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--no-at-rest", "--rank-proxy", "8", "--check-handover"]
        for sched in ("in-turn", "one-launch"):
            good = _run(1, big + ["--range-schedule", sched])["config"]["handover"]
            assert good["steps"] >= 6 and good["mismatched_steps"] == 0 and good["nonzero"] > 100000, (sched, good)
            bad = _run(1, big + ["--range-schedule", sched, "--understate-boundary", "0.1"])["config"]["handover"]
            assert bad["steps"] >= 6 and bad["mismatched_steps"] > 0, (sched, bad)


def test_native_rccl_exchange_steps_on_one_gpu():
    """tools/rccl_native_selftest.py: zs_rocm_dist_* (RCCL called from libzsrocm.so, no torch.distributed anywhere) with world size 1
    and this rank as its own peer -- communicator, allreduce sum / max / min, counts all-to-all, uneven all-to-all, ghost-block
    exchange (one and two messages, null stream and side stream), barrier."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_native_selftest.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"rccl native selftest ok" in r.stdout, (r.stdout.decode()[-1000:], r.stderr.decode()[-3000:])


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("extra", [[], ["--compact", "--drift", "0,0,0"], ["--comm", "torch"],
                                   ["--compact", "--drift", "0,7,0", "--migrate-every", "3", "--steps", "9"]])
def test_rccl_backend_on_all_visible_gpus_reproduces_single_rank(extra):
    """The production transport: one rank per visible GPU (self-launching `bench.py --gpus N`), ghost sums / CFL allreduce / migration
    over RCCL from libzsrocm.so (or torch.distributed's nccl backend with --comm torch).  Needs >= 2 GPUs; the 1-GPU test boxes skip
    it and the multi-rank logic is covered by the gloo runs above."""
    n = _visible_gpus()
    if n < 2:
        pytest.skip("needs >= 2 visible GPUs (found %d)" % n)
    n = 8 if n >= 8 else (4 if n >= 4 else 2)
    args = ["--cells", "24,48,24", "--steps", "6", "--warmup", "0", "--no-cpu-baseline", "--lift", "8", "--checksum", "--no-at-rest"] + extra
    ref = _run(1, args)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == n and out["config"]["particles"] == ref["config"]["particles"] and out["hip_error"] == 0
    assert ("rccl via libzsrocm" in out["config"]["exchange"]) == ("--comm" not in extra)
    a, b = np.array(ref["checksum"]), np.array(out["checksum"])
    nch = len(a) // 2
    scale = np.sqrt(ref["config"]["particles"] * np.maximum(a[nch:], 1e-30))
    assert (np.abs(a[:nch] - b[:nch]) <= 2e-5 * scale + 1e-12).all(), np.abs(a[:nch] - b[:nch]) / scale
    assert (np.abs(a[nch:] - b[nch:]) <= 1e-4 * np.abs(a[nch:]) + 1e-12).all()
    # the CFL bound is a global maximum: every decomposition reports the same value up to summation order in P2G
    assert abs(out["config"]["cfl_max_vel_sqr"] - ref["config"]["cfl_max_vel_sqr"]) <= 1e-4 * ref["config"]["cfl_max_vel_sqr"] + 1e-12
