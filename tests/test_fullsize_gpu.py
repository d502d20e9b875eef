"""Size-independent properties at the sizes BASELINE.json names (the parity tests proper run at sizes the oracle finishes in seconds):
config 1 primitives at 1 M ints against numpy; config 2 bht at 16 M keys; config 3 the 8 M-particle jello cube; config 4 the 64 Mi-particle
sand column (fused slotted == fused compact == unfused, nothing lost, particle number conserved); config 5 LBvh at 10 M boxes."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


CHANNELS = (["m"] + ["x%d" % k for k in range(3)] + ["v%d" % k for k in range(3)] + ["C%d" % k for k in range(9)]
            + ["F%d" % k for k in range(9)] + ["logJp"] + ["PFt%d" % k for k in range(9)])


def _same_state(a, b, npart, tol_sum, tol_sq):
    """channel sums and sums of squares of two runs; the failure message names the channels and by how much"""
    a, b = np.array(a), np.array(b)
    nch = len(a) // 2
    names = CHANNELS if nch >= 35 else [c for c in CHANNELS if c != "logJp"]
    scale = np.sqrt(npart * np.maximum(a[nch:], 1e-30))
    r_sum = np.abs(a[:nch] - b[:nch]) / (tol_sum * scale + 1e-12)
    r_sq = np.abs(a[nch:] - b[nch:]) / (tol_sq * np.abs(a[nch:]) + 1e-12)
    bad = [("sum " + names[k], float(r_sum[k]), float(a[k]), float(b[k])) for k in range(nch) if r_sum[k] > 1] + \
          [("sq " + names[k], float(r_sq[k]), float(a[nch + k]), float(b[nch + k])) for k in range(nch) if r_sq[k] > 1]
    assert not bad, "channels outside tolerance (name, |d|/tol, a, b): %r" % (bad,)


def test_config1_primitives_at_1m_ints(pol):
    """BASELINE config 1: reduce / exclusive_scan / radix_sort on 1 M ints, bit-exact against numpy"""
    import zpc_amd as zs
    a = np.random.default_rng(11).integers(-2 ** 31, 2 ** 31 - 1, 1 << 20, dtype=np.int64).astype(np.int32)
    d = torch.from_numpy(a).cuda()
    out = torch.empty_like(d)
    r = torch.zeros(1, dtype=torch.int32, device="cuda")
    zs.reduce(pol, d, None, r)
    assert int(r.item()) == int(np.sum(a, dtype=np.int64).astype(np.int32))
    zs.exclusive_scan(pol, d, out)
    assert np.array_equal(out.cpu().numpy(), (np.cumsum(a, dtype=np.int64) - a).astype(np.int32))
    zs.radix_sort(pol, d, out)
    assert np.array_equal(out.cpu().numpy(), np.sort(a))


def test_config2_bht_16m_keys(pol, oracle):
    """BASELINE config 2: bht build over 16 M random cells (10.6 M distinct).  With the reference's sizing (2 next_2pow(n) slots in
    buckets of 16, 15 usable) and its universal hash, a few dozen keys find all three of their buckets full: the reference's insert
    returns its failure token for them and clears `success` (Bht.hpp:536-541; the caller is expected to resize and rebuild).  The
    sequential oracle loses 73 keys on this input, the GPU build -- another insertion order -- about as many.  Checked: every key is
    either stored or reported; stored keys get one dense index each, queries return it, activeKeys[index] == key."""
    from zpc_amd.containers import Bht
    n = 1 << 24
    g = np.random.default_rng(12)
    keys = g.integers(0, 256, (n, 3), dtype=np.int32)
    packed = (keys[:, 0].astype(np.int64) << 16) | (keys[:, 1].astype(np.int64) << 8) | keys[:, 2]
    ndist = np.unique(packed).shape[0]
    oracle.orc_bht_create_b.restype = C.c_void_p
    oracle.orc_bht_size.restype = C.c_int32
    ot = C.c_void_p(oracle.orc_bht_create_b(3, C.c_size_t(n), 16))
    oracle.orc_bht_insert_many(ot, keys.ctypes.data_as(C.c_void_p), C.c_size_t(n), None)
    olost = ndist - oracle.orc_bht_size(ot)
    oracle.orc_bht_destroy(ot)
    tab = Bht(3, n)
    dk = torch.from_numpy(keys).cuda()
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert(pol, dk.data_ptr(), n, ret.data_ptr())
    q = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.query(pol, dk.data_ptr(), n, q.data_ptr())
    pol.syncCtx()
    qi, r = q.cpu().numpy(), ret.cpu().numpy()
    size = tab.size()
    lost = ndist - size
    assert 0 <= lost <= 2 * olost + 16, (lost, olost)
    v = tab.view()
    succ = np.empty(1, np.int32)
    C.CDLL("libamdhip64.so").hipMemcpy(succ.ctypes.data_as(C.c_void_p), C.c_void_p(v.success), C.c_size_t(4), 2)
    assert (succ[0] == 0) == (lost > 0)
    failed = r < -1                                                   # failure_token_v: reported, never silent
    assert np.unique(packed[failed]).shape[0] == lost and (qi[failed] < 0).all()
    stored = qi >= 0
    assert stored.sum() == n - np.isin(packed, np.unique(packed[failed])).sum()
    assert qi[stored].max() == size - 1 and (r >= 0).sum() == size   # one winner per stored key
    o = np.argsort(packed, kind="stable")                             # same key <-> same index
    same_key = packed[o][1:] == packed[o][:-1]
    both = stored[o][1:] & stored[o][:-1]
    assert np.array_equal(same_key[both], (qi[o][1:] == qi[o][:-1])[both])
    act = np.empty(size * 3, np.int32)
    C.CDLL("libamdhip64.so").hipMemcpy(act.ctypes.data_as(C.c_void_p), C.c_void_p(v.activeKeys), C.c_size_t(act.nbytes), 2)
    sample = np.nonzero(stored)[0][::83]
    assert np.array_equal(act.reshape(size, 3)[qi[sample]], keys[sample])


def test_config5_lbvh_10m_boxes(pol):
    """BASELINE config 5: LBvh over 10 M boxes: structural invariants of the pre-order layout (parents, levels, escape indices, leaf
    positions, boxes contain their children) and the self-collision pairs of sampled leaves against brute force"""
    from zpc_amd.containers import LBvh
    n = 10_000_000
    g = np.random.default_rng(15)
    c = g.uniform(0, 1, (n, 3)).astype(np.float32)
    e = g.uniform(0.0005, 0.002, (n, 3)).astype(np.float32)
    bv = np.ascontiguousarray(np.concatenate([c - e, c + e], axis=1))
    d = torch.from_numpy(bv).cuda()
    bvh = LBvh()
    bvh.build(pol, d)
    pol.syncCtx()
    v = bvh.view()
    nn = v.numNodes
    assert nn == 2 * n - 1

    def d2h(p, cnt, dt=np.int32):
        out = np.empty(cnt, dt)
        C.CDLL("libamdhip64.so").hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(out.nbytes), 2)
        return out
    par, lev, aux, lin = d2h(v.parents, nn), d2h(v.levels, nn), d2h(v.auxIndices, nn), d2h(v.leafInds, n)
    boxes = d2h(v.orderedBvs, nn * 6, np.float32).reshape(nn, 6)
    assert par[0] == -1 and (par[1:] >= 0).all() and (par[1:] < np.arange(1, nn)).all()       # pre-order: a parent precedes its children
    leaf = lev == 0
    assert leaf.sum() == n and np.array_equal(np.sort(aux[leaf]), np.arange(n))                 # every primitive is one leaf
    assert np.array_equal(np.sort(lin), np.nonzero(leaf)[0])
    trunk = ~leaf
    esc = aux[trunk]
    pos = np.nonzero(trunk)[0]
    assert ((esc == -1) | (esc > pos)).all()                                                    # escape indices point forward
    # a trunk node's box contains the box of the node right behind it (its left child) and of its parent's view of it
    lc = pos + 1
    assert (boxes[pos, :3] <= boxes[lc, :3]).all() and (boxes[pos, 3:] >= boxes[lc, 3:]).all()
    assert np.array_equal(boxes[lin], bv[aux[lin]])                                             # leaf boxes are the primitives' boxes
    # self-collision pairs of 300 sampled primitives vs brute force
    offs, pairs = bvh.self_query(pol)
    pol.syncCtx()
    p = pairs.cpu().numpy()
    sample = g.integers(0, n, 300)
    for s in sample:
        ov = ((bv[:, :3] <= bv[s, 3:]) & (bv[:, 3:] >= bv[s, :3])).all(1)
        want = set(np.nonzero(ov)[0].tolist()) - {int(s)}
        got = set(p[p[:, 0] == s][:, 1].tolist()) | set(p[p[:, 1] == s][:, 0].tolist())
        assert got == want, (s, len(got), len(want))


def test_config3_jello_8m_fused_equals_unfused():
    """BASELINE config 3 (8 M-particle FixedCorotated cube on the 256^3 grid): three steps, slotted fused == compact fused == unfused"""
    base = ["--cells", "100,100,100", "--model", "jello", "--grid", "256", "--steps", "3", "--warmup", "0", "--no-cpu-baseline", "--checksum",
            "--no-at-rest", "--drift", "0,-0.5,0"]
    a = _bench(base)
    b = _bench(base + ["--compact"])
    # (with --checksum the fused variants run one more step that materialises v, C and the stress of every particle)
    c = _bench([x if x != "3" else "4" for x in base] + ["--compact", "--unfused"])
    assert a["config"]["particles"] == 8_000_000 and a["hip_error"] == 0 and "slotted" in a["config"]["storage"]
    _same_state(a["checksum"], b["checksum"], 8_000_000, 2e-5, 1e-4)
    _same_state(a["checksum"], c["checksum"], 8_000_000, 2e-5, 1e-4)


def test_config4_sand_64m_fused_equals_unfused():
    """BASELINE config 4 at N = 1 (64 Mi-particle DruckerPrager column, dx = 1/512) falling at 0.05 cell per step: after three steps the
    slotted fused step, the compact fused step and the unfused P2G / G2P kernels hold the same particle state (channel sums and sums of
    squares); the slotted run delivered every mover it sent and lost no particle"""
    base = ["--steps", "3", "--warmup", "0", "--no-cpu-baseline", "--checksum", "--no-at-rest"]
    a = _bench(base)
    b = _bench(base + ["--compact"])
    c = _bench([x if x != "3" else "4" for x in base] + ["--compact", "--unfused"])
    n = 67_108_864
    assert a["config"]["particles"] == n and a["hip_error"] == 0 and a["config"]["movers_per_step_rank0"] > 1e6
    _same_state(a["checksum"], b["checksum"], n, 2e-5, 1e-4)
    _same_state(a["checksum"], c["checksum"], n, 2e-5, 1e-4)
    # mass channel: sum == n * m exactly in float64 (every particle still there, once)
    m = 1000.0 * (1.0 / 512) ** 3 / 8
    assert abs(a["checksum"][0] - n * np.float32(m)) <= 1e-9 * n * m


def test_config4_sand_64m_slotted_24_moving_steps_equal_compact_with_rebins():
    """24 steps of the falling 64 Mi-particle column (1.3 cells of travel: a third of the particles change cell, every bin's rounds
    have become uneven): the slotted step -- packed producers, movers through the outboxes, no re-bin -- against the compact storage
    with the re-bin controller; same particle state, every mover delivered, nobody lost."""
    base = ["--steps", "24", "--warmup", "0", "--no-cpu-baseline", "--checksum", "--no-at-rest"]
    a = _bench(base + ["--slot-stats"])
    b = _bench(base + ["--compact", "--rebin-check", "2"])
    n = 67_108_864
    assert a["config"]["particles"] == n and a["hip_error"] == 0 and b["hip_error"] == 0
    assert a["config"]["rebins"] == 0 and b["config"]["rebins"] > 0
    assert a["slot_stats"]["mean_rounds"] > a["slot_stats"]["mean_particles_per_bin_div64"] + 1.0  # the regime the packing is for
    _same_state(a["checksum"], b["checksum"], n, 1e-4, 3e-4)
    m = 1000.0 * (1.0 / 512) ** 3 / 8
    assert abs(a["checksum"][0] - n * np.float32(m)) <= 1e-9 * n * m
