"""GPU parity for the gather-style transfers P2C2G / G2C2P (simulation/transfer/P2C2G.hpp, G2C2P.hpp) vs the CPU oracle.

Tolerances: the HIP path sums a node's 8 cells (and a particle's 8 cells) in a fixed order that differs from the oracle's
cell-major order, and contracts a*b+c: grid channels to rel 2e-4 of the channel magnitude (5e-4 NACC, as for P2G), particle
velocities / B to rel 2e-4, F to 2e-5.  The HIP path itself issues no float atomics: two runs agree bit for bit."""
import numpy as np
import pytest

from util import rng, make_cloud, OracleMpm

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _cmp(ga, gb, rtol, chans):
    assert set(ga.keys()) == set(gb.keys())
    A = np.stack([ga[k] for k in sorted(ga)])
    B = np.stack([gb[k] for k in sorted(ga)])
    for ch in chans:
        s = np.abs(B[:, ch]).max() + 1e-30
        assert np.abs(A[:, ch] - B[:, ch]).max() <= rtol * s, (ch, np.abs(A[:, ch] - B[:, ch]).max() / s)


def _setup(pol, oracle, side, model, seed, aos=False, key_is_origin=False, lane_width=64, dense=False):
    from zpc_amd.mpm import MpmTransfer
    dx, dt = 1.0 / 64, 1e-4
    mass, pos, vel, Bm, F = make_cloud(7, dx, 2, seed=seed, vel_scale=0.3)
    Bm = (Bm * dx * dx * 0.25).astype(np.float32)   # B = C / Dinv ~ C dx^2 / 4 .. dx^2 / 2
    n = pos.shape[0]
    vol = dx ** 3 / 8
    kw = dict(yield_stress=200.0, beta=0.5 if model == 3 else 1.0, bulk=4e4, viscosity=0.01)
    if model == 4:
        F = (1.0 + 0.01 * rng(seed + 1).standard_normal((n, 9))).astype(np.float32)   # J in component 0
    om = OracleMpm(oracle, model, dx, dt, side, vol, **kw)
    om.build_partition(pos, n)
    om.build_buckets(pos)
    mt = MpmTransfer(pol, n, dx, dt, model=model, side=side, volume=vol, lane_width=lane_width, aos=aos, key_is_origin=key_is_origin, **kw)
    lj0 = (0.01 * rng(seed + 2).standard_normal(n)).astype(np.float32)
    mt.upload(mass, pos, vel, Bm, F[:, :1] if model == 4 else F, lj0 if model in (1, 3) else None)
    assert mt.build_partition(n) == om.nblocks
    mt.build_buckets(dense=dense)
    return om, mt, (mass, pos, vel, Bm, F, lj0)


@pytest.mark.parametrize("side", [4, 8])
@pytest.mark.parametrize("model,kind", [(0, 0), (0, 1), (0, 2), (1, 0), (2, 2), (3, 0), (4, 0)])
def test_p2c2g_vs_oracle(pol, oracle, side, model, kind):
    om, mt, (mass, pos, vel, Bm, F, lj0) = _setup(pol, oracle, side, model, seed=70 + side + model)
    lj_o = om.p2c2g(kind, mass, pos, vel, Bm, F, lj0.copy())
    mt.clear_grid()
    mt.p2c2g(kind)
    pol.syncCtx()
    g1 = mt.grid.clone()
    _cmp(mt.grid_by_key(), om.grid_by_key(), 5e-4 if model == 3 else 2e-4, range(4))
    g = mt.grid.cpu().numpy().reshape(mt.nblocks, 7, side ** 3)
    assert not g[:, 4:].any()                                     # channels 4..6 are not touched (P2C2G.hpp:176-186)
    if kind != 2:   # conservation: the linear and the 1/8 weights are partitions of unity
        assert abs(g[:, 0].sum() - mass.sum()) < 1e-4 * mass.sum()
    else:
        assert not g[:, 0].any()
    if kind == 1:   # the affine term carries no net momentum
        mom_p = (mass[:, None] * vel).sum(0)
        assert np.abs(g[:, 1:4].sum(axis=(0, 2)) - mom_p).max() < 2e-3 * np.abs(mass[:, None] * vel).sum()
    if model in (1, 3) and kind != 1:
        assert np.abs(mt.download()["logJp"] - lj_o).max() < (2e-3 if model == 3 else 2e-5)
    # no atomics anywhere: a second run (fresh logJp) reproduces the grid bit for bit
    mt.upload(mass, pos, vel, Bm, F[:, :1] if model == 4 else F, lj0 if model in (1, 3) else None)
    mt.clear_grid()
    mt.p2c2g(kind)
    pol.syncCtx()
    assert torch.equal(g1, mt.grid)


def test_p2c2g_adds_and_splits(pol, oracle):
    """The transfer ADDS into the grid (atomic_add in the reference), and Transfer == Momentum + Force up to rounding."""
    om, mt, (mass, pos, vel, Bm, F, lj0) = _setup(pol, oracle, 4, 0, seed=90)
    mt.clear_grid()
    mt.p2c2g(0)
    pol.syncCtx()
    full = mt.grid.clone()
    mt.clear_grid()
    mt.p2c2g(1)
    mt.p2c2g(2)
    pol.syncCtx()
    s = full.abs().max().item()
    assert (mt.grid - full).abs().max().item() < 1e-5 * s


@pytest.mark.parametrize("variant", ["aos", "origin_keys", "lane32"])
def test_p2c2g_layout_variants(pol, oracle, variant):
    """AoS particle storage, SparseGrid-style block-origin keys and a 32-wide AoSoA give the same bits as the default layout."""
    om, mt, data = _setup(pol, oracle, 8, 0, seed=95)
    _, mt2, _ = _setup(pol, oracle, 8, 0, seed=95, aos=variant == "aos", key_is_origin=variant == "origin_keys",
                       lane_width=32 if variant == "lane32" else 64)
    for m in (mt, mt2):
        m.clear_grid()
        m.p2c2g(0)
    pol.syncCtx()
    ga, gb = mt.grid_by_key(), mt2.grid_by_key()
    if variant == "origin_keys":
        gb = {tuple(k // 8 for k in key): v for key, v in gb.items()}
    assert set(ga) == set(gb)
    for k in ga:
        assert np.array_equal(ga[k], gb[k])


@pytest.mark.parametrize("side", [4, 8])
@pytest.mark.parametrize("model", [0, 4])
def test_g2c2p_vs_oracle(pol, oracle, side, model):
    om, mt, (mass, pos, vel, Bm, F, lj0) = _setup(pol, oracle, side, model, seed=80 + side + model)
    om.p2c2g(0, mass, pos, vel, Bm, F, lj0.copy())
    mt.clear_grid()
    mt.p2c2g(0)
    om.grid_update((0.0, -9.8, 0.0))
    mt.grid_update((0.0, -9.8, 0.0))
    po, vo, Bo, Fo = pos.copy(), vel.copy(), Bm.copy(), F.copy()
    om.g2c2p(po, vo, Bo, Fo)
    buf0 = mt.buf.clone()
    mt.g2c2p(fused=False)
    pol.syncCtx()
    three = mt.buf.clone()
    mt.buf.copy_(buf0)
    mt.g2c2p()          # Pre + G2C2P + Post in one pass: same bits
    pol.syncCtx()
    assert torch.equal(three, mt.buf)
    d = mt.download()
    assert np.abs(d["x"] - po).max() < 1e-6
    assert np.abs(d["v"] - vo).max() < 2e-4 * np.abs(vo).max()
    assert np.abs(d["C"] - Bo).max() < 2e-4 * np.abs(Bo).max() + 1e-9
    if model == 4:
        assert np.abs(d["J"][:, 0] - Fo[:, 0]).max() < 2e-5
    else:
        assert np.abs(d["F"] - Fo).max() < 2e-5


def test_g2c2p_reproduces_affine_field(pol, oracle):
    """Size-independent property: for grid velocities v_i = A x_i + b the transfer returns v_p = A x_p + b and, with
    C = B Dinv, the velocity gradient A itself (the point of the Dinv factor of P2C2G.hpp:88-89)."""
    om, mt, (mass, pos, vel, Bm, F, lj0) = _setup(pol, oracle, 4, 0, seed=99)
    dx, nc = mt.params.dx, 64
    A = np.array([[0.3, -0.2, 0.1], [0.05, 0.4, -0.3], [-0.1, 0.2, 0.25]], np.float32)
    b = np.array([0.5, -0.25, 0.125], np.float32)
    keys = mt.active_keys()
    loc = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3)
    xi = ((keys[:, None, :] * 4 + loc[None]) * dx).astype(np.float32)             # [nb, 64, 3]
    v = xi @ A.T + b
    g = np.zeros((mt.nblocks, 7, nc), np.float32)
    g[:, 1:4] = v.transpose(0, 2, 1)
    mt.grid.copy_(torch.from_numpy(g.reshape(-1)).cuda())
    mt.params.dt = 0.0   # keep x and F: look at v and B only
    mt.g2c2p()
    pol.syncCtx()
    d = mt.download()
    want = pos @ A.T + b
    assert np.abs(d["v"] - want).max() < 2e-5
    r = pos - np.floor(pos / dx + 0.5) * dx
    Dinv = 2.0 / (dx * dx - 2 * r * r)                                              # per axis
    Cm = d["C"].reshape(-1, 3, 3) * Dinv[:, :, None]                                # C[d] *= Dinv[d / 3], column-major: C[r + 3 c]
    grad = Cm.transpose(0, 2, 1)                                                     # [n][r][c]
    assert np.abs(grad - A[None]).max() < 2e-3


@pytest.mark.parametrize("model", [0, 1])
def test_c2_time_loop_vs_oracle(pol, oracle, model):
    """Three full sub-steps (buckets -> P2C2G -> grid update -> G2C2P) on the GPU and in the oracle: the states stay together
    (the buckets are rebuilt in place every step; the partition is fixed, the cloud moves by a fraction of a cell)."""
    om, mt, (mass, pos, vel, Bm, F, lj0) = _setup(pol, oracle, 4, model, seed=120 + model)
    po, vo, Bo, Fo, ljo = pos.copy(), vel.copy(), Bm.copy(), F.copy(), lj0.copy()
    for step in range(3):
        om.build_buckets(po)
        om.grid[:] = 0
        ljo = om.p2c2g(0, mass, po, vo, Bo, Fo, ljo)
        om.grid_update((0.0, -9.8, 0.0))
        om.g2c2p(po, vo, Bo, Fo)
        mt.build_buckets()
        mt.clear_grid()
        mt.p2c2g(0)
        mt.grid_update((0.0, -9.8, 0.0))
        mt.g2c2p()
    pol.syncCtx()
    d = mt.download()
    assert np.abs(d["x"] - po).max() < 2e-6
    assert np.abs(d["v"] - vo).max() < 5e-4 * np.abs(vo).max()
    assert np.abs(d["C"] - Bo).max() < 5e-4 * np.abs(Bo).max()
    assert np.abs(d["F"] - Fo).max() < 5e-5
    if model == 1:
        assert np.abs(d["logJp"] - ljo).max() < 5e-5


def _run_both(pol, oracle, mass, pos, vel, Bm, F, side, displacement=0.0):
    from zpc_amd.mpm import MpmTransfer
    dx, dt = 1.0 / 64, 1e-4
    n = pos.shape[0]
    vol = dx ** 3 / 8
    om = OracleMpm(oracle, 0, dx, dt, side, vol)
    om.build_partition(pos, n)
    om.build_buckets(pos, displacement)
    om.p2c2g(0, mass, pos, vel, Bm, F)
    mt = MpmTransfer(pol, n, dx, dt, model=0, side=side, volume=vol)
    mt.upload(mass, pos, vel, Bm, F)
    assert mt.build_partition(n) == om.nblocks
    mt.build_buckets(displacement)
    mt.clear_grid()
    mt.p2c2g(0)
    pol.syncCtx()
    return om, mt


@pytest.mark.parametrize("side", [4, 8])
def test_p2c2g_crowded_buckets(pol, oracle, side):
    """More than 255 particles in one cell: that bucket is not octant-ordered and is walked whole (only the range check decides)."""
    dx = 1.0 / 64
    mass, pos, vel, Bm, F = make_cloud(4, dx, 2, seed=130, vel_scale=0.3)
    g = rng(131)
    k = 700
    extra = (np.array([0.33, 0.34, 0.32], np.float32) // dx + g.random((k, 3)).astype(np.float32)) * dx   # 700 particles in ONE cell
    pos = np.concatenate([pos, extra.astype(np.float32)])
    n = pos.shape[0]
    mass = np.full(n, mass[0], np.float32)
    vel = (0.3 * g.standard_normal((n, 3))).astype(np.float32)
    Bm = (0.1 * dx * dx * 0.25 * g.standard_normal((n, 9))).astype(np.float32)
    F = (np.eye(3).reshape(1, 9) + 0.01 * g.standard_normal((n, 9))).astype(np.float32)
    om, mt = _run_both(pol, oracle, mass, pos, vel, Bm, F, side)
    v = mt.buckets.view()
    cnt = torch.zeros(v.numBuckets + 1, dtype=torch.int32, device="cuda")
    import ctypes
    ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(v.counts), ctypes.c_size_t(4 * v.numBuckets), 3)
    assert int(cnt.max()) > 255
    _cmp(mt.grid_by_key(), om.grid_by_key(), 2e-4, range(4))
    gsum = mt.grid.cpu().numpy().reshape(mt.nblocks, 7, side ** 3)[:, 0].sum()
    assert abs(gsum - mass.sum()) < 1e-4 * mass.sum()


def test_p2c2g_foreign_buckets(pol, oracle):
    """Buckets that are not the cells of this grid (built with displacement 0.5): no octant shortcut is taken, the 27-bucket walk and
    the range check give what the reference functor gives on such buckets (every particle within dx of a cell centre still lies in
    one of the 27 buckets around the cell, so mass is conserved)."""
    dx = 1.0 / 64
    mass, pos, vel, Bm, F = make_cloud(6, dx, 2, seed=140, vel_scale=0.3)
    Bm = (Bm * dx * dx * 0.25).astype(np.float32)
    om, mt = _run_both(pol, oracle, mass, pos, vel, Bm, F, 4, displacement=0.5)
    _cmp(mt.grid_by_key(), om.grid_by_key(), 2e-4, range(4))
    gsum = mt.grid.cpu().numpy().reshape(mt.nblocks, 7, 64)[:, 0].sum()
    assert abs(gsum - mass.sum()) < 1e-4 * mass.sum()


@pytest.mark.parametrize("side", [4, 8])
@pytest.mark.parametrize("origin_keys", [False, True])
def test_partition_buckets_give_the_same_bits(pol, oracle, side, origin_keys):
    """zs_rocm_index_buckets_for_partition (bucket = block * side^3 + cell, no hash table) vs the hashed IndexBuckets: the same particles
    per cell in the same (ascending id) order, so P2C2G produces the same bits; also through a three-step loop with in-place rebuilds."""
    _, ma, data = _setup(pol, oracle, side, 1, seed=150, key_is_origin=origin_keys)
    _, mb, _ = _setup(pol, oracle, side, 1, seed=150, key_is_origin=origin_keys, dense=True)
    v = mb.buckets.view()
    assert not v.table and v.numBuckets == mb.nblocks * side ** 3 and v.numEntries == mb.n
    for step in range(3):
        for m, dense in ((ma, False), (mb, True)):
            m.build_buckets(dense=dense)
            m.clear_grid()
            m.p2c2g(0)
        pol.syncCtx()
        ga, gb = ma.grid_by_key(), mb.grid_by_key()
        assert set(ga) == set(gb)
        for k in ga:
            assert np.array_equal(ga[k], gb[k]), (step, k)
        for m in (ma, mb):
            m.grid_update((0.0, -9.8, 0.0))
            m.g2c2p()
    pol.syncCtx()
    assert torch.equal(ma.buf, mb.buf)


def test_partition_buckets_with_unlisted_particles(pol, oracle):
    """Particles whose cell is not in the partition sit in the extra bucket no transfer visits: the grid gets exactly what the listed
    particles give, nothing is written out of bounds, and G2C2P leaves the unlisted particles' v, B at zero."""
    from zpc_amd.mpm import MpmTransfer
    dx, dt = 1.0 / 64, 1e-4
    mass, pos, vel, Bm, F = make_cloud(5, dx, 2, seed=160, vel_scale=0.3)
    Bm = (Bm * dx * dx * 0.25).astype(np.float32)
    n0 = pos.shape[0]
    vol = dx ** 3 / 8
    ref = MpmTransfer(pol, n0, dx, dt, model=0, side=4, volume=vol)
    ref.upload(mass, pos, vel, Bm, F)
    ref.build_partition(n0)
    ref.build_buckets(dense=True)
    ref.clear_grid()
    ref.p2c2g(0)
    # the same cloud + 300 particles far outside the partition (> 255 of them in one cell: the extra bucket is not octant-ordered)
    g = rng(161)
    far = (np.array([0.9, 0.9, 0.9], np.float32) + g.random((300, 3)).astype(np.float32) * dx * 0.9).astype(np.float32)
    pos2 = np.concatenate([pos, far])
    n = pos2.shape[0]
    cat = lambda a, w: np.concatenate([a, np.tile(a[:1], (300,) + (1,) * (a.ndim - 1))]).astype(np.float32)
    mt = MpmTransfer(pol, n, dx, dt, model=0, side=4, volume=vol)
    mt.upload(cat(mass, 1), pos2, cat(vel, 3), cat(Bm, 9), cat(F, 9))
    mt.table, mt.nblocks, mt.nbr = ref.table, ref.nblocks, ref.nbr            # the partition of the first cloud only
    mt.grid = torch.zeros_like(ref.grid)
    mt.build_buckets(dense=True)
    v = mt.buckets.view()
    assert v.numEntries == n and v.numBuckets == ref.nblocks * 64
    mt.p2c2g(0)
    pol.syncCtx()
    assert torch.equal(mt.grid, ref.grid)
    mt.grid_update((0.0, -9.8, 0.0))
    mt.g2c2p()
    pol.syncCtx()
    d = mt.download()
    assert not d["v"][n0:].any() and not d["C"][n0:].any() and np.abs(d["v"][:n0]).max() > 0
    assert zs_no_error()


def zs_no_error():
    import zpc_amd
    return zpc_amd.lib().zs_rocm_last_error(-1) == 0


@pytest.mark.parametrize("name,side", [("fixedcorotated", 4), ("fixedcorotated", 8), ("eos", 4), ("eos", 8)])
def test_p2c2g_g2c2p_match_reference_golden(pol, name, side):
    """zs_rocm_mpm_p2c2g / zs_rocm_mpm_g2c2p_step against tests/golden/c2.npz: P2C2GTransfer / G2C2PTransfer bodies spelled over the
    reference's own headers (oracle/ref_shim.cpp, tools/gen_golden.py).  Grid channels to 2e-4 of their magnitude (summation order, a*b+c
    contraction), v and B likewise; the partition is the fixture's, block for block.  (The fixture's von Mises case comes from the HOST
    constitutive header and is compared with the oracle's host variant on the CPU; the GPU follows the CUDA header, which differs where the
    material yields -- as for P2G, tests/test_mpm_gpu.py RHS_TOL.)"""
    from util import golden_c2
    from zpc_amd.mpm import MpmTransfer
    g = golden_c2(name, side)
    n = g["pos"].shape[0]
    mt = MpmTransfer(pol, n, g["dx"], g["dt"], model=g["model"], side=side, volume=g["volume"], **g["kw"])
    mt.upload(g["mass"], g["pos"], g["vel"], g["B"], g["F"][:, :mt.nF], None)
    keys = torch.from_numpy(np.ascontiguousarray(g["keys"])).cuda()
    mt.adopt_partition(keys.data_ptr(), g["keys"].shape[0])
    mt.build_buckets()
    mt.clear_grid()
    mt.p2c2g(0)
    pol.syncCtx()
    got = mt.grid.cpu().numpy().reshape(g["keys"].shape[0], 7, side ** 3)
    scale = np.abs(g["grid"]).max(axis=(0, 2))
    err = np.abs(got[:, :4] - g["grid"]).max(axis=(0, 2)) / scale
    assert (err <= 2e-4).all() and not got[:, 4:].any(), err
    # G2C2P on the fixture's velocity grid
    gv = np.zeros_like(got)
    gv[:, 1:4] = g["gridv"]
    mt.grid.copy_(torch.from_numpy(gv.reshape(-1)).cuda())
    mt.g2c2p()
    pol.syncCtx()
    d = mt.download()
    assert np.abs(d["v"] - g["vel1"]).max() <= 2e-4 * np.abs(g["vel1"]).max()
    assert np.abs(d["C"] - g["B1"]).max() <= 2e-4 * np.abs(g["B1"]).max()
