"""Shared helpers for the parity tests: seeded inputs (xorshift64*) and oracle wrappers."""
import ctypes as C

import numpy as np

SEED0 = 0x9E3779B97F4A7C15


def rng(config_id=0):
    return np.random.Generator(np.random.PCG64(SEED0 ^ config_id))


def xorshift64star(n, seed):
    """The synthetic-input generator named by SURVEY.md 8(d); returns n uint64."""
    out = np.empty(n, dtype=np.uint64)
    x = np.uint64(seed if seed else 1)
    m = np.uint64(0x2545F4914F6CDD1D)
    with np.errstate(over="ignore"):
        for i in range(n):
            x ^= x >> np.uint64(12)
            x ^= x << np.uint64(25)
            x ^= x >> np.uint64(27)
            out[i] = x * m
    return out


def ptr(a, ct=None):
    return a.ctypes.data_as(C.c_void_p)


def orc_call(oracle, name, *args):
    f = getattr(oracle, name)
    f.restype = None
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(a.ctypes.data_as(C.c_void_p))
        elif isinstance(a, (int, np.integer)):
            conv.append(C.c_size_t(int(a)) if int(a) >= 0 else C.c_int(int(a)))
        else:
            conv.append(a)
    return f(*conv)


# ---------------------------------------------------------------------------------------- oracle MPM helpers (oracle/orc.py)
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "oracle"))
from orc import OrcMpmParams, YIELD_SURFACE, OracleMpm  # noqa: E402,F401


def make_cloud(n_side, dx, ppc_side=2, origin=(0.30, 0.31, 0.29), seed=3, noise=0.01, vel_scale=0.5):
    """Jittered lattice of particles: ppc_side^3 particles per cell in an n_side^3-cell cube."""
    g = rng(seed)
    k = n_side * ppc_side
    idx = np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3)
    h = dx / ppc_side
    pos = (np.asarray(origin) + (idx + 0.5) * h + (g.random(idx.shape) - 0.5) * h * 0.8).astype(np.float32)
    n = pos.shape[0]
    vel = (vel_scale * g.standard_normal((n, 3))).astype(np.float32)
    F = (np.eye(3).reshape(1, 9) + noise * g.standard_normal((n, 9))).astype(np.float32)
    Cm = (0.1 * g.standard_normal((n, 9))).astype(np.float32)
    mass = np.full(n, 1000.0 * dx ** 3 / ppc_side ** 3, np.float32)
    return mass, pos, vel, Cm, F


def lbvh_boxes(n, seed, dup=False):
    """n AABBs [n][6] = {min xyz, max xyz}: jittered centres in the unit cube, extents 0.5-3 % (dup: centres snapped to a
    coarse lattice so that many morton codes coincide)."""
    g = rng(seed)
    c = g.uniform(0, 1, (n, 3)).astype(np.float32)
    if dup:
        c = (np.round(c * 6) / 6).astype(np.float32)
    e = g.uniform(0.005, 0.03, (n, 3)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([c - e, c + e], axis=1).astype(np.float32))


def oracle_lbvh(oracle, bv, refit=1):
    import ctypes as C
    n = bv.shape[0]
    oracle.orc_lbvh_create.restype = C.c_void_p
    oracle.orc_lbvh_num_nodes.restype = C.c_size_t
    for f in ("parents", "levels", "leaf_inds", "aux_indices"):
        getattr(oracle, "orc_lbvh_" + f).restype = C.POINTER(C.c_int32)
    oracle.orc_lbvh_bvs.restype = C.POINTER(C.c_float)
    b = C.c_void_p(oracle.orc_lbvh_create())
    oracle.orc_lbvh_build(b, bv.ctypes.data_as(C.c_void_p), C.c_size_t(n), refit)
    nn = oracle.orc_lbvh_num_nodes(b)
    A = lambda p, shape: np.ctypeslib.as_array(p, shape=shape).copy()
    arrs = {"numNodes": nn, "parents": A(oracle.orc_lbvh_parents(b), (nn,)), "levels": A(oracle.orc_lbvh_levels(b), (nn,)),
            "auxIndices": A(oracle.orc_lbvh_aux_indices(b), (nn,)), "leafInds": A(oracle.orc_lbvh_leaf_inds(b), (n,)),
            "bvs": A(oracle.orc_lbvh_bvs(b), (nn, 6))}
    return b, arrs


def collider_struct(cs):
    """row of tests/golden/collider.npz `cases` ([geometry, type, param x8, s, dsdt, R x9, omega x3, b x3, dbdt x3]) ->
    ctypes struct with the layout shared by zs_rocm_collider and orc_collider"""
    import ctypes as C

    class Col(C.Structure):
        _fields_ = [("geometry", C.c_int), ("type", C.c_int), ("param", C.c_float * 8), ("s", C.c_float), ("dsdt", C.c_float),
                    ("R", C.c_float * 9), ("omega", C.c_float * 3), ("b", C.c_float * 3), ("dbdt", C.c_float * 3)]
    return Col(int(cs[0]), int(cs[1]), (C.c_float * 8)(*cs[2:10]), float(cs[10]), float(cs[11]), (C.c_float * 9)(*cs[12:21]),
               (C.c_float * 3)(*cs[21:24]), (C.c_float * 3)(*cs[24:27]), (C.c_float * 3)(*cs[27:30]))


# ---------------------------------------------------------------------------------------- whole-function golden fixtures
GOLDEN_MODELS = {"fixedcorotated": 0, "sand": 1, "vonmises": 2, "nacc": 3, "eos": 4}
GOLDEN_CASES = [("fixedcorotated", 4), ("fixedcorotated", 8), ("sand", 4), ("sand", 8), ("vonmises", 8), ("nacc", 8), ("eos", 8)]


def golden_p2g_g2p(name, side):
    """One case of tests/golden/p2g_g2p.npz (made by tools/gen_golden.py from the reference's own LocalArena / compute_stress_* /
    matrixMatrixMultiplication3d, oracle/ref_shim.cpp): dict with the inputs, the reference's P2G grid, the grid handed to G2P, the
    G2P outputs and the next step's P2G grid, plus the model number / parameters in this repo's convention."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "p2g_g2p.npz"))
    tag = "%s_s%d" % (name, side)
    prm = z["prm_" + name]
    F = z["F"].copy()
    if name == "eos":
        F[:, 0] = z["J"]
    # oracle / zs_rocm model numbering: 0 FixedCorotated, 1 DruckerPrager, 2 VonMises, 3 NACC, 4 EquationOfState
    kw = dict(E=float(prm[1]) if name != "eos" else 5e4, nu=float(prm[2]) if name != "eos" else 0.4)
    if name == "sand":
        kw.update(cohesion=float(prm[3]), beta=float(prm[4]))
    if name == "vonmises":
        kw.update(yield_stress=float(prm[7]))
    if name == "nacc":
        kw.update(beta=float(prm[4]), xi=float(prm[8]), friction_angle=float(prm[9]), hardening=bool(prm[10]))
    if name == "eos":
        kw.update(bulk=float(prm[11]), viscosity=float(prm[12]))
    return dict(model=GOLDEN_MODELS[name], side=side, dx=float(z["dx"]), dt=float(z["dt"]), volume=float(prm[0]), kw=kw,
                gravity=tuple(float(v) for v in z["gravity"]), keys=z["keys_s%d" % side], mass=z["mass"], pos=z["pos"], vel=z["vel"],
                C=z["C"], F=F, logJp=z["logJp"], grid=z["grid_" + tag], logJp1=z["logJp1_" + tag], gridv=z["gridv_" + tag],
                pos1=z["pos_" + tag], vel1=z["vel_" + tag], C1=z["C_" + tag], F1=z["F_" + tag], grid2=z["grid2_" + tag],
                logJp2=z["logJp2_" + tag],
                # physical scale of a nodal force entry for this cloud: the elastic (FixedCorotated) response to the same F set.
                # The NACC case projects every particle onto the tip of the yield surface, where P F^T is the difference of
                # nearly equal numbers (1e-4 of the elastic stress), so "relative to its own maximum" would measure noise.
                rhs_scale=float(np.abs(z["grid_fixedcorotated_s8"][:, 4:]).max()))


def golden_c2(name, side):
    """One case of tests/golden/c2.npz (P2C2GTransfer / G2C2PTransfer bodies over the reference's own pieces, oracle/ref_shim.cpp via
    tools/gen_golden.py): inputs, the reference's P2C2G grid (channels 0..3), a velocity grid and the reference's G2C2P v / B."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2.npz"))
    prm = z["prm_" + name]
    F = z["F"].copy()
    if name == "eos":
        F[:, 0] = z["J"]
    kw = dict(E=float(prm[1]) if name != "eos" else 5e4, nu=float(prm[2]) if name != "eos" else 0.4)
    if name == "vonmises":
        kw.update(yield_stress=float(prm[7]))
    if name == "eos":
        kw.update(bulk=float(prm[11]), viscosity=float(prm[12]))
    return dict(model=GOLDEN_MODELS[name], side=side, dx=float(z["dx"]), dt=float(z["dt"]), volume=float(prm[0]), kw=kw, keys=z["keys_s%d" % side],
                mass=z["mass"], pos=z["pos"], vel=z["vel"], B=z["B"], F=F, grid=z["grid_%s_s%d" % (name, side)], gridv=z["gridv_s%d" % side],
                vel1=z["g2c2p_vel_s%d" % side], B1=z["g2c2p_B_s%d" % side])
