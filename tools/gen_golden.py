#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own code compiled in place (oracle/_ref/libzpcref.so,
recipe oracle/Makefile `make ref`).  Runs only where /root/reference exists; the fixtures (inputs + the
reference's outputs) are committed, the reference itself never travels.

    python tools/gen_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzpcref.so"))
fp = C.POINTER(C.c_float)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


MODEL_NAMES = ["fixedcorotated", "sand", "vonmises", "nacc", "eos"]


def partition_keys(pos, dx, side):
    """ComputeSparsity (offset -2, displacement 0.5) + EnlargeSparsity {0,1}^3 (simulation/sparsity/SparsityOp.hpp:59-115) as plain
    integer arithmetic; keys in lexicographic order (the table's numbering is an implementation detail, results are keyed)."""
    coord = np.floor(pos / np.float32(dx) + np.float32(0.5)).astype(np.int64) - 2
    base = np.unique(np.floor_divide(coord, side), axis=0)
    off = np.stack(np.meshgrid([0, 1], [0, 1], [0, 1], indexing="ij"), -1).reshape(-1, 3)
    return np.ascontiguousarray(np.unique((base[:, None, :] + off[None]).reshape(-1, 3), axis=0).astype(np.int32))


def gen_p2g_g2p():
    """Whole-function fixtures for P2GTransfer / G2PTransfer (simulation/transfer/P2G.hpp:51-125, G2P.hpp:44-83): 4096 particles, every
    constitutive model, block sides 4 and 8; outputs = the reference's own arena / stress / matrix code driven by oracle/ref_shim.cpp
    in SequentialExecutionPolicy order.  One fixture = inputs, the P2G grid, the grid handed to G2P (the P2G grid after the
    ComputeGridBlockVelocity arithmetic done here in float32 -- an INPUT of the G2P leg, stored), the G2P outputs, and the grid of
    the next step's P2G on those outputs (what the fused G2P2G pass produces)."""
    g = np.random.default_rng(20251003)
    dx, dt = np.float32(1.0 / 128), np.float32(1e-4)
    ppc, ncs = 2, 8
    k = ncs * ppc
    idx = np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3)
    h = dx / ppc
    pos0 = (np.array([0.3021, 0.2871, 0.3127]) + (idx + 0.5) * h + (g.random(idx.shape) - 0.5) * h * 0.9).astype(np.float32)
    n = pos0.shape[0]
    vel0 = (0.6 * g.standard_normal((n, 3)) + np.array([0.3, -1.0, 0.2])).astype(np.float32)
    C0 = (2.0 * g.standard_normal((n, 9))).astype(np.float32)
    F0 = (np.eye(3).reshape(1, 9) + 0.04 * g.standard_normal((n, 9))).astype(np.float32)
    J0 = (1 + 0.02 * g.standard_normal(n)).astype(np.float32)
    lj0 = (0.01 * g.standard_normal(n)).astype(np.float32)
    vol = np.float32(float(dx) ** 3 / ppc ** 3)
    mass = np.full(n, 1000.0 * vol, np.float32) * (1 + 0.1 * g.random(n)).astype(np.float32)
    # {volume, E, nu, cohesion, beta, yieldSurface, volumeCorrection, yieldStress, xi, fa, hardeningOn, bulk, viscosity}
    prm = {0: [vol, 5e4, 0.4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
           1: [vol, 5e4, 0.4, 0.0, 1.0, 0.816496580927726 * 2.0 * 0.5 / (3.0 - 0.5), 1, 0, 0, 0, 0, 0, 0],
           2: [vol, 5e4, 0.4, 0, 0, 0, 0, 500.0, 0, 0, 0, 0, 0],
           3: [vol, 5e4, 0.4, 0, 0.5, 0, 0, 0, 0.8, 45.0, 1, 0, 0],
           4: [vol, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4e4, 0.01]}
    ref.ref_mpm_p2g.restype = C.c_int
    ref.ref_mpm_g2p.restype = C.c_int
    out = dict(dx=dx, dt=dt, mass=mass, pos=pos0, vel=vel0, C=C0, F=F0, J=J0, logJp=lj0, gravity=np.array([0, -9.8, 0], np.float32))
    for model in range(5):
        out["prm_%s" % MODEL_NAMES[model]] = np.array(prm[model], np.float32)
        for side in ((4, 8) if model < 2 else (8,)):
            keys = partition_keys(pos0, dx, side)
            nb, nc = keys.shape[0], side ** 3
            pr = np.array(prm[model], np.float32)
            Fin = F0.copy()
            if model == 4:
                Fin[:, 0] = J0
            lj = lj0.copy()
            grid = np.zeros((nb, 7, nc), np.float32)
            args = (model, P(pr), C.c_float(dx), C.c_float(dt), side, nb, P(keys))
            miss = ref.ref_mpm_p2g(*args, P(grid), C.c_size_t(n), P(mass), P(pos0), P(vel0), P(C0), P(Fin), P(lj))
            assert miss == 0
            # ComputeGridBlockVelocity arithmetic (simulation/grid/GridOp.hpp:71-108) in float32: an input of the G2P leg
            gv = grid.copy()
            m = gv[:, 0, :]
            nz = m != 0
            inv = np.where(nz, np.float32(1) / np.where(nz, m, np.float32(1)), np.float32(0)).astype(np.float32)
            for d in range(3):
                gv[:, 1 + d, :] = np.where(nz, gv[:, 1 + d, :] * inv + out["gravity"][d] * dt, gv[:, 1 + d, :]).astype(np.float32)
            pos, vel, Cm, Fm = pos0.copy(), vel0.copy(), C0.copy(), Fin.copy()
            miss = ref.ref_mpm_g2p(*args, P(gv), C.c_size_t(n), P(pos), P(vel), P(Cm), P(Fm))
            assert miss == 0
            # P2G of the next step on the G2P outputs (the second half of the fused G2P2G pass)
            keys2 = partition_keys(pos, dx, side)
            assert {tuple(r) for r in keys2} <= {tuple(r) for r in keys}, "particles left the partition within one step"
            grid2 = np.zeros((nb, 7, nc), np.float32)
            lj2 = lj.copy()
            miss = ref.ref_mpm_p2g(*args, P(grid2), C.c_size_t(n), P(mass), P(pos), P(vel), P(Cm), P(Fm), P(lj2))
            assert miss == 0
            tag = "%s_s%d" % (MODEL_NAMES[model], side)
            out.update({"keys_s%d" % side: keys, "grid_" + tag: grid, "logJp1_" + tag: lj, "gridv_" + tag: gv, "pos_" + tag: pos,
                        "vel_" + tag: vel, "C_" + tag: Cm, "F_" + tag: Fm, "grid2_" + tag: grid2, "logJp2_" + tag: lj2})
            print("p2g_g2p %s: %d blocks, |m| %.3e, |v_p| max %.3f, plastic %d" %
                  (tag, nb, grid[:, 0].sum(), np.abs(vel).max(), (lj != lj0).sum()))
    np.savez_compressed(os.path.join(OUT, "p2g_g2p.npz"), **out)


def gen_c2():
    """Whole-function fixtures for P2C2GTransfer / G2C2PTransfer (simulation/transfer/P2C2G.hpp:53-189, G2C2P.hpp:59-135): 4096 particles,
    block sides 4 and 8, the reference's functor bodies driven by oracle/ref_shim.cpp over every (block, cell) of the partition in launch
    order, buckets in ascending particle id.  P2C2G: the models whose constitutive update is a pure function of the particle (fixed
    corotated, von Mises, the fluid) -- for the plastic models with logJp the reference functor re-runs the update once per (cell, particle)
    pair and stores logJp each time, the restatement evaluates each particle once; they are pinned through P2G.  G2C2P: v and B
    from a grid of velocities."""
    g = np.random.default_rng(20260929)
    dx, dt = np.float32(1.0 / 128), np.float32(1e-4)
    ppc, ncs = 2, 8
    k = ncs * ppc
    idx = np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3)
    h = dx / ppc
    pos0 = (np.array([0.3021, 0.2871, 0.3127]) + (idx + 0.5) * h + (g.random(idx.shape) - 0.5) * h * 0.9).astype(np.float32)
    n = pos0.shape[0]
    vel0 = (0.6 * g.standard_normal((n, 3)) + np.array([0.3, -1.0, 0.2])).astype(np.float32)
    B0 = (2.0 * g.standard_normal((n, 9)) * float(dx) ** 2 * 0.3).astype(np.float32)   # B = C / Dinv, Dinv in [2, 4] / dx^2
    F0 = (np.eye(3).reshape(1, 9) + 0.04 * g.standard_normal((n, 9))).astype(np.float32)
    J0 = (1 + 0.02 * g.standard_normal(n)).astype(np.float32)
    vol = np.float32(float(dx) ** 3 / ppc ** 3)
    mass = np.full(n, 1000.0 * vol, np.float32) * (1 + 0.1 * g.random(n)).astype(np.float32)
    prm = {0: [vol, 5e4, 0.4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
           2: [vol, 5e4, 0.4, 0, 0, 0, 0, 500.0, 0, 0, 0, 0, 0],
           4: [vol, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4e4, 0.01]}
    ref.ref_mpm_p2c2g.restype = C.c_int
    ref.ref_mpm_g2c2p.restype = C.c_int
    out = dict(dx=dx, dt=dt, mass=mass, pos=pos0, vel=vel0, B=B0, F=F0, J=J0)
    for side in (4, 8):
        keys = partition_keys(pos0, dx, side)
        nb, nc = keys.shape[0], side ** 3
        out["keys_s%d" % side] = keys
        for model in (0, 2, 4):
            pr = np.array(prm[model], np.float32)
            out["prm_%s" % MODEL_NAMES[model]] = pr
            Fin = F0.copy()
            if model == 4:
                Fin[:, 0] = J0
            lj = np.zeros(n, np.float32)
            grid = np.zeros((nb, 7, nc), np.float32)
            miss = ref.ref_mpm_p2c2g(model, P(pr), C.c_float(dx), C.c_float(dt), side, nb, P(keys), P(grid), C.c_size_t(n), P(mass), P(pos0),
                                     P(vel0), P(B0), P(Fin), P(lj))
            assert not grid[:, 4:].any() and abs(grid[:, 0].sum() - mass.sum()) < 1e-4 * mass.sum(), (miss, grid[:, 0].sum(), mass.sum())
            out["grid_%s_s%d" % (MODEL_NAMES[model], side)] = grid[:, :4].copy()
            print("c2 p2c2g %s side %d: %d blocks, %d (cell, node) pairs outside the partition, |m| %.3e" % (MODEL_NAMES[model], side, nb, miss, grid[:, 0].sum()))
        gv = np.zeros((nb, 7, nc), np.float32)
        gv[:, 1:4] = (0.5 * g.standard_normal((nb, 3, nc)) + np.array([0.3, -1.0, 0.2])[None, :, None]).astype(np.float32)
        vel, Bm = np.zeros((n, 3), np.float32), np.zeros((n, 9), np.float32)
        miss = ref.ref_mpm_g2c2p(C.c_float(dx), side, nb, P(keys), P(gv), C.c_size_t(n), P(pos0), P(vel), P(Bm))
        out.update({"gridv_s%d" % side: gv[:, 1:4].copy(), "g2c2p_vel_s%d" % side: vel, "g2c2p_B_s%d" % side: Bm})
        print("c2 g2c2p side %d: |v| max %.3f, |B| max %.3e, %d node lookups outside" % (side, np.abs(vel).max(), np.abs(Bm).max(), miss))
    np.savez_compressed(os.path.join(OUT, "c2.npz"), **out)


def gen_containers_seq():
    """Whole-function fixtures for bht<int, dim, int, B> (container/Bht.hpp:154-158, 612-698) and HashTable<int, 3, int>
    (container/HashTable.hpp:88-91, 383-400, 454-463, 496-500) under sequential insertion in input order: the tables byte for byte
    (padded key slots, indices, status), activeKeys, insert return values and query results -- produced by oracle/ref_shim.cpp
    over the reference's own universal_hash_base / hash_combine / storage_key_type_impl / next_2pow."""
    g = np.random.default_rng(20260928)
    ref.ref_bht_table_size_b.restype = C.c_size_t
    ref.ref_hashtable_table_size.restype = C.c_size_t
    out = {}
    # (tag, dim, B, nExpected, keys): duplicates on purpose; the `tight` case overflows buckets (failure tokens, success = 0)
    k3 = g.integers(-32, 32, (4096, 3), dtype=np.int32)
    cases = [("d3_b16", 3, 16, 4096, k3), ("d3_b32", 3, 32, 4096, k3), ("d3_b16_tight", 3, 16, 600, g.integers(-7, 7, (4096, 3), dtype=np.int32)),
             ("d1_b16", 1, 16, 1024, g.integers(-300, 300, (1024, 1), dtype=np.int32)),
             ("d2_b16", 2, 16, 2048, g.integers(-40, 40, (2048, 2), dtype=np.int32)),
             ("d4_b16", 4, 16, 2048, g.integers(-6, 6, (2048, 4), dtype=np.int32))]
    for tag, dim, B, nexp, keys in cases:
        keys = np.ascontiguousarray(keys)
        n = keys.shape[0]
        ts = int(ref.ref_bht_table_size_b(C.c_size_t(nexp), B))
        ks = 1 << (dim - 1).bit_length()
        ret = np.zeros(n, np.int32)
        kt = np.zeros((ts, ks), np.int32)
        ind, st = np.zeros(ts, np.int32), np.zeros(ts, np.int32)
        act = np.zeros((ts, dim), np.int32)
        cs = np.zeros(2, np.int32)
        q = np.ascontiguousarray(np.concatenate([keys[::7], g.integers(-400, 400, (64, dim), dtype=np.int32)]))
        qr = np.zeros(q.shape[0], np.int32)
        ref.ref_bht_seq(dim, B, C.c_size_t(nexp), P(keys), C.c_size_t(n), P(ret), P(kt), P(ind), P(st), P(act), P(cs), P(q), C.c_size_t(q.shape[0]), P(qr))
        out.update({tag + "_keys": keys, tag + "_n_expected": np.int64(nexp), tag + "_ret": ret, tag + "_table_keys": kt, tag + "_indices": ind,
                    tag + "_status": st, tag + "_active_keys": act[: cs[0]].copy(), tag + "_cnt": cs[0], tag + "_success": cs[1],
                    tag + "_queries": q, tag + "_query_ret": qr})
        # (indices of never-written slots are whatever the allocation held: masked out by the comparison, which uses keys != sentinel)
    hk = np.ascontiguousarray(g.integers(-64, 64, (4096, 3), dtype=np.int32))
    ts = int(ref.ref_hashtable_table_size(C.c_size_t(4096)))
    ret = np.zeros(4096, np.int32)
    kt, ind, act = np.zeros((ts, 3), np.int32), np.zeros(ts, np.int32), np.zeros((ts, 3), np.int32)
    cnt = C.c_int(0)
    q = np.ascontiguousarray(np.concatenate([hk[::5], g.integers(-500, 500, (64, 3), dtype=np.int32)]))
    qr = np.zeros(q.shape[0], np.int32)
    ref.ref_hashtable_seq(C.c_size_t(4096), P(hk), C.c_size_t(4096), P(ret), P(kt), P(ind), P(act), C.byref(cnt), P(q), C.c_size_t(q.shape[0]), P(qr))
    out.update({"ht_keys": hk, "ht_ret": ret, "ht_table_keys": kt, "ht_indices": ind, "ht_active_keys": act[: cnt.value].copy(), "ht_cnt": np.int32(cnt.value),
                "ht_queries": q, "ht_query_ret": qr})
    np.savez_compressed(os.path.join(OUT, "containers_seq.npz"), **out)
    print("containers_seq.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith(("_cnt", "_success"))})


def gen_grid_arena():
    """GridArena (math/curve/InterpolationKernel.hpp:271-560) -- the reference's own class instantiated over a dense box of values
    (oracle/ref_shim.cpp): stencil corner, local position, weights and their first / second derivatives per axis, isample of two channels,
    minimum / maximum, weight and weightsGradient at three stencil nodes; six kernels, derivative orders 0-2, collocated and the three
    staggered faces.  Points near the faces of the box exercise the default (background) value."""
    g = np.random.default_rng(20260929)
    ext, lo, dx, dflt = 12, np.array([-3, 2, -5], np.int32), np.float32(0.125), np.float32(7.0)
    data = g.standard_normal((2, ext, ext, ext)).astype(np.float32)
    npts = 96
    X = (lo + g.random((npts, 3)) * ext).astype(np.float32)       # index space; some stencils reach outside the box
    X[:8] = (lo + np.floor(g.random((8, 3)) * ext) + np.array([0.0, 0.5, 0.25])).astype(np.float32)   # on nodes / mid-cell planes
    cases = [(kt, o, f) for kt in range(3) for o in range(3) for f in (-1, 1)] + [(kt, 0, f) for kt in (3, 4, 5) for f in (-1, 0, 2)] + \
            [(1, 1, 0), (1, 1, 2), (2, 2, 0)]
    outs = np.zeros((len(cases), npts, 58), np.float32)
    for ci, (kt, o, f) in enumerate(cases):
        rc = ref.ref_grid_arena(kt, o, P(data), 2, P(lo), ext, C.c_float(float(dx)), P(X), C.c_size_t(npts), f, C.c_float(float(dflt)), P(outs[ci]))
        assert rc == 0
    np.savez_compressed(os.path.join(OUT, "grid_arena.npz"), data=data, lo=lo, ext=np.int32(ext), dx=dx, default=dflt, X=X,
                        cases=np.array(cases, np.int32), out=outs)
    print("grid_arena.npz: %d cases x %d points" % (len(cases), npts))


def main():
    os.makedirs(OUT, exist_ok=True)
    g = np.random.Generator(np.random.PCG64(0x9E3779B97F4A7C15 ^ 77))
    # ---- 3x3 SVD + constitutive models (math/matrix/SVD.hpp, physics/ConstitutiveModel_Vol_dP.hpp)
    n = 512
    F = np.concatenate([
        np.eye(3).reshape(1, 9) + 0.01 * g.standard_normal((n // 4, 9)),   # jello-like (C3)
        np.eye(3).reshape(1, 9) + 0.2 * g.standard_normal((n // 4, 9)),    # strongly deformed
        g.standard_normal((n // 4, 9)),                                    # generic, some inverted
        np.eye(3).reshape(1, 9) * g.uniform(0.6, 1.4, (n // 4, 1)) + 0.05 * g.standard_normal((n // 4, 9)),
    ]).astype(np.float32)
    U, S, V = np.zeros((n, 9), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 9), np.float32)
    for i in range(n):
        ref.ref_svd3(P(F[i]), P(U[i]), P(S[i]), P(V[i]))
    mu, lam = C.c_float(), C.c_float()
    ref.ref_lame(C.c_float(5e4), C.c_float(0.4), C.byref(mu), C.byref(lam))
    vol = 1.0 / 256 ** 3 / 8
    PF_fc = np.zeros((n, 9), np.float32)
    for i in range(n):
        ref.ref_stress_fixedcorotated(C.c_float(vol), mu, lam, P(F[i]), P(PF_fc[i]))
    ys = np.float32(0.816496580927726 * 2.0 * 0.5 / (3.0 - 0.5))
    logJp_in = (0.02 * g.standard_normal(n)).astype(np.float32)
    logJp_out = logJp_in.copy()
    F_sand = F.copy()
    PF_sand = np.zeros((n, 9), np.float32)
    for i in range(n):
        l = C.c_float(logJp_out[i])
        ref.ref_stress_sand(C.c_float(vol), mu, lam, C.c_float(0.0), C.c_float(1.0), C.c_float(ys), 1, C.byref(l),
                            P(F_sand[i]), P(PF_sand[i]))
        logJp_out[i] = l.value
    np.savez_compressed(os.path.join(OUT, "svd_stress.npz"), F=F, U=U, S=S, V=V, mu=np.float32(mu.value),
                        lam=np.float32(lam.value), vol=np.float32(vol), PF_fixedcorotated=PF_fc, yieldSurface=ys,
                        logJp_in=logJp_in, logJp_out=logJp_out, F_sand_out=F_sand, PF_sand=PF_sand)
    # ---- quadratic B-spline weights + base node (math/curve/InterpolationKernel.hpp:47-55,93-130)
    x = np.concatenate([g.uniform(-3, 3, (300, 3)), g.uniform(0.5, 1.5, (212, 3))]).astype(np.float32)
    ref.ref_base_node_quadratic.argtypes = [C.c_float]
    base = np.array([[ref.ref_base_node_quadratic(float(v)) for v in r] for r in x], np.int32)
    w = np.zeros((x.shape[0], 9), np.float32)
    for i in range(x.shape[0]):
        ref.ref_quadratic_weights(P(x[i]), P(w[i]))
    np.savez_compressed(os.path.join(OUT, "bspline.npz"), x=x, base_node=base, weights=w)
    # ---- hash functions (py_interop/HashUtils.hpp, math/Hash.hpp) + bht seeds (Bht.hpp:165-169)
    hp = (C.c_uint * 6)()
    ref.ref_bht_hash_params(hp)
    keys = np.concatenate([g.integers(-32, 32, (64, 3)), g.integers(-2 ** 31, 2 ** 31 - 1, (64, 3))]).astype(np.int32)
    ref.ref_universal_hash3.restype = C.c_uint
    ref.ref_universal_hash2.restype = C.c_uint
    ref.ref_universal_hash1.restype = C.c_uint
    h3 = np.array([[ref.ref_universal_hash3(hp[2 * f], hp[2 * f + 1], P(k)) for f in range(3)] for k in keys], np.uint32)
    h2 = np.array([[ref.ref_universal_hash2(hp[2 * f], hp[2 * f + 1], P(k)) for f in range(3)] for k in keys], np.uint32)
    h1 = np.array([[ref.ref_universal_hash1(hp[2 * f], hp[2 * f + 1], int(k[0])) for f in range(3)] for k in keys], np.uint32)
    ref.ref_next_2pow.restype = C.c_ulonglong
    ref.ref_next_2pow.argtypes = [C.c_ulonglong]
    ns = np.array([1, 2, 3, 4, 5, 7, 8, 9, 1000, 4096, 4097, 200000, 1 << 20, (1 << 20) + 1], np.uint64)
    n2 = np.array([ref.ref_next_2pow(int(v)) for v in ns], np.uint64)
    np.savez_compressed(os.path.join(OUT, "hash.npz"), hash_params=np.array(list(hp), np.uint32), keys=keys, h3=h3, h2=h2, h1=h1,
                        next_2pow_in=ns, next_2pow_out=n2)
    # ---- HashTable hash (container/HashTable.hpp:496-500 over math/Hash.hpp:19-28, 64-bit seed), own generator so that the
    # fixtures above stay byte-identical
    g2 = np.random.default_rng(20250928)
    hk = np.concatenate([g2.integers(-40, 40, (96, 4)), g2.integers(-2 ** 31, 2 ** 31 - 1, (96, 4))]).astype(np.int32)
    ref.ref_hashtable_do_hash.restype = C.c_int
    hh = np.array([[ref.ref_hashtable_do_hash(P(k), d) for d in (1, 2, 3, 4)] for k in hk], np.int32)
    np.savez_compressed(os.path.join(OUT, "hashtable.npz"), keys=hk, do_hash=hh)
    # ---- LBvh morton chain (container/Bvh.hpp:177-188 over AABBBox::getBoxCenter / getUniformCoord / morton_code<3>) + overlaps
    g3 = np.random.default_rng(20250929)
    c = g3.uniform(-2, 3, (400, 3)).astype(np.float32)
    e = g3.uniform(0.001, 0.2, (400, 3)).astype(np.float32)
    bvs = np.concatenate([c - e, c + e], axis=1).astype(np.float32)
    whole = np.concatenate([bvs[:, :3].min(0) - np.float32(10 * 1.1920929e-07), bvs[:, 3:].max(0) + np.float32(10 * 1.1920929e-07)]).astype(np.float32)
    bvs[0, :3] = whole[:3]; bvs[0, 3:] = whole[:3]      # centre on the low corner -> coord 0
    bvs[1, :3] = whole[3:]; bvs[1, 3:] = whole[3:]      # centre on the high corner -> coord 1 (the 1024 overflow case)
    ref.ref_lbvh_morton.restype = C.c_uint
    codes = np.array([ref.ref_lbvh_morton(P(whole), P(b)) for b in bvs], np.uint32)
    ov = np.array([[ref.ref_aabb_overlaps(P(bvs[i]), P(bvs[j])) for j in range(40)] for i in range(40)], np.int32)
    np.savez_compressed(os.path.join(OUT, "lbvh.npz"), whole=whole, bvs=bvs, codes=codes, overlaps40=ov)
    # ---- von Mises and NACC (physics/ConstitutiveModel_Vol_dP.hpp:48-243) on the F set of svd_stress.npz
    Fm = np.load(os.path.join(OUT, "svd_stress.npz"))["F"]
    nm = Fm.shape[0]
    g5 = np.random.default_rng(20251001)
    vm_yield = np.float32(500.0)
    F_vm, PF_vm = Fm.copy(), np.zeros((nm, 9), np.float32)
    for i in range(nm):
        ref.ref_stress_vonmises(C.c_float(vol), mu, lam, C.c_float(float(vm_yield)), P(F_vm[i]), P(PF_vm[i]))
    bulk, msqr = C.c_float(), C.c_float()
    ref.ref_nacc_config(C.c_float(5e4), C.c_float(0.4), C.c_float(45.0), C.byref(bulk), C.byref(msqr))
    lj_in = (0.02 * np.abs(g5.standard_normal(nm))).astype(np.float32)
    lj_in[nm // 2:] *= -1            # second half: -logJp > 0, where the host header's p0 differs from the CUDA header's
    lj_out, F_nacc, PF_nacc = lj_in.copy(), Fm.copy(), np.zeros((nm, 9), np.float32)
    for i in range(nm):
        l = C.c_float(lj_out[i])
        ref.ref_stress_nacc(C.c_float(vol), mu, lam, bulk, C.c_float(0.8), C.c_float(0.5), msqr, 1, C.byref(l), P(F_nacc[i]), P(PF_nacc[i]))
        lj_out[i] = l.value
    np.savez_compressed(os.path.join(OUT, "stress_models.npz"), vm_yield=vm_yield, F_vm_out=F_vm, PF_vm=PF_vm, nacc_bulk=np.float32(bulk.value),
                        nacc_msqr=np.float32(msqr.value), nacc_xi=np.float32(0.8), nacc_beta=np.float32(0.5), logJp_in=lj_in,
                        logJp_out=lj_out, F_nacc_out=F_nacc, PF_nacc=PF_nacc)
    print("von Mises: %d of %d projected; NACC: %d F changed, %d logJp changed" %
          ((np.abs(F_vm - Fm).max(1) > 1e-7).sum(), nm, (np.abs(F_nacc - Fm).max(1) > 1e-7).sum(), (lj_out != lj_in).sum()))
    # ---- colliders: Collider<AnalyticLevelSet<Plane|Cuboid|Sphere|Cylinder>>::resolveCollision (geometry/Collider.h:82-112)
    g4 = np.random.default_rng(20250930)
    cases = []
    for geom in range(4):
        for ctype in range(3):
            for rep in range(2):
                moving = rep == 1
                if geom == 0:
                    nrm = g4.standard_normal(3); nrm /= np.linalg.norm(nrm)
                    par = np.concatenate([g4.uniform(-0.2, 0.2, 3), nrm, [0, 0]])
                elif geom == 1:
                    lo = g4.uniform(-0.5, -0.1, 3); par = np.concatenate([lo, lo + g4.uniform(0.3, 0.9, 3), [0, 0]])
                elif geom == 2:
                    par = np.concatenate([g4.uniform(-0.2, 0.2, 3), [g4.uniform(0.3, 0.6)], [0, 0, 0, 0]])
                else:
                    par = np.concatenate([g4.uniform(-0.2, 0.2, 3), [g4.uniform(0.2, 0.5), g4.uniform(0.3, 0.8), float(g4.integers(0, 3))], [0, 0]])
                if moving:
                    q = g4.standard_normal(4); q /= np.linalg.norm(q)
                    w, x, y, z = q
                    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
                    sc, dsdt = g4.uniform(0.7, 1.5), g4.uniform(-0.3, 0.3)
                    om, b, dbdt = g4.uniform(-1, 1, 3), g4.uniform(-0.1, 0.1, 3), g4.uniform(-0.5, 0.5, 3)
                else:
                    R, sc, dsdt, om, b, dbdt = np.eye(3), 1.0, 0.0, np.zeros(3), np.zeros(3), np.zeros(3)
                cases.append(np.concatenate([[geom, ctype], par, [sc, dsdt], R.reshape(-1), om, b, dbdt]).astype(np.float32))
    cases = np.stack(cases)                       # [24, 2 + 8 + 2 + 9 + 9]
    npts = 96
    cx = g4.uniform(-0.9, 0.9, (cases.shape[0], npts, 3)).astype(np.float32)
    cv = g4.uniform(-1, 1, (cases.shape[0], npts, 3)).astype(np.float32)
    cout, cin = cv.copy(), np.zeros((cases.shape[0], npts), np.int32)
    for k, cs in enumerate(cases):
        par, sc, dsdt = np.ascontiguousarray(cs[2:10]), float(cs[10]), float(cs[11])
        R, om, b, dbdt = (np.ascontiguousarray(cs[12:21]), np.ascontiguousarray(cs[21:24]), np.ascontiguousarray(cs[24:27]),
                          np.ascontiguousarray(cs[27:30]))
        for i in range(npts):
            cin[k, i] = ref.ref_collider_resolve(int(cs[0]), int(cs[1]), P(par), C.c_float(sc), C.c_float(dsdt), P(R), P(om), P(b), P(dbdt),
                                                 P(cx[k, i]), P(cout[k, i]))
    np.savez_compressed(os.path.join(OUT, "collider.npz"), cases=cases, x=cx, v=cv, v_out=cout, inside=cin)
    print("collider: %d of %d points inside" % (cin.sum(), cin.size))
    gen_p2g_g2p()
    gen_containers_seq()
    gen_grid_arena()
    gen_c2()
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    sys.exit(main())
