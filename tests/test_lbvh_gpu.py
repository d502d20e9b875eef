"""GPU parity for zs::LBvh<3, int, f32> (container/Bvh.hpp): every array of the built tree (node order, parents, levels,
escape indices, leaf map, boxes) is compared BIT FOR BIT with the CPU oracle's restatement of LBvh::build / refit; the
stack-less traversal returns the oracle's hit lists in the same order and the brute-force overlap sets."""
import ctypes as C

import numpy as np
import pytest

from util import lbvh_boxes, oracle_lbvh

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _d2h(ptr, n, dtype=np.int32):
    out = np.empty(n, dtype)
    C.CDLL("libamdhip64.so").hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2)
    return out


def _arrays(bvh, n):
    v = bvh.view()
    nn = v.numNodes
    return {"numNodes": nn, "parents": _d2h(v.parents, nn), "levels": _d2h(v.levels, nn), "auxIndices": _d2h(v.auxIndices, nn),
            "leafInds": _d2h(v.leafInds, n), "bvs": _d2h(v.orderedBvs, nn * 6, np.float32).reshape(nn, 6)}


@pytest.mark.parametrize("n,dup", [(1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (37, 0), (1000, 0), (4097, 0), (100_000, 0), (20_000, 1), (300_000, 1)])
def test_build_bit_exact(pol, oracle, n, dup):
    from zpc_amd.containers import LBvh
    bv = lbvh_boxes(n, 90 + n % 97, dup)
    b, ref = oracle_lbvh(oracle, bv)
    bvh = LBvh()
    d = torch.from_numpy(bv).cuda()
    bvh.build(pol, d)
    assert bvh.numLeaves() == n and bvh.numNodes() == ref["numNodes"] == (2 * n - 1 if n > 2 else n)
    got = _arrays(bvh, n)
    keys = ["auxIndices", "leafInds", "bvs"] if n <= 2 else ["parents", "levels", "auxIndices", "leafInds", "bvs"]
    for k in keys:
        assert np.array_equal(got[k], ref[k]), k
    oracle.orc_lbvh_destroy(b)


def test_refit_and_rebuild_reuse(pol, oracle):
    from zpc_amd.containers import LBvh
    n = 50_000
    bv = lbvh_boxes(n, 131)
    bvh = LBvh()
    d = torch.from_numpy(bv).cuda()
    bvh.build(pol, d, refit=False)
    topo = _arrays(bvh, n)
    # move the boxes: topology stays, boxes follow (Bvh.hpp:1219-1248)
    g = np.random.default_rng(5)
    bv2 = bv + np.tile(g.normal(0, 0.01, (n, 3)).astype(np.float32), (1, 2))
    d2 = torch.from_numpy(np.ascontiguousarray(bv2)).cuda()
    bvh.refit(pol, d2)
    got = _arrays(bvh, n)
    for k in ("parents", "levels", "auxIndices", "leafInds"):
        assert np.array_equal(got[k], topo[k])
    b, ref = oracle_lbvh(oracle, bv, refit=0)
    oracle.orc_lbvh_refit(b, bv2.ctypes.data_as(C.c_void_p))
    oracle.orc_lbvh_bvs.restype = C.POINTER(C.c_float)
    rb = np.ctypeslib.as_array(oracle.orc_lbvh_bvs(b), shape=(2 * n - 1, 6))
    assert np.array_equal(got["bvs"], rb)
    with pytest.raises(RuntimeError):
        bvh.refit(pol, d2.reshape(-1)[: (n - 1) * 6])
    # a smaller rebuild reuses the storage
    bvh.build(pol, d.reshape(-1)[: 1000 * 6])
    b2, ref2 = oracle_lbvh(oracle, bv[:1000])
    got2 = _arrays(bvh, 1000)
    for k in ("parents", "levels", "auxIndices", "leafInds", "bvs"):
        assert np.array_equal(got2[k], ref2[k])
    oracle.orc_lbvh_destroy(b)
    oracle.orc_lbvh_destroy(b2)


@pytest.mark.parametrize("n,dup", [(2, 0), (3000, 0), (40_000, 1)])
def test_iter_neighbors_matches_oracle_order_and_brute_force(pol, oracle, n, dup):
    from zpc_amd.containers import LBvh
    bv = lbvh_boxes(n, 151, dup)
    b, ref = oracle_lbvh(oracle, bv)
    bvh = LBvh()
    d = torch.from_numpy(bv).cuda()
    bvh.build(pol, d)
    nq = min(n, 2000)
    q = np.ascontiguousarray(bv[:nq] + np.float32(0.003))
    offsets, ids = bvh.query(pol, torch.from_numpy(q).cuda())
    pol.syncCtx()
    off, ids = offsets.cpu().numpy(), ids.cpu().numpy()
    out = np.zeros(n, np.int32)
    oracle.orc_lbvh_iter_neighbors.restype = C.c_size_t
    for k in range(nq):
        cnt = oracle.orc_lbvh_iter_neighbors(b, q[k].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        mine = ids[off[k]:off[k + 1]]
        assert np.array_equal(mine, out[:cnt])  # same tree, same walk -> same order
        if k < 200:
            brute = np.nonzero(((bv[:, :3] <= q[k, 3:]) & (bv[:, 3:] >= q[k, :3])).all(1))[0]
            assert np.array_equal(np.sort(mine), brute)
    oracle.orc_lbvh_destroy(b)


def test_bulk_iter_neighbors_many_queries_in_random_order(pol, oracle):
    """50 000 queries in random order, some far outside the tree's box (no hits), some large (many hits): per query the same ids in the
    same order as the oracle's walk (Bvh.hpp:661-693), written to the query's own slot."""
    from zpc_amd.containers import LBvh
    n, nq = 40_000, 50_000
    bv = lbvh_boxes(n, 977, 1)
    b, ref = oracle_lbvh(oracle, bv)
    bvh = LBvh()
    bvh.build(pol, torch.from_numpy(bv).cuda())
    rng = np.random.default_rng(5)
    src = rng.integers(0, n, nq)
    q = bv[src] + rng.normal(0, 0.004, (nq, 1)).astype(np.float32)
    q[::97] += np.float32(3.0)          # far outside the whole box: no hits
    q[1::89, 3:] += np.float32(0.05)    # large boxes: many hits
    q = np.ascontiguousarray(q.astype(np.float32))
    offsets, ids = bvh.query(pol, torch.from_numpy(q).cuda())
    pol.syncCtx()
    off, ids = offsets.cpu().numpy(), ids.cpu().numpy()
    assert off[nq] == ids.shape[0] and (np.diff(off) >= 0).all()
    assert (np.diff(off)[::97] == 0).all()
    out = np.zeros(n, np.int32)
    oracle.orc_lbvh_iter_neighbors.restype = C.c_size_t
    for k in list(range(0, nq, 23)) + list(range(1, nq, 89))[:200]:
        cnt = oracle.orc_lbvh_iter_neighbors(b, q[k].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        assert np.array_equal(ids[off[k]:off[k + 1]], out[:cnt]), k
    oracle.orc_lbvh_destroy(b)


@pytest.mark.parametrize("n,dup", [(1, 0), (2, 0), (3000, 0), (40_000, 1)])
def test_self_collision_broadphase_matches_oracle_walk_and_brute_force(pol, oracle, n, dup):
    """zs_rocm_lbvh_self_query_{count,fill}: self_iter_neighbors over every leaf; per leaf the same ids in the same order as the
    oracle's walk of the same tree, and the pair set = brute-force overlapping pairs, each once."""
    from zpc_amd.containers import LBvh
    bv = lbvh_boxes(n, 977, dup)
    b, ref = oracle_lbvh(oracle, bv)
    bvh = LBvh()
    bvh.build(pol, torch.from_numpy(bv).cuda())
    offsets, pairs = bvh.self_query(pol)
    pol.syncCtx()
    off, pr = offsets.cpu().numpy(), pairs.cpu().numpy()
    leaf_prim = ref["auxIndices"][ref["leafInds"]] if n > 2 else np.arange(n)
    out = np.zeros(n, np.int32)
    oracle.orc_lbvh_self_iter_neighbors.restype = C.c_size_t
    for k in range(0, n, max(1, n // 1500)):
        c = oracle.orc_lbvh_self_iter_neighbors(b, C.c_int32(k), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        want = out[:c][out[:c] != leaf_prim[k]]
        mine = pr[off[k]:off[k + 1]]
        assert (mine[:, 0] == leaf_prim[k]).all() and np.array_equal(mine[:, 1], want)
    if n <= 3000:
        ov = ((bv[:, None, :3] <= bv[None, :, 3:]) & (bv[:, None, 3:] >= bv[None, :, :3])).all(2)
        ii, jj = np.nonzero(np.triu(ov, 1))
        got = np.sort(pr, axis=1)
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        assert np.array_equal(got, np.stack([ii, jj], 1).astype(np.int32))
    oracle.orc_lbvh_destroy(b)
