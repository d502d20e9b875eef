"""GPU parity for zs::HashTable<i32, dim, int> (container/HashTable.hpp): set-exact against the CPU oracle (sequential
insertion), table format, resize / preserve, heavy same-key contention; partition_for_particles + EnlargeSparsity on the
HashTable reproduce the bht partition and, adopted through zs_rocm_assign__bht, the same P2G grid."""
import ctypes as C

import numpy as np
import pytest

from util import rng, make_cloud

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1


def _d2h(ptr, nbytes):
    out = np.empty(nbytes // 4, np.int32)
    C.CDLL("libamdhip64.so").hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2)
    return out


def _oracle(oracle, dim, keys, nexp):
    oracle.orc_hashtable_create.restype = C.c_void_p
    oracle.orc_hashtable_active_keys.restype = C.POINTER(C.c_int32)
    t = C.c_void_p(oracle.orc_hashtable_create(dim, C.c_size_t(nexp)))
    oracle.orc_hashtable_insert_many(t, keys.ctypes.data_as(C.c_void_p), C.c_size_t(keys.shape[0]), None)
    n = oracle.orc_hashtable_size(t)
    act = np.ctypeslib.as_array(oracle.orc_hashtable_active_keys(t), shape=(n, dim)).copy()
    return t, n, act


@pytest.mark.parametrize("dim,n,span", [(3, 4096, 32), (3, 200_000, 40), (2, 50_000, 300), (1, 30_000, 20_000), (4, 100_000, 12),
                                        (3, 1, 5), (3, 100_000, 2), (4, 100_000, 2)])
def test_build_query_set_exact(pol, oracle, dim, n, span):
    from zpc_amd.containers import HashTable
    g = rng(70 + dim)
    keys = g.integers(-span, span, (n, dim), dtype=np.int32)
    if dim >= 3 and n > 1000:  # keys holding INT_MAX components look like half-written slots to a probe
        keys[0:32] = np.array([[INT_MAX, i % 5, i % 3] + [0] * (dim - 3) for i in range(32)], np.int32)
        keys[32:64] = np.array([[i % 5, INT_MAX, INT_MAX] + [1] * (dim - 3) for i in range(32)], np.int32)
    tab = HashTable(dim, n)
    t, on, oact = _oracle(oracle, dim, keys, n)
    assert tab.tableSize() == oracle.orc_hashtable_get_table_size(t) == 16 * (1 << (n - 1).bit_length())
    dk = torch.from_numpy(keys).cuda()
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert(pol, dk.data_ptr(), n, ret.data_ptr())
    assert tab.size() == on
    r = ret.cpu().numpy()
    won = r[r >= 0]
    assert won.size == on and np.array_equal(np.sort(won), np.arange(on))  # one inserter per distinct key, dense indices
    assert (r[r < 0] == -1).all()
    v = tab.view()
    act = _d2h(v.activeKeys, on * dim * 4).reshape(on, dim)
    assert set(map(tuple, act)) == set(map(tuple, oact))
    assert np.array_equal(act[r[r >= 0]], keys[r >= 0])
    # table format: every occupied slot holds a distinct key whose index points back at it; status all -1
    ts = tab.tableSize()
    raw = _d2h(v.keys, ts * dim * 4).reshape(ts, dim)
    idx = _d2h(v.indices, ts * 4)
    occ = (raw != INT_MAX).any(axis=1)
    assert occ.sum() == on and (idx[~occ] == -1).all()
    assert np.array_equal(act[idx[occ]], raw[occ])
    assert (_d2h(v.status, ts * 4) == -1).all()
    # query / entry
    q = np.concatenate([keys[: min(n, 5000)], g.integers(span + 1, span + 50, (500, dim), dtype=np.int32)])
    qr = torch.empty(q.shape[0], dtype=torch.int32, device="cuda")
    dq = torch.from_numpy(q).cuda()
    tab.query(pol, dq.data_ptr(), q.shape[0], qr.data_ptr())
    qh = qr.cpu().numpy()
    npres = min(n, 5000)
    assert (qh[:npres] >= 0).all() and np.array_equal(act[qh[:npres]], q[:npres]) and (qh[npres:] == -1).all()
    tab.entry(pol, dq.data_ptr(), q.shape[0], qr.data_ptr())
    eh = qr.cpu().numpy()
    assert np.array_equal(raw[eh[:npres]], q[:npres]) and (eh[npres:] == -1).all()
    oracle.orc_hashtable_destroy(t)


def test_resize_preserve_reset_insert_ids(pol, oracle):
    from zpc_amd.containers import HashTable
    g = rng(79)
    keys = np.unique(g.integers(-25, 25, (4000, 3), dtype=np.int32), axis=0)
    n = keys.shape[0]
    tab = HashTable(3, n)
    dk = torch.from_numpy(np.ascontiguousarray(keys)).cuda()
    tab.insert(pol, dk.data_ptr(), n)
    act0 = _d2h(tab.view().activeKeys, n * 12).reshape(n, 3)
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    # resize: indices preserved (HashTable.hpp:281-292)
    tab.resize(pol, 50_000)
    assert tab.tableSize() == 16 * 65536 and tab.size() == n
    da = torch.from_numpy(act0.copy()).cuda()
    tab.query(pol, da.data_ptr(), n, ret.data_ptr())
    assert np.array_equal(ret.cpu().numpy(), np.arange(n))
    # preserve(m < n): cnt = m, the first m active keys keep their indices, the others are gone (:258-279)
    m = n // 3
    tab.preserve(pol, m)
    assert tab.size() == m
    tab.query(pol, da.data_ptr(), n, ret.data_ptr())
    r = ret.cpu().numpy()
    assert np.array_equal(r[:m], np.arange(m)) and (r[m:] == -1).all()
    # reset + insert(key, id) (:405-421)
    tab.reset(pol, True)
    assert tab.size() == 0
    ids = torch.from_numpy((np.arange(n, dtype=np.int32) * 7 + 3)).cuda()
    ok = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert_ids(pol, dk.data_ptr(), ids.data_ptr(), n, ok.data_ptr())
    assert (ok.cpu().numpy() == 1).all() and tab.size() == 0  # cnt untouched by insert(key, id)
    tab.query(pol, dk.data_ptr(), n, ret.data_ptr())
    assert np.array_equal(ret.cpu().numpy(), np.arange(n) * 7 + 3)
    tab.insert_ids(pol, dk.data_ptr(), ids.data_ptr(), n, ok.data_ptr())
    assert (ok.cpu().numpy() == 0).all()


@pytest.mark.parametrize("side", [4, 8])
def test_partition_for_particles_matches_bht_partition_and_p2g(pol, oracle, side):
    """partition_for_particles (CleanSparsity + ComputeSparsity) + EnlargeSparsity[0,2)^3 on a HashTable give the same block
    set as the bht partition; adopting its numbering (zs_rocm_assign__bht) the P2G grid equals the bht-path grid block by
    block (both compared by block key)."""
    import zpc_amd as zs
    from zpc_amd.containers import HashTable
    from zpc_amd.mpm import MpmTransfer
    dx, dt = 1.0 / 64, 1e-4
    mass, pos, vel, Cm, F = make_cloud(12, dx, 4, seed=81)
    n = pos.shape[0]
    mp = MpmTransfer(pol, n, dx, dt, model=0, side=side, volume=dx ** 3 / 4)
    mp.upload(mass, pos, vel, Cm, F)
    nb = mp.build_partition(n)
    mp.rebin()
    mp.clear_grid()
    mp.p2g()
    pol.syncCtx()
    ref = mp.grid_by_key()
    # HashTable partition, reference-style
    ht = HashTable(3, max(1, n // side ** 3) * 8 + 64)
    L = zs.lib()
    mp2 = MpmTransfer(pol, n, dx, dt, model=0, side=side, volume=dx ** 3 / 4)
    mp2.upload(mass, pos, vel, Cm, F)
    L.zs_rocm_mpm_partition_for_particles(pol.handle, ht.handle, mp2._port("x"), n, C.c_float(dx), side)
    lo, hi = (C.c_int * 3)(0, 0, 0), (C.c_int * 3)(2, 2, 2)
    L.zs_rocm_mpm_enlarge_sparsity__hashtable(pol.handle, ht.handle, lo, hi)
    pol.syncCtx()
    assert ht.size() == nb
    hkeys = _d2h(ht.view().activeKeys, nb * 12).reshape(nb, 3)
    assert set(map(tuple, hkeys)) == set(ref.keys())
    mp2.adopt_partition(ht.view().activeKeys, nb)
    mp2.rebin()
    mp2.clear_grid()
    mp2.p2g()
    pol.syncCtx()
    got = mp2.grid_by_key()
    # same numbering as the HashTable: block i of the grid is activeKeys[i]
    g = mp2.grid.cpu().numpy().reshape(nb, 7, side ** 3)
    scale = np.abs(np.stack(list(ref.values()))).max(axis=(0, 2)) + 1e-30
    for i in range(0, nb, max(1, nb // 200)):
        k = tuple(int(x) for x in hkeys[i])
        assert np.array_equal(g[i], got[k])
        assert (np.abs(g[i] - ref[k]).max(axis=1) <= 2e-4 * scale).all()


def test_gpu_build_against_reference_made_table(pol):
    """HashTable<int, 3, int> on the GPU against tests/golden/containers_seq.npz (table made by the reference's insert / query bodies over its
    own hash_combine, oracle/ref_shim.cpp): table size, key set, count, queries found / not found, activeKeys[index] == key."""
    import os
    from zpc_amd.containers import HashTable
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "containers_seq.npz"))
    keys = np.ascontiguousarray(z["ht_keys"])
    n = keys.shape[0]
    tab = HashTable(3, n)
    assert tab.tableSize() == z["ht_table_keys"].shape[0]
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert(pol, torch.from_numpy(keys).cuda().data_ptr(), n, ret.data_ptr())
    cnt = int(z["ht_cnt"])
    assert tab.size() == cnt
    act = _d2h(tab.view().activeKeys, cnt * 12).reshape(cnt, 3)
    assert set(map(tuple, act)) == set(map(tuple, z["ht_active_keys"]))
    q = np.ascontiguousarray(z["ht_queries"])
    qr = torch.empty(q.shape[0], dtype=torch.int32, device="cuda")
    tab.query(pol, torch.from_numpy(q).cuda().data_ptr(), q.shape[0], qr.data_ptr())
    qr = qr.cpu().numpy()
    gq = z["ht_query_ret"]
    assert np.array_equal(qr >= 0, gq >= 0) and np.array_equal(act[qr[qr >= 0]], q[qr >= 0])
    # the occupied slots are the reference's occupied slots: same hash, same probe sequence, same keys at the same places whenever a
    # key's probe path met no other key (arrival order only matters where two keys compete for a slot)
    ts = tab.tableSize()
    raw = _d2h(tab.view().keys, ts * 12).reshape(ts, 3)
    gk = z["ht_table_keys"]
    occ, gocc = (raw != INT_MAX).any(1), (gk != INT_MAX).any(1)
    assert occ.sum() == gocc.sum() == cnt
    same_place = (raw[occ & gocc] == gk[occ & gocc]).all(1).mean()
    assert same_place > 0.95, same_place
