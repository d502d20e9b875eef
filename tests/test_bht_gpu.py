"""GPU parity for bht<int, dim, int, B> (dim 1-4, B 16|32): set-exact against the CPU oracle (sequential insertion), byte-exact
after canonicalisation (SURVEY.md 8a note), hash constants pinned, heavy same-key contention."""
import ctypes as C

import numpy as np
import pytest

from util import rng

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _oracle_table(oracle, dim, keys, nexp, bucket=16):
    oracle.orc_bht_create_b.restype = C.c_void_p
    oracle.orc_bht_size.restype = C.c_int32
    oracle.orc_bht_get_table_size.restype = C.c_size_t
    oracle.orc_bht_active_keys.restype = C.POINTER(C.c_int32)
    t = C.c_void_p(oracle.orc_bht_create_b(dim, C.c_size_t(nexp), bucket))
    oracle.orc_bht_insert_many(t, keys.ctypes.data_as(C.c_void_p), C.c_size_t(keys.shape[0]), None)
    n = oracle.orc_bht_size(t)
    act = np.ctypeslib.as_array(oracle.orc_bht_active_keys(t), shape=(n, dim)).copy()
    return t, n, act


def _d2h(ptr, nbytes):
    out = np.empty(nbytes // 4, np.int32)
    C.CDLL("libamdhip64.so").hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2)
    return out


@pytest.mark.parametrize("dim,n,span,bucket", [(3, 4096, 32, 16), (3, 200_000, 40, 16), (2, 50_000, 300, 16), (1, 30_000, 20_000, 16),
                                               (3, 1, 5, 16), (3, 100_000, 2, 16),
                                               # dim 4 (status-lock protocol) and bucket 32 (py_interop/BhtInstantiations.cpp:120-127)
                                               (4, 4096, 12, 16), (4, 200_000, 14, 16), (4, 100_000, 2, 16), (4, 1, 5, 32),
                                               (1, 30_000, 20_000, 32), (2, 50_000, 300, 32), (3, 200_000, 40, 32), (4, 150_000, 10, 32),
                                               # larger batches: mostly distinct keys, heavy duplication, few distinct keys, bucket 32
                                               (3, 400_000, 60, 16), (3, 1_000_000, 30, 16), (3, 300_000, 3, 16), (3, 500_000, 50, 32)])
def test_build_query_set_exact(pol, oracle, dim, n, span, bucket):
    from zpc_amd.containers import Bht
    g = rng(20 + dim)
    keys = g.integers(-span, span, (n, dim), dtype=np.int32)
    if dim == 4 and n > 1000:
        # keys holding sentinel words (0x3f3f3f3f) in either 8-byte half look like half-written slots to a probe
        S = 0x3f3f3f3f
        keys[0:64] = np.array([[S, S, i % 5, i % 3] for i in range(64)], np.int32)
        keys[64:128] = np.array([[i % 5, i % 3, S, S] for i in range(64)], np.int32)
        keys[128:160] = np.array([[S, i % 4, S, i % 2] for i in range(32)], np.int32)
    tab = Bht(dim, n, bucket=bucket)
    t, on, oact = _oracle_table(oracle, dim, keys, n, bucket)
    assert tab.tableSize() == oracle.orc_bht_get_table_size(t)
    v = tab.view()
    hp = (C.c_uint32 * 6)()
    oracle.orc_bht_hash_params(hp)
    assert [v.hf0x, v.hf0y, v.hf1x, v.hf1y, v.hf2x, v.hf2y] == list(hp) == [1872583848, 794921487, 111352301, 4000937544, 2360782358, 4070471979]
    dk = torch.from_numpy(keys).cuda()
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert(pol, dk.data_ptr(), n, ret.data_ptr())
    assert tab.size() == on
    r = ret.cpu().numpy()
    # exactly one inserter per distinct key got a dense index, everyone else sentinel_v (-1)
    won = r[r >= 0]
    assert won.size == on and np.array_equal(np.sort(won), np.arange(on))
    act = _d2h(tab.view().activeKeys, on * dim * 4).reshape(on, dim)
    assert set(map(tuple, act)) == set(map(tuple, oact))
    assert np.array_equal(act[r[r >= 0]], keys[r >= 0])  # activeKeys[index] == key for the winners
    # query: present keys -> index with activeKeys[i] == k ; absent -> -1
    q = np.concatenate([keys[: min(n, 5000)], g.integers(span + 1, span + 50, (500, dim), dtype=np.int32)])
    qr = torch.empty(q.shape[0], dtype=torch.int32, device="cuda")
    tab.query(pol, torch.from_numpy(q).cuda().data_ptr(), q.shape[0], qr.data_ptr())
    qr = qr.cpu().numpy()
    npres = min(n, 5000)
    assert (qr[:npres] >= 0).all() and np.array_equal(act[qr[:npres]], q[:npres])
    assert (qr[npres:] == -1).all()
    # status words all -1, padding ints untouched (reference table format)
    tsz = tab.tableSize()
    assert (_d2h(tab.view().status, tsz * 4) == -1).all()
    if dim >= 3:
        raw = _d2h(tab.view().keys, tsz * 16).reshape(tsz, 4)
        if dim == 3:
            assert (raw[:, 3] == 0x3f3f3f3f).all()
        filled = raw[(raw[:, :dim] != 0x3f3f3f3f).any(axis=1)][:, :dim]
        assert filled.shape[0] == on and len(set(map(tuple, filled))) == on  # no duplicate slots
    # canonical form is byte-identical to canonicalised oracle numbering
    tab.canonicalize(pol)
    act2 = _d2h(tab.view().activeKeys, on * dim * 4).reshape(on, dim)
    order = np.lexsort(tuple(oact[:, d] for d in range(dim - 1, -1, -1)))
    assert np.array_equal(act2, oact[order])
    tab.query(pol, torch.from_numpy(act2.copy()).cuda().data_ptr(), on, ret.data_ptr())
    assert np.array_equal(ret.cpu().numpy()[:on], np.arange(on))
    oracle.orc_bht_destroy(t)


def test_resize_and_reset(pol, oracle):
    from zpc_amd.containers import Bht
    g = rng(29)
    keys = g.integers(-20, 20, (3000, 3), dtype=np.int32)
    tab = Bht(3, 3000)
    dk = torch.from_numpy(keys).cuda()
    tab.insert(pol, dk.data_ptr(), 3000)
    n0 = tab.size()
    act0 = _d2h(tab.view().activeKeys, n0 * 12).reshape(n0, 3)
    tab.resize(pol, 100_000)
    oracle.orc_bht_table_size.restype = C.c_size_t
    assert tab.tableSize() == oracle.orc_bht_table_size(C.c_size_t(100_000)) and tab.size() == n0
    ret = torch.empty(n0, dtype=torch.int32, device="cuda")
    tab.query(pol, torch.from_numpy(act0.copy()).cuda().data_ptr(), n0, ret.data_ptr())
    assert np.array_equal(ret.cpu().numpy(), np.arange(n0))  # indices preserved (Bht.hpp:320-340)
    tab.reset(True)
    assert tab.size() == 0
    tab.query(pol, dk.data_ptr(), 100, ret.data_ptr())
    assert (ret.cpu().numpy()[:100] == -1).all()


def test_assign_adopts_external_numbering(pol, oracle):
    """zs_rocm_assign: table := {keys[i] -> i} (adopting e.g. a zs::HashTable partition's _activeKeys order)."""
    from zpc_amd.containers import Bht
    g = rng(61)
    keys = np.unique(g.integers(-50, 50, (20000, 3), dtype=np.int32), axis=0)
    keys = keys[g.permutation(keys.shape[0])]
    n = keys.shape[0]
    tab = Bht(3, n)
    dk = torch.from_numpy(np.ascontiguousarray(keys)).cuda()
    tab.assign(pol, dk.data_ptr(), n)
    assert tab.size() == n
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.query(pol, dk.data_ptr(), n, ret.data_ptr())
    assert np.array_equal(ret.cpu().numpy(), np.arange(n))
    act = _d2h(tab.view().activeKeys, n * 12).reshape(n, 3)
    assert np.array_equal(act, keys)


def test_three_batches_into_one_table(pol, oracle):
    """three insert calls (50 k, 600 k, 20 k keys, overlapping) into one table: the union is what the oracle's sequential insertion of
    all three batches holds; indices stay dense and unique; `ret` tells new keys (their index) from known ones (-1) in every batch."""
    from zpc_amd.containers import Bht
    g = rng(91)
    a = g.integers(-40, 40, (50_000, 3), dtype=np.int32)
    b = g.integers(-45, 45, (600_000, 3), dtype=np.int32)      # overlaps a
    c = g.integers(-50, 50, (20_000, 3), dtype=np.int32)
    allk = np.concatenate([a, b, c])
    tab = Bht(3, allk.shape[0])
    t, on, oact = _oracle_table(oracle, 3, allk, allk.shape[0])
    seen = set()
    base = 0
    for batch in (a, b, c):
        d = torch.from_numpy(batch).cuda()
        ret = torch.empty(batch.shape[0], dtype=torch.int32, device="cuda")
        tab.insert(pol, d.data_ptr(), batch.shape[0], ret.data_ptr())
        r = ret.cpu().numpy()
        size = tab.size()
        won = r[r >= 0]
        assert np.array_equal(np.sort(won), np.arange(base, size))     # the new keys took the next dense indices
        newkeys = set(map(tuple, batch[r >= 0]))
        assert len(newkeys) == won.size and not (newkeys & seen)
        assert set(map(tuple, batch[r < 0])) <= (seen | newkeys)          # -1: known before, or a duplicate inside the batch
        seen |= newkeys
        base = size
    assert base == on
    act = _d2h(tab.view().activeKeys, on * 12).reshape(on, 3)
    assert set(map(tuple, act)) == set(map(tuple, oact))
    qr = torch.empty(allk.shape[0], dtype=torch.int32, device="cuda")
    tab.query(pol, torch.from_numpy(allk).cuda().data_ptr(), allk.shape[0], qr.data_ptr())
    qr = qr.cpu().numpy()
    assert (qr >= 0).all() and np.array_equal(act[qr], allk)
    tsz = tab.tableSize()
    assert (_d2h(tab.view().status, tsz * 4) == -1).all()
    raw = _d2h(tab.view().keys, tsz * 16).reshape(tsz, 4)
    assert (raw[:, 3] == 0x3f3f3f3f).all()
    filled = raw[(raw[:, :3] != 0x3f3f3f3f).any(axis=1)][:, :3]
    assert filled.shape[0] == on and len(set(map(tuple, filled))) == on   # every key in exactly one slot
