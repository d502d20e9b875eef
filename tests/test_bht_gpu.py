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


BHT_SEQ_CASES = [("d3_b16", 3, 16), ("d3_b32", 3, 32), ("d1_b16", 1, 16), ("d2_b16", 2, 16), ("d4_b16", 4, 16)]


@pytest.mark.parametrize("tag,dim,bucket", BHT_SEQ_CASES)
def test_gpu_build_against_reference_made_tables(pol, tag, dim, bucket):
    """The GPU build against tests/golden/containers_seq.npz (tables made by the bodies of the reference's BHTView::insert / query over its own
    hash / slot / sizing code, oracle/ref_shim.cpp).  The dense index a key receives is arrival order in the reference too (Bht.hpp:517-521),
    so the comparison is the one SURVEY 8(a) states: same table size, same set of active keys, same build-success flag, queries agree on
    found / not found, and the canonical (lexicographic) numbering reproduces the sorted reference keys byte for byte."""
    import os
    from zpc_amd.containers import Bht
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "containers_seq.npz"))
    keys, nexp = np.ascontiguousarray(z[tag + "_keys"]), int(z[tag + "_n_expected"])
    n = keys.shape[0]
    tab = Bht(dim, nexp, bucket=bucket)
    assert tab.tableSize() == z[tag + "_table_keys"].shape[0]
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert(pol, torch.from_numpy(keys).cuda().data_ptr(), n, ret.data_ptr())
    cnt = int(z[tag + "_cnt"])
    assert tab.size() == cnt
    v = tab.view()
    succ = _d2h(v.success, 4)[0]
    assert succ == int(z[tag + "_success"]) == 1
    act = _d2h(v.activeKeys, cnt * dim * 4).reshape(cnt, dim)
    gact = z[tag + "_active_keys"]
    assert set(map(tuple, act)) == set(map(tuple, gact))
    q = np.ascontiguousarray(z[tag + "_queries"])
    qr = torch.empty(q.shape[0], dtype=torch.int32, device="cuda")
    tab.query(pol, torch.from_numpy(q).cuda().data_ptr(), q.shape[0], qr.data_ptr())
    qr = qr.cpu().numpy()
    gq = z[tag + "_query_ret"]
    assert np.array_equal(qr >= 0, gq >= 0) and np.array_equal(act[qr[qr >= 0]], q[qr >= 0]) and np.array_equal(gact[gq[gq >= 0]], q[gq >= 0])
    tab.canonicalize(pol)
    order = np.lexsort(tuple(gact[:, d] for d in range(dim - 1, -1, -1)))
    assert np.array_equal(_d2h(tab.view().activeKeys, cnt * dim * 4).reshape(cnt, dim), gact[order])
    # slot format as the reference leaves it: pad words untouched, status -1
    ts = tab.tableSize()
    assert (_d2h(tab.view().status, ts * 4) == -1).all() and np.array_equal(np.unique(z[tag + "_status"]), [-1])
    if dim == 3:
        raw = _d2h(tab.view().keys, ts * 16).reshape(ts, 4)
        assert (raw[:, 3] == 0x3f3f3f3f).all() and (z[tag + "_table_keys"][:, 3] == 0x3f3f3f3f).all()


def test_gpu_overfull_table_reports_like_the_reference(pol):
    """`d3_b16_tight`: 2 744 distinct keys into a table sized for 600 (129 buckets of 16, 15 usable): the reference's sequential build stores
    `cnt` keys, returns its failure token (INT_MIN) for the rest and clears `success` (Bht.hpp:536-541).  Which keys lose depends on arrival
    order (also between two runs of the reference's parallel policies); what must agree: the flag, failure tokens for exactly the keys that
    are not stored, every stored key queryable, and a stored count within the spread the three-bucket scheme allows."""
    import os
    from zpc_amd.containers import Bht
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "containers_seq.npz"))
    keys, nexp = np.ascontiguousarray(z["d3_b16_tight_keys"]), int(z["d3_b16_tight_n_expected"])
    n = keys.shape[0]
    tab = Bht(3, nexp)
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    dk = torch.from_numpy(keys).cuda()
    tab.insert(pol, dk.data_ptr(), n, ret.data_ptr())
    r = ret.cpu().numpy()
    assert _d2h(tab.view().success, 4)[0] == int(z["d3_b16_tight_success"]) == 0
    cnt, gcnt = tab.size(), int(z["d3_b16_tight_cnt"])
    ndist = np.unique(keys, axis=0).shape[0]
    assert gcnt < ndist and abs(cnt - gcnt) <= 0.05 * gcnt, (cnt, gcnt, ndist)
    q = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.query(pol, dk.data_ptr(), n, q.data_ptr())
    q = q.cpu().numpy()
    failed = r == np.iinfo(np.int32).min
    assert failed.any() and (q[failed] == -1).all()                                   # reported, and really not stored
    assert np.unique(keys[q >= 0], axis=0).shape[0] == cnt and (r[q < 0] < -1).all()   # every key is either stored or was reported


@pytest.mark.parametrize("dim", [1, 2, 3, 4])
def test_order_morton_renumbers_along_the_z_curve(pol, dim):
    """zs_rocm_order_morton__bht_*: the same key set, numbered in ascending Morton code of (key - min key) (64 / dim bits per component,
    component 0 most significant inside a bit group); queries return the new numbers."""
    from zpc_amd.containers import Bht
    g = np.random.default_rng(31 + dim)
    keys = np.unique(g.integers(-300, 700, (6000, dim), dtype=np.int32), axis=0)
    g.shuffle(keys)
    n = keys.shape[0]
    tab = Bht(dim, n)
    ret = torch.empty(n, dtype=torch.int32, device="cuda")
    tab.insert(pol, torch.from_numpy(keys).cuda().data_ptr(), n, ret.data_ptr())
    pol.syncCtx()
    assert tab.size() == n
    tab.order_morton(pol)
    pol.syncCtx()
    act = _d2h(tab.view().activeKeys, n * dim * 4).reshape(n, dim)
    assert set(map(tuple, act)) == set(map(tuple, keys))
    rel = (act.astype(np.int64) - keys.min(0).astype(np.int64)).astype(np.uint64)
    bits = 64 // dim
    code = np.zeros(n, dtype=object)
    for i in range(n):
        c = 0
        for d in range(dim):
            v = int(rel[i, d])
            for b in range(min(bits, 32)):
                c |= ((v >> b) & 1) << (b * dim + (dim - 1 - d))
        code[i] = c
    assert all(code[i] <= code[i + 1] for i in range(n - 1))
    tab.query(pol, torch.from_numpy(act.copy()).cuda().data_ptr(), n, ret.data_ptr())
    assert np.array_equal(ret.cpu().numpy(), np.arange(n))
