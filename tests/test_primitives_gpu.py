"""GPU parity: zs::reduce / exclusive_scan / inclusive_scan / radix_sort(_pair) through the C ABI vs the CPU
oracle (SequentialExecutionPolicy semantics).  Integers bit-exact; floats with the reference test's own
relative tolerance 1e-6 scaled by the condition of the sum (test/utils/parallel_primitives.hpp:27-31)."""
import ctypes as C

import numpy as np
import pytest

from util import rng

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

# sizes of the reference's own reduce test (test/parallel_primitives.cpp:24) + tile-edge cases
SIZES = [1, 2, 7, 16, 128, 1024, 4095, 4096, 4097, 65537, 2_000_000]


def dev(a):
    return torch.from_numpy(a).cuda()


def c_int(x):
    return C.c_int(int(x))


@pytest.mark.parametrize("n", SIZES)
def test_reduce_int_matches_serial_fold(pol, oracle, n):
    import zpc_amd as zs
    g = rng(1)
    a = g.integers(-2**30, 2**30, n, dtype=np.int32)
    d = dev(a)
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    for op, name in ((zs.plus, "sum"), (zs.getmin, "min"), (zs.getmax, "max"), (zs.multiplies, "prod")):
        zs.reduce(pol, d, None, out, op=op)
        exp = np.zeros(1, np.int32)
        getattr(oracle, "orc_reduce_%s_i32" % name)(a.ctypes.data_as(C.c_void_p), C.c_size_t(n), exp.ctypes.data_as(C.c_void_p))
        assert int(out.item()) == int(exp[0]), (name, n)


@pytest.mark.parametrize("n", [1, 7, 1024, 65537, 1_000_000])
def test_reduce_i64_f32_f64(pol, oracle, n):
    import zpc_amd as zs
    g = rng(2)
    a64 = g.integers(-2**40, 2**40, n, dtype=np.int64)
    out = torch.zeros(1, dtype=torch.int64, device="cuda")
    zs.reduce(pol, dev(a64), None, out)
    assert int(out.item()) == int(a64.sum())
    for dt, tdt, tol in ((np.float32, torch.float32, 1e-6), (np.float64, torch.float64, 1e-14)):
        a = g.standard_normal(n).astype(dt)
        out = torch.zeros(1, dtype=tdt, device="cuda")
        zs.reduce(pol, dev(a), None, out)
        ref = np.sum(a.astype(np.float64))
        assert abs(float(out.item()) - ref) <= tol * np.sum(np.abs(a.astype(np.float64))) + 1e-30
        zs.reduce(pol, dev(a), None, out, op=zs.getmax)
        assert float(out.item()) == float(a.max())
        zs.reduce(pol, dev(a), None, out, op=zs.getmin)
        assert float(out.item()) == float(a.min())


@pytest.mark.parametrize("n", SIZES)
def test_scan_int_bit_exact(pol, oracle, n):
    import zpc_amd as zs
    g = rng(3)
    a = g.integers(-2**30, 2**30, n, dtype=np.int32)
    d = dev(a)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    exp = np.zeros(n, np.int32)
    zs.exclusive_scan(pol, d, out)
    oracle.orc_exclusive_scan_sum_i32(a.ctypes.data_as(C.c_void_p), C.c_size_t(n), exp.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out.cpu().numpy(), exp)
    zs.inclusive_scan(pol, d, out)
    oracle.orc_inclusive_scan_sum_i32(a.ctypes.data_as(C.c_void_p), C.c_size_t(n), exp.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out.cpu().numpy(), exp)
    # in place + non-identity init (C++ face: exclusive_scan(pol, first, last, d_first, init))
    d2 = d.clone()
    zs.exclusive_scan(pol, d2, d2, init=17)
    assert np.array_equal(d2.cpu().numpy(), (np.cumsum(np.concatenate([[17], a[:-1]]), dtype=np.int64)).astype(np.int32))


@pytest.mark.parametrize("n", [1, 5, 2048, 2049, 300_001])
def test_scan_i64_f64_f32(pol, oracle, n):
    import zpc_amd as zs
    g = rng(4)
    a = g.integers(-2**40, 2**40, n, dtype=np.int64)
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    zs.inclusive_scan(pol, dev(a), out)
    assert np.array_equal(out.cpu().numpy(), np.cumsum(a))
    zs.exclusive_scan(pol, dev(a), out)
    assert np.array_equal(out.cpu().numpy(), np.cumsum(a) - a)
    f = g.random(n)
    outf = torch.empty(n, dtype=torch.float64, device="cuda")
    zs.inclusive_scan(pol, dev(f), outf)
    assert np.allclose(outf.cpu().numpy(), np.cumsum(f), rtol=1e-12)
    f32 = g.random(n).astype(np.float32)
    outf = torch.empty(n, dtype=torch.float32, device="cuda")
    zs.exclusive_scan(pol, dev(f32), outf)
    exp = np.cumsum(f32.astype(np.float64)) - f32
    assert np.allclose(outf.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n", [1, 2, 7, 16, 128, 1024, 4096, 4097, 65537, 1_000_000])
def test_radix_sort_int_bit_exact(pol, oracle, n):
    import zpc_amd as zs
    g = rng(5)
    a = g.integers(-2**30, 2**30, n, dtype=np.int32)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    zs.radix_sort(pol, dev(a), out)
    exp = np.zeros(n, np.int32)
    oracle.orc_radix_sort_i32(a.ctypes.data_as(C.c_void_p), exp.ctypes.data_as(C.c_void_p), C.c_size_t(n), c_int(0), c_int(32))
    assert np.array_equal(exp, np.sort(a))  # oracle == mathematical definition
    assert np.array_equal(out.cpu().numpy(), exp)


@pytest.mark.parametrize("variant", ["all_equal", "sorted", "negatives", "dups"])
def test_radix_sort_variants(pol, oracle, variant):
    import zpc_amd as zs
    n = 100_003
    g = rng(6)
    a = {"all_equal": np.full(n, 12345, np.int32), "sorted": (np.arange(n, dtype=np.int32) - n // 2),
         "negatives": -g.integers(0, 2**31 - 1, n, dtype=np.int64).astype(np.int32),
         "dups": g.integers(-3, 3, n, dtype=np.int32)}[variant]
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    zs.radix_sort(pol, dev(a), out)
    assert np.array_equal(out.cpu().numpy(), np.sort(a))


@pytest.mark.parametrize("n,sbit,ebit", [(1, 0, 32), (1000, 0, 32), (70_001, 0, 32), (70_001, 8, 24), (70_001, 3, 13), (5000, 0, 4), (5000, 31, 32)])
def test_radix_sort_pair_stable_and_window(pol, oracle, n, sbit, ebit):
    import zpc_amd as zs
    g = rng(7)
    k = g.integers(-50, 50, n, dtype=np.int32) if sbit == 0 and ebit == 32 else g.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    v = np.arange(n, dtype=np.int32)
    ko = torch.empty(n, dtype=torch.int32, device="cuda")
    vo = torch.empty(n, dtype=torch.int32, device="cuda")
    zs.radix_sort_pair(pol, dev(k), dev(v), ko, vo, sbit=sbit, ebit=ebit)
    ek, ev = np.zeros(n, np.int32), np.zeros(n, np.int32)
    oracle.orc_radix_sort_pair_i32(k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), ek.ctypes.data_as(C.c_void_p),
                                   ev.ctypes.data_as(C.c_void_p), C.c_size_t(n), c_int(sbit), c_int(ebit))
    assert np.array_equal(ko.cpu().numpy(), ek)
    assert np.array_equal(vo.cpu().numpy(), ev)  # stability: unique permutation


def _small_path_keys(kind, n, g):
    u = lambda: g.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    if kind == "uniform":        # three launches, buckets finished in LDS
        return u()
    if kind == "half_range":     # BASELINE config 1's ints in [-2^30, 2^30): half of the top window's digits occur, buckets of n / 128
        return g.integers(-2**30, 2**30, n, dtype=np.int32)
    if kind == "morton":         # 30 significant bits under the 32-bit window: the split digit sits under the highest differing bit
        return g.integers(0, 2**30, n, dtype=np.int32)
    if kind == "narrow":
        return g.integers(0, 70000, n, dtype=np.int32)
    if kind == "sorted":
        return np.sort(u())
    if kind == "equal":          # nothing differs: copy
        return np.full(n, -7, np.int32)
    if kind == "two_values":     # <= 8 differing bits: the split is the sort
        return (g.integers(0, 2, n, dtype=np.int32) * 200 - 3).astype(np.int32)
    if kind == "sentinel":       # one bucket too large for LDS: sorted by all workgroups inside the finish launch
        a = u()
        a[g.random(n) < 0.2] = 2**31 - 1
        return a
    if kind == "few_values":     # several such buckets: LSD passes inside the finish launch
        return (g.integers(-8, 8, n, dtype=np.int32) * (1 << 27) + g.integers(0, 3, n, dtype=np.int32)).astype(np.int32)
    if kind == "outlier":        # tiles disagree about the highest differing bit, rows cannot be rebuilt
        a = g.integers(0, 70000, n, dtype=np.int32)
        a[n // 3], a[n - 1] = 70000 * 5, 70000 * 3
        return a
    if kind == "far_outlier":    # ... and can: every other tile's keys share one digit of the true window
        a = g.integers(0, 70000, n, dtype=np.int32)
        a[n // 2] = 2**30 + 12345
        return a
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "half_range", "morton", "narrow", "sorted", "equal", "two_values", "sentinel", "few_values", "outlier",
                                  "far_outlier"])
@pytest.mark.parametrize("n,sbit,ebit", [(8193, 0, 32), (300_001, 0, 32), (300_001, 5, 29), (1_000_000, 0, 32), (2_048_000, 0, 32),
                                         (2_048_001, 0, 32)])
def test_radix_sort_small_input_path_every_mode(pol, oracle, kind, n, sbit, ebit):
    """The three-launch path for <= 2 048 000 4-byte keys (primitives.hip, "split + finish") in each of its modes, keys and pairs, against the
    oracle's restatement of the reference's stable LSD sort (execution/ExecutionPolicy.hpp:777-781 semantics: order by the bits
    [sbit, ebit) of the sign-flipped key, ties in input order); 2 048 001 keys is the first size on the ordinary passes again."""
    import zpc_amd as zs
    k = _small_path_keys(kind, n, rng(n % 1000 + len(kind)))
    v = np.arange(n, dtype=np.int32)
    ko, vo, k1 = (torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(3))
    zs.radix_sort_pair(pol, dev(k), dev(v), ko, vo, sbit=sbit, ebit=ebit)
    zs.radix_sort(pol, dev(k), k1, sbit=sbit, ebit=ebit)
    ek, ev = np.zeros(n, np.int32), np.zeros(n, np.int32)
    oracle.orc_radix_sort_pair_i32(k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), ek.ctypes.data_as(C.c_void_p),
                                   ev.ctypes.data_as(C.c_void_p), C.c_size_t(n), c_int(sbit), c_int(ebit))
    assert np.array_equal(ko.cpu().numpy(), ek)
    assert np.array_equal(vo.cpu().numpy(), ev)
    assert np.array_equal(k1.cpu().numpy(), ek)
    # in place (execution/ExecutionPolicy.hpp stages through temporaries, so aliasing is legal there)
    kk = dev(k)
    zs.radix_sort(pol, kk, kk, sbit=sbit, ebit=ebit)
    assert np.array_equal(kk.cpu().numpy(), ek)


@pytest.mark.parametrize("dtype", ["u32", "i64", "u64"])
def test_radix_sort_other_key_widths(pol, oracle, dtype):
    import zpc_amd as zs
    n = 200_001
    g = rng(8)
    npdt, tdt, lo, hi = {"u32": (np.uint32, torch.uint32, 0, 2**32), "i64": (np.int64, torch.int64, -2**62, 2**62),
                         "u64": (np.uint64, torch.uint64, 0, 2**63)}[dtype]
    k = g.integers(lo, hi, n, dtype=np.int64 if dtype != "u64" else np.uint64).astype(npdt)
    v = np.arange(n, dtype=np.int32)
    ko = torch.empty(n, dtype=tdt, device="cuda")
    vo = torch.empty(n, dtype=torch.int32, device="cuda")
    zs.radix_sort_pair(pol, dev(k), dev(v), ko, vo)
    order = np.argsort(k, kind="stable")
    assert np.array_equal(ko.cpu().numpy(), k[order])
    assert np.array_equal(vo.cpu().numpy(), v[order])


MS_DT = {"i32": (np.int32, torch.int32), "u32": (np.uint32, torch.uint32), "i64": (np.int64, torch.int64),
         "u64": (np.uint64, torch.uint64), "f32": (np.float32, torch.float32), "f64": (np.float64, torch.float64)}


def _ms_keys(g, S, n, dups):
    if S in ("f32", "f64"):
        k = (g.integers(-20, 20, n) * 0.25) if dups else g.standard_normal(n)
    elif S == "u64":
        k = g.integers(0, 40, n, dtype=np.uint64) if dups else g.integers(0, 2**64 - 1, n, dtype=np.uint64)
    else:
        k = g.integers(0 if S[0] == "u" else -20, 20, n) if dups else g.integers(0 if S[0] == "u" else -2**31, 2**31 - 1, n)
    return np.ascontiguousarray(k.astype(MS_DT[S][0]))


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 255, 2047, 2048, 2049, 4096, 6145, 65537, 1_000_003])
@pytest.mark.parametrize("S", ["i32", "f32"])
def test_merge_sort_pair_bit_exact(pol, oracle, n, S):
    """zs::merge_sort_pair (stable, in place) vs the oracle's restatement of the sequential routine; duplicate keys pin
    stability (the values are the unique stable permutation)."""
    import zpc_amd as zs
    g = rng(31)
    for dups in (True, False):
        k = _ms_keys(g, S, n, dups)
        v = np.arange(n, dtype=np.int32)
        dk, dv = dev(k.copy()), dev(v.copy())
        zs.merge_sort_pair(pol, dk, dv)
        ek, ev = k.copy(), v.copy()
        getattr(oracle, "orc_merge_sort_" + S)(ek.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p), C.c_size_t(n), c_int(0))
        assert np.array_equal(dk.cpu().numpy(), ek)
        assert np.array_equal(dv.cpu().numpy(), ev)


@pytest.mark.parametrize("S", ["i32", "u32", "i64", "u64", "f32", "f64"])
@pytest.mark.parametrize("desc", [0, 1])
def test_merge_sort_types_and_descending(pol, oracle, S, desc):
    import zpc_amd as zs
    n = 300_001
    g = rng(32)
    k = _ms_keys(g, S, n, True)
    v = np.arange(n, dtype=np.int32)
    dk, dv = dev(k.copy()), dev(v.copy())
    zs.merge_sort_pair(pol, dk, dv, descending=bool(desc))
    ek, ev = k.copy(), v.copy()
    getattr(oracle, "orc_merge_sort_" + S)(ek.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p), C.c_size_t(n), c_int(desc))
    assert np.array_equal(dk.cpu().numpy(), ek) and np.array_equal(dv.cpu().numpy(), ev)
    k2 = _ms_keys(g, S, n, False)
    dk2 = dev(k2.copy())
    zs.merge_sort(pol, dk2, descending=bool(desc))
    exp = np.sort(k2)
    assert np.array_equal(dk2.cpu().numpy(), exp[::-1] if desc else exp)


def test_merge_sort_through_tilevector_iterators(pol, oracle):
    """C ABI form merge_sort(_pair)__rocm_T_1 over aosoa iterators (py_interop/cuda/ExecutionPolicy.cpp:99-111): sort channel
    1 of a TileVector<float,32>{a:3} in place, carrying an int permutation; the other channels stay untouched."""
    import zpc_amd as zs
    from zpc_amd.primitives import Iter
    L, Cn, n = 32, 3, 10_000
    g = rng(33)
    tiles = (n + L - 1) // L
    buf = g.standard_normal(tiles * L * Cn).astype(np.float32)
    d = dev(buf.copy())
    it = Iter.aosoa(d, 0, L, 1, Cn)
    idx = np.arange(n)
    off = (idx // L * Cn + 1) * L + idx % L
    keys = buf[off].copy()
    vals = dev(np.arange(n, dtype=np.int32))
    zs.merge_sort_pair(pol, it, vals, n)
    order = np.argsort(keys, kind="stable")
    exp = buf.copy()
    exp[off] = keys[order]
    assert np.array_equal(d.cpu().numpy(), exp)
    assert np.array_equal(vals.cpu().numpy(), order.astype(np.int32))
    # keys-only form on an int AoS iterator with an index offset
    a = g.integers(-1000, 1000, 5000, dtype=np.int32)
    da = dev(a.copy())
    zs.merge_sort(pol, Iter.aos(da, idx=100), 4000)
    e = a.copy()
    e[100:4100] = np.sort(a[100:4100])
    assert np.array_equal(da.cpu().numpy(), e)


def test_tilevector_channel_iterators(pol, oracle):
    """reduce over the "b" channel of TileVector<int,32>{a:3,b:2,c:1}: the reference's own test
    (test/parallel_primitives.cpp:7-30, test/utils/initialization.hpp:75-93) through the iterator ABI."""
    import zpc_amd as zs
    from zpc_amd.primitives import Iter
    L_, C_ = 32, 6
    for n in (1, 2, 7, 16, 128, 1024, 200_000):
        g = rng(9)
        tiles = (n + L_ - 1) // L_
        buf = g.integers(-1000, 1000, tiles * L_ * C_, dtype=np.int32)
        d = dev(buf)
        chn = 3  # offset of "b"
        idx = np.arange(n)
        vals = buf[(idx // L_ * C_ + chn) * L_ + idx % L_]
        out = torch.zeros(1, dtype=torch.int32, device="cuda")
        it = Iter.aosoa(d, 0, L_, chn, C_)
        zs.reduce(pol, it, n, out, op=zs.plus)
        assert int(out.item()) == int(vals.sum(dtype=np.int64).astype(np.int32))
        zs.reduce(pol, it, n, out, op=zs.getmax)
        assert int(out.item()) == int(vals.max())
        zs.reduce(pol, it, n, out, op=zs.getmin)
        assert int(out.item()) == int(vals.min())
        # scan + sort from the channel iterator into a contiguous vector
        o = torch.empty(n, dtype=torch.int32, device="cuda")
        zs.exclusive_scan(pol, it, o, n=n)
        assert np.array_equal(o.cpu().numpy(), (np.cumsum(vals, dtype=np.int64) - vals).astype(np.int32))
        zs.radix_sort(pol, it, o, n=n)
        assert np.array_equal(o.cpu().numpy(), np.sort(vals))


def test_empty_ranges(pol):
    """n = 0 through the C ABI: reduce returns init (SequentialExecutionPolicy: the fold over an empty range,
    execution/ExecutionPolicy.hpp:267-274), scans and sorts leave their outputs untouched, nothing is launched out of bounds."""
    import zpc_amd as zs
    a = torch.zeros(4, dtype=torch.int32, device="cuda")[:0]
    out1 = torch.full((1,), 123, dtype=torch.int32, device="cuda")
    for op, init in ((zs.plus, 0), (zs.multiplies, 1), (zs.getmin, 2**31 - 1), (zs.getmax, -2**31)):
        zs.reduce(pol, a, 0, out1, op=op)
        assert int(out1.item()) == init
    zs.reduce(pol, a, 0, out1, init=77, op=zs.plus)
    assert int(out1.item()) == 77
    guard = torch.full((8,), -5, dtype=torch.int32, device="cuda")
    zs.exclusive_scan(pol, a, guard[:0])
    zs.inclusive_scan(pol, a, guard[:0])
    zs.radix_sort(pol, a, guard[:0])
    zs.radix_sort_pair(pol, a, a, guard[:0], guard[:0])
    zs.merge_sort(pol, guard[:0])
    pol.syncCtx()
    assert (guard.cpu().numpy() == -5).all() and zs.lib().zs_rocm_last_error(-1) == 0


def test_radix_sort_randomised_stress():
    """tools/sort_stress.py: 120 random sizes (1 .. 5 M keys, ragged last tiles), four key ranges (2 values .. 31 bits), keys and pairs,
    against torch.sort(stable=True) -- the decoupled look-back with early tile aggregates must be independent of timing"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sort_stress.py"), "120"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"0 mismatches" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-1500:])


@pytest.mark.parametrize("kind", ["uniform", "narrow", "sorted", "equal", "sentinel", "few_values", "outlier"])
@pytest.mark.parametrize("n,sbit,ebit", [(300_001, 0, 64), (1_000_000, 0, 64), (1_000_000, 7, 50), (1_024_001, 0, 64)])
def test_radix_sort_small_input_path_8_byte_keys(pol, kind, n, sbit, ebit):
    """The same path for 8-byte keys (buckets of up to 8192 keys, up to seven LDS passes; 1 024 001 keys: the ordinary passes again): order by
    the bits [sbit, ebit) of the sign-flipped key, ties in input order (numpy stable argsort of the extracted window)."""
    import zpc_amd as zs
    g = rng(n % 977 + len(kind))
    if kind == "uniform":
        k = g.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    elif kind == "narrow":
        k = g.integers(0, 70000, n, dtype=np.int64) + (1 << 40)
    elif kind == "sorted":
        k = np.sort(g.integers(-2**63, 2**63 - 1, n, dtype=np.int64))
    elif kind == "equal":
        k = np.full(n, -(1 << 50) + 3, np.int64)
    elif kind == "sentinel":
        k = g.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
        k[g.random(n) < 0.2] = 2**63 - 1
    elif kind == "few_values":
        k = g.integers(-8, 8, n, dtype=np.int64) * (1 << 59) + g.integers(0, 3, n, dtype=np.int64)
    else:
        k = g.integers(0, 70000, n, dtype=np.int64)
        k[n // 3], k[n - 1] = 70000 * 5, 1 << 45
    v = np.arange(n, dtype=np.int32)
    ko = torch.empty(n, dtype=torch.int64, device="cuda")
    vo = torch.empty(n, dtype=torch.int32, device="cuda")
    k1 = torch.empty(n, dtype=torch.int64, device="cuda")
    zs.radix_sort_pair(pol, dev(k), dev(v), ko, vo, sbit=sbit, ebit=ebit)
    zs.radix_sort(pol, dev(k), k1, sbit=sbit, ebit=ebit)
    w = ((k.view(np.uint64) ^ np.uint64(1 << 63)) >> np.uint64(sbit)) & np.uint64((1 << (ebit - sbit)) - 1 if ebit - sbit < 64 else 2**64 - 1)
    order = np.argsort(w, kind="stable")
    assert np.array_equal(vo.cpu().numpy(), v[order])
    assert np.array_equal(ko.cpu().numpy(), k[order])
    assert np.array_equal(k1.cpu().numpy(), k[order])


def test_radix_sort_small_path_randomised_stress():
    """tools/sort_small_stress.py: 300 random sizes (1 .. 2.1 M keys) x eleven key distributions (random ranges and offsets, sentinels,
    sorted / reverse-sorted / block-sorted inputs, few distinct values, outliers at random distances, equal keys, two clusters) x random bit
    windows, keys and pairs, against torch.sort(stable=True): every mode of the three-launch path, including the in-launch grid barriers"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sort_small_stress.py"), "300", "7"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert r.returncode == 0 and b"0 mismatches" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-1500:])


def test_two_streams_sorting_in_the_in_launch_lsd_mode_do_not_wait_for_each_other():
    """tools/sort_two_streams.py: two streams, each sorting 2 M keys in the small path's slowest mode (245 tiles each: more workgroups than
    CUs together) while a third stream keeps the device busy; 100 rounds under a two-minute limit.  The in-launch passes are ticketed
    (no co-residency assumed), so this must neither hang nor mis-sort."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sort_two_streams.py"), "100"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=120)
    assert r.returncode == 0 and b"0 mismatches" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-1500:])


def test_scan_interleaved_types_share_the_control_block():
    """tools/scan_stress.py: 300 scans alternating int32 (exclusive) and int64 (inclusive), 1 .. 3 M elements: scans of up to 4096 tiles run as
    ONE launch on generation-tagged descriptors in the stream's control block (no memset); descriptors left by earlier calls, of either
    layout, must read as invalid"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan_stress.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"0 mismatches" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-1500:])


@pytest.mark.parametrize("n", [5000, 300_001, 1_000_000, 3_000_000])
@pytest.mark.parametrize("case", ["constant_keys", "window_over_equal_bits", "ordinary"])
def test_radix_sort_pair_keys_in_place_values_out_of_place(pol, case, n):
    """radix_sort_pair(keys, iota, keys, perm): keys sorted IN PLACE while the values go to a separate array -- the argsort idiom.
    When no key differs inside [sbit, ebit) the sort degenerates to a copy; the copy of the values must still happen although the key
    arrays coincide (r03's small-input path skipped both copies behind one `keys_in != keys_out` test: ADVICE r03, medium).
    Reference semantics: stable LSD sort carrying ValueT, execution/ExecutionPolicy.hpp:530-608; results are unique (bit-exact)."""
    import zpc_amd as zs
    g = np.random.default_rng(n + len(case))
    sbit, ebit = 0, 32
    if case == "constant_keys":
        k = np.full(n, 12345, np.int32)
    elif case == "window_over_equal_bits":
        k = g.integers(0, 1 << 14, n, dtype=np.int32)     # bits 16..31 agree
        sbit, ebit = 16, 32
    else:
        k = g.integers(-2 ** 30, 2 ** 30, n, dtype=np.int32)
    keys = torch.from_numpy(k.copy()).cuda()
    iota = torch.arange(n, dtype=torch.int32, device="cuda")
    perm = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    zs.radix_sort_pair(pol, keys, iota, keys, perm, sbit=sbit, ebit=ebit)
    pol.syncCtx()
    kk = (k.astype(np.int64) >> sbit) & ((1 << (ebit - sbit)) - 1)
    if ebit == 32:   # signed order on the top bit
        kk = np.where(kk >= (1 << (ebit - sbit - 1)), kk - (1 << (ebit - sbit)), kk)
    o = np.argsort(kk, kind="stable")
    assert np.array_equal(perm.cpu().numpy(), o.astype(np.int32))
    assert np.array_equal(keys.cpu().numpy(), k[o])
