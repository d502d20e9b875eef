"""GPU parity for index_buckets_for_particles (simulation/particle/Query.tpp:9-58): per cell KEY the bucket holds exactly the
oracle's particle-id sequence (ascending ids = the sequential policy's result); counts / offsets consistent."""
import ctypes as C

import numpy as np
import pytest

from util import rng

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _d2h(ptr, n):
    out = np.empty(n, np.int32)
    C.CDLL("libamdhip64.so").hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2)
    return out


@pytest.mark.parametrize("n,dx,disp,aosoa", [(1, 0.1, 0.5, False), (5000, 0.05, 0.5, False), (200_000, 1 / 64, 0.0, False), (100_000, 1 / 32, 0.5, True)])
def test_index_buckets_match_oracle_by_cell(pol, oracle, n, dx, disp, aosoa):
    from zpc_amd import Port
    from zpc_amd.containers import IndexBuckets
    from zpc_amd._lib import HashTableView, lib
    g = rng(170)
    pos = g.uniform(-0.4, 0.9, (n, 3)).astype(np.float32)
    if aosoa:  # positions as channels 1..3 of a TileVector<f32,32> with 5 channels
        L, Cn = 32, 5
        tiles = (n + L - 1) // L
        buf = np.zeros((tiles, Cn, L), np.float32)
        idx = np.arange(n)
        for d in range(3):
            buf[idx // L, 1 + d, idx % L] = pos[:, d]
        dev = torch.from_numpy(buf.reshape(-1)).cuda()
        port = Port(dev.data_ptr() + 1 * L * 4, 0, 5, L - 1, Cn)
    else:
        dev = torch.from_numpy(pos).cuda()
        port = Port(dev.data_ptr(), 0, 0, 0, 3)  # AoS vec3: component stride 1, element stride 3
    ib = IndexBuckets()
    ib.build(pol, port, n, dx, disp, expected_cells=n)
    pol.syncCtx()
    v = ib.view()
    nb = v.numBuckets
    counts, offsets, indices = _d2h(v.counts, nb + 1), _d2h(v.offsets, nb + 1), _d2h(v.indices, n)
    hv = HashTableView()
    lib().zs_rocm_hashtable_get_view(v.table, C.byref(hv))
    keys = _d2h(hv.activeKeys, nb * 3).reshape(nb, 3)
    # oracle
    oracle.orc_index_buckets_for_particles.restype = C.c_void_p
    oracle.orc_hashtable_active_keys.restype = C.POINTER(C.c_int32)
    pc, po, pi = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    t = C.c_void_p(oracle.orc_index_buckets_for_particles(pos.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_float(dx), C.c_float(disp),
                                                           C.c_size_t(n), C.byref(pc), C.byref(po), C.byref(pi)))
    onb = oracle.orc_hashtable_size(t)
    assert nb == onb
    okeys = np.ctypeslib.as_array(oracle.orc_hashtable_active_keys(t), shape=(onb, 3)).copy()
    oc, oo, oi = (np.ctypeslib.as_array(p, shape=(s,)).copy() for p, s in ((pc, onb + 1), (po, onb + 1), (pi, n)))
    assert counts[nb] == 0 and offsets[nb] == n and np.array_equal(offsets, np.concatenate([[0], np.cumsum(counts[:-1])]))
    ref = {tuple(k): oi[oo[i]:oo[i] + oc[i]] for i, k in enumerate(okeys)}
    assert len(ref) == nb
    for i, k in enumerate(keys):
        assert np.array_equal(indices[offsets[i]:offsets[i] + counts[i]], ref[tuple(k)])
    # keys really are the cells of their particles
    cell = np.floor(pos * np.float32(1.0 / dx) + np.float32(disp)).astype(np.int32)
    assert np.array_equal(cell[indices], np.repeat(keys, counts[:-1], axis=0))
    oracle.orc_index_buckets_free(pc, po, pi)
    oracle.orc_hashtable_destroy(t)
