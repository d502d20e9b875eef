"""Boundary pass of the MPM sub-step on the GPU: zs_rocm_collider_resolve (bulk Collider::resolveCollision) against the golden
vectors produced by the reference's own level-set code, and zs_rocm_mpm_apply_boundary (ApplyBoundaryConditionOnGridBlocks,
simulation/grid/GridOp.hpp:111-164) against the oracle on a real partition / grid."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from util import make_cloud, collider_struct

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_resolve_collision_matches_reference_golden(pol):
    from zpc_amd import lib
    from zpc_amd._lib import Collider
    g = np.load(os.path.join(GOLD, "collider.npz"))
    for k, cs in enumerate(g["cases"]):
        c = Collider.from_buffer_copy(bytes(collider_struct(cs)))
        x = torch.from_numpy(np.ascontiguousarray(g["x"][k])).cuda()
        v = torch.from_numpy(np.ascontiguousarray(g["v"][k])).cuda()
        ins = torch.zeros(x.shape[0], dtype=torch.int32, device="cuda")
        lib().zs_rocm_collider_resolve(pol.handle, C.byref(c), x.data_ptr(), v.data_ptr(), x.shape[0], ins.data_ptr())
        pol.syncCtx()
        assert np.array_equal(ins.cpu().numpy(), g["inside"][k])
        # same operations in the same order with contraction off and correctly rounded sqrt / divide: bit for bit
        assert np.array_equal(v.cpu().numpy(), g["v_out"][k]), (k, cs[:2], np.abs(v.cpu().numpy() - g["v_out"][k]).max())


@pytest.mark.parametrize("side,origin", [(4, False), (8, False), (8, True)])
@pytest.mark.parametrize("case", [1, 9, 17, 21])  # moving plane (sticky), cuboid (slip), sphere (separate), cylinder (slip)
def test_apply_boundary_on_grid_matches_oracle(pol, oracle, side, origin, case):
    from zpc_amd.mpm import MpmTransfer
    dx, dt = 1.0 / 64, 1e-4
    mass, pos, vel, Cm, F = make_cloud(10, dx, 4, seed=5 + case)
    pos = (pos - pos.mean(0)).astype(np.float32) * np.float32(1.5)      # around the origin, where the golden colliders sit
    n = pos.shape[0]
    mt = MpmTransfer(pol, n, dx, dt, model=0, side=side, volume=dx ** 3 / 4, key_is_origin=origin)
    mt.upload(mass, pos, vel, Cm, F)
    mt.build_partition(4096)
    mt.rebin()
    mt.clear_grid()
    mt.p2g()
    mt.grid_update((0.0, -9.8, 0.0))
    pol.syncCtx()
    before = mt.grid.cpu().numpy().copy()
    keys = mt.active_keys()
    g = np.load(os.path.join(GOLD, "collider.npz"))
    from zpc_amd._lib import Collider
    cs = g["cases"][case]
    mt.apply_boundary(Collider.from_buffer_copy(bytes(collider_struct(cs))))
    pol.syncCtx()
    got = mt.grid.cpu().numpy()
    ref = before.copy()
    oc = collider_struct(cs)
    oracle.orc_mpm_apply_boundary(C.byref(oc), np.ascontiguousarray(keys).ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p),
                                  C.c_size_t(mt.nblocks), side, side if origin else 1, C.c_float(dx))
    assert np.array_equal(got, ref)
    changed = (ref != before).reshape(mt.nblocks, 7, side ** 3)
    assert changed[:, 1:4].any() and not changed[:, 0].any() and not changed[:, 4:].any()  # only velocities, and some of them
