#!/usr/bin/env python3
"""The exchange steps of libzsrocm.so's zs_rocm_dist_* (zpc_amd/csrc/dist.hip: RCCL called from C++, no torch.distributed) on ONE GPU:
world size 1, the only peer is this rank.  Checks the communicator life cycle, allreduce (sum / max / min, f32 and i64), the counts
all-to-all, the uneven all-to-all, the ghost-block exchange (pack kernel -> grouped ncclSend / ncclRecv -> atomic unpack-add) on the
null stream and on a side stream, and the barrier.  It cannot prove multi-GPU transport; it does prove the library's RCCL calls, dtypes,
stream use and buffer arithmetic.     python tools/rccl_native_selftest.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpc_amd  # noqa: E402
from zpc_amd.dist import HaloExchange, NativeComm  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
pol = zpc_amd.rocm_exec().sync(False).external_stream(torch.cuda.current_stream().cuda_stream)
comm = NativeComm(0, 1, 0)
assert zpc_amd.lib().zs_rocm_dist_rank(comm._h) == 0 and zpc_amd.lib().zs_rocm_dist_world(comm._h) == 1

t = torch.tensor([1.5, -2.0, 7.25], dtype=torch.float32, device=dev)
for op in ("sum", "max", "min"):
    u = t.clone()
    comm.allreduce(pol, u, op)
    pol.syncCtx()
    assert torch.equal(u, t), op
i = torch.tensor([1 << 40, -5], dtype=torch.int64, device=dev)
j = i.clone()
comm.allreduce(pol, j, "sum")
pol.syncCtx()
assert torch.equal(i, j)

rc = comm.alltoall_counts(pol, torch.tensor([11], dtype=torch.int64, device=dev))
pol.syncCtx()
assert int(rc.item()) == 11
send = torch.arange(11 * 26, dtype=torch.float32, device=dev)
recv = torch.zeros_like(send)
comm.alltoallv(pol, send, [11 * 26], recv, [11 * 26])
comm.alltoallv(pol, send, [0], recv[:0], [0])  # nobody leaves: no message at all
pol.syncCtx()
assert torch.equal(send, recv)

# ghost-block exchange with myself: blocks 4, 1 and 6 of an 8-block grid, as one message and as two (like two peers sharing block 1)
for side in (4, 8):
    nc = side ** 3
    grid = torch.rand(8, 7, nc, device=dev)
    before = grid.clone()
    h = HaloExchange.__new__(HaloExchange)
    h.peers = [(0, 0, 3)]
    h.blocks_all = torch.tensor([4, 1, 6], dtype=torch.int32, device=dev)
    h.total_blocks = 3
    h.sendbuf = torch.empty(3 * 7 * nc, device=dev)
    h.recvbuf = torch.zeros(3 * 7 * nc, device=dev)
    h._native_args = None
    h.exchange_native(comm, pol, grid, side)
    pol.syncCtx()
    want = before.clone()
    want[[4, 1, 6]] *= 2
    assert torch.equal(grid, want), side
    # only the momentum channels, on a high-priority side stream after an event (the overlapped step of bench.py)
    side_stream = torch.cuda.Stream(device=dev, priority=-1)
    pol2 = zpc_amd.rocm_exec().sync(False).external_stream(side_stream.cuda_stream)
    ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()
    h2 = HaloExchange.__new__(HaloExchange)
    h2.peers = [(0, 0, 2), (0, 2, 2)]
    h2.blocks_all = torch.tensor([4, 1, 1, 6], dtype=torch.int32, device=dev)
    h2.total_blocks = 4
    h2.sendbuf = torch.empty(4 * 3 * nc, device=dev)
    h2.recvbuf = torch.zeros(4 * 3 * nc, device=dev)
    h2._native_args = None
    for rep in range(2):
        ev0.record()
        with torch.cuda.stream(side_stream):
            side_stream.wait_event(ev0)
            h2.exchange_native(comm, pol2, grid, side, chn0=1, nchn=3)
            ev1.record()
        torch.cuda.current_stream().wait_event(ev1)
    torch.cuda.synchronize()
    # per exchange: block 4 and 6 double, block 1 (listed twice: both messages carry the value from before the exchange) triples
    want[[4, 6], 1:4] *= 4
    want[1, 1:4] *= 9
    assert torch.allclose(grid, want, rtol=1e-6), side

# the halo plan (key all-gather + plan + buffers inside the library): world 1 = nobody to share with
from zpc_amd.dist import NativeHaloPlan  # noqa: E402
keys = torch.tensor([[0, 0, 0], [8, 0, 0], [8, 8, -16]], dtype=torch.int32, device=dev)
plan = NativeHaloPlan(comm, pol, keys.data_ptr(), 3, 8)
assert plan.total_blocks == 0 and len(plan.peers) == 0 and plan.bytes_per_exchange == 0
plan.exchange_native(comm, pol, torch.zeros(3, 7, 512, device=dev), 8)  # no-op
del plan
# ... and from a real partition (the bht of an MpmTransfer), as bench.py builds it
from zpc_amd.mpm import MpmTransfer  # noqa: E402
import numpy as np  # noqa: E402
g = np.random.default_rng(1)
npart = 4000
pos = (0.3 + 0.2 * g.random((npart, 3))).astype(np.float32)
mt = MpmTransfer(pol, npart, 1.0 / 64, 1e-4, model=1, side=8, volume=1e-7, cache_stress=True)
mt.upload(np.full(npart, 1e-3, np.float32), pos, np.zeros((npart, 3), np.float32), np.zeros((npart, 9), np.float32),
          np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (npart, 1)), np.zeros(npart, np.float32))
mt.build_partition(npart, margin=1)
pol.syncCtx()
plan = NativeHaloPlan(comm, pol, mt.table, mt.nblocks, 8)
assert plan.total_blocks == 0 and mt.nblocks > 0
del plan
comm.barrier(pol)
assert zpc_amd.lib().zs_rocm_last_error(0) == 0
del comm
print("rccl native selftest ok")
