#!/usr/bin/env python3
"""RCCL smoke test of the exact torch.distributed calls the multi-GPU path makes, on ONE GPU (world_size 1, peer = self):
all_gather of int64 / int32 tensors, all_to_all_single with split sizes, grouped isend/irecv issued under a side stream with
event hand-over, barrier.  It cannot prove multi-GPU transport, but it does catch API / dtype / stream-usage mistakes on the
backend the driver's N > 1 runs use.  python tools/nccl_selftest.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zpc_amd.dist import HaloExchange, gather_block_keys  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
keys = np.array([[0, 0, 0], [1, 0, 0], [2, 5, -3]], np.int32)
allk = gather_block_keys(dist, 1, keys, dev)
assert len(allk) == 1 and np.array_equal(allk[0], keys)
# all_to_all_single with explicit splits (migrate_particles)
sc = torch.tensor([7], dtype=torch.int64, device=dev)
rc = torch.empty_like(sc)
dist.all_to_all_single(rc, sc)
assert int(rc.item()) == 7
send = torch.arange(7 * 26, dtype=torch.float32, device=dev)
recv = torch.empty_like(send)
dist.all_to_all_single(recv, send, [7 * 26], [7 * 26])
assert torch.equal(recv, send)
empty_out, empty_in = recv[:0], send[:0]
dist.all_to_all_single(empty_out, empty_in, [0], [0])
# halo exchange with myself as the only peer, issued on a high-priority side stream after an event (bench.py step_fused)
bf = 7 * 64
grid = torch.rand(3, bf, device=dev)
h = HaloExchange.__new__(HaloExchange)
h.dist, h.rank, h.world = dist, 0, 1
h.block_floats, h.total_blocks = bf, 3
h.blocks_all = torch.arange(3, dtype=torch.int32, device=dev)
h.sendbuf = torch.empty(3 * bf, device=dev)
h.recvbuf = torch.zeros(3 * bf, device=dev)
h.peers = [(0, 0, 2), (0, 2, 1)]  # two messages, like two peers
before = grid.clone()
comm = torch.cuda.Stream(device=dev, priority=-1)
ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()
for rep in range(3):  # the message list (P2POp objects) is built once and re-used every step
    ev0.record()
    with torch.cuda.stream(comm):
        comm.wait_event(ev0)
        h.exchange(lambda b, nb, buf: buf.copy_(grid.reshape(-1)), lambda b, nb, buf: grid.add_(buf.reshape(3, bf)))
        ev1.record()
    torch.cuda.current_stream().wait_event(ev1)
torch.cuda.synchronize()
assert torch.allclose(grid, 8 * before)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("nccl selftest ok")
