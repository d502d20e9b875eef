"""Multi-GPU domain decomposition for the MPM transfer path: one process per GPU, torch.distributed
("nccl" == RCCL on ROCm) for the ghost-block halo exchange over xGMI.

The reference has no collective layer at all (only per-device contexts + peer access), so this is
new, designed for the MI355X node topology: 8 GPUs, fully connected point-to-point xGMI links.  A spatial
split into 1/2/4/8 boxes gives every rank at most 7 peers -- one direct link each -- so the exchange is a single
grouped send/recv (ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd via batch_isend_irecv), never a ring
collective.

Exchange scheme (one phase per step): every rank runs P2G for its own particles into its own copy of every
grid block its particles can touch (owned + ghost blocks).  For each pair of ranks the set of blocks BOTH hold
is fixed at partition-build time; after P2G the two ranks swap their partial sums {m, mv, rhs} for exactly
those blocks and add the peer's partials.  A block held by k ranks ends up with the k-way total on each
holder, so the grid update and G2P need no second exchange.  (Totals on different holders may differ in the
last ulp -- the additions happen in a different order -- which is inside the stated P2G tolerance.)
"""
import numpy as np


_SPLIT_OVERRIDE = {}


def set_split_dims(world_size, dims):
    """Use `dims` = (dx, dy, dz), dx * dy * dz == world_size, as the rank grid for this world size."""
    assert int(dims[0]) * int(dims[1]) * int(dims[2]) == world_size
    _SPLIT_OVERRIDE[world_size] = tuple(int(d) for d in dims)


def split_dims(world_size):
    """Rank grid.  8 ranks: the 2 x 2 x 2 split SURVEY.md 8(e) / BASELINE config 4 name (every pair of ranks is one direct xGMI link; up
    to seven peers per rank).  Other sizes: slabs along y, the long axis of the column, (1, N, 1): two neighbours per rank.
    `set_split_dims` / bench.py --decomp override both."""
    if world_size in _SPLIT_OVERRIDE:
        return _SPLIT_OVERRIDE[world_size]
    if world_size == 8:
        return (2, 2, 2)
    return (1, world_size, 1)


def rank_coords(rank, dims):
    return (rank // (dims[1] * dims[2]), (rank // dims[2]) % dims[1], rank % dims[2])


def cell_box(rank, world_size, lo, hi, align=1):
    """Sub-box [lo_r, hi_r) of cell coordinates owned by `rank`: the global box [lo, hi) cut at planes aligned
    to `align` cells (block side) so that grid blocks are never split between owners."""
    dims = split_dims(world_size)
    rc = rank_coords(rank, dims)
    blo, bhi = [], []
    for d in range(3):
        n = hi[d] - lo[d]
        cuts = [lo[d]]
        for k in range(1, dims[d]):
            c = lo[d] + (n * k) // dims[d]
            c = (c // align) * align
            cuts.append(c)
        cuts.append(hi[d])
        blo.append(cuts[rc[d]])
        bhi.append(cuts[rc[d] + 1])
    return tuple(blo), tuple(bhi)


def shared_keys(my_keys, peer_keys):
    """Lexicographically sorted intersection of two [n,3] int32 block-key arrays (identical on both sides)."""
    if my_keys.shape[0] == 0 or peer_keys.shape[0] == 0:
        return np.zeros((0, 3), np.int32)
    dt = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32)])
    a = np.ascontiguousarray(my_keys.astype(np.int32)).view(dt).reshape(-1)
    b = np.ascontiguousarray(peer_keys.astype(np.int32)).view(dt).reshape(-1)
    s = np.intersect1d(a, b)  # sorted by (x, y, z)
    return s.view(np.int32).reshape(-1, 3)


def gather_block_keys(dist, world_size, my_keys, backend_dev):
    """all_gather of the ranks' [n,3] int32 block-key lists (variable length -> padded to the longest); list of numpy arrays."""
    import torch
    n = np.array([my_keys.shape[0]], np.int64)
    cnt_t = torch.from_numpy(n).to(backend_dev)
    counts = [torch.zeros_like(cnt_t) for _ in range(world_size)]
    dist.all_gather(counts, cnt_t)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    mine = torch.zeros(mx, 3, dtype=torch.int32, device=backend_dev)
    if my_keys.shape[0]:
        mine[: my_keys.shape[0]] = torch.from_numpy(np.ascontiguousarray(my_keys.astype(np.int32))).to(backend_dev)
    allk = [torch.zeros_like(mine) for _ in range(world_size)]
    dist.all_gather(allk, mine)
    return [allk[p][: counts[p]].cpu().numpy() for p in range(world_size)]


def near_shared_mask(my_keys, all_keys, rank, stride, margin=2):
    """mask[i] = 1 when block my_keys[i] is within `margin` blocks (Chebyshev, key units of `stride`) of a block that also
    exists on another rank.  These blocks are numbered and launched first; the others may run while the ghost sums travel:
    a block's bins write to the blocks at offsets {0,1}^3, their exact-path particles (<= one 4^3 bin away, enforced by the
    drift flag of zs_rocm_mpm_g2p2g_range) to offsets {-1..2}^3 when a block is one bin (side 4: margin 2) and to
    {-1..1}^3 when a block holds 2^3 bins (side 8: base nodes -4..+11 of the block, stencils up to +13 < 16: margin 1)."""
    if my_keys.shape[0] == 0:
        return np.zeros(0, bool)
    sh = [shared_keys(my_keys, all_keys[p]) for p in range(len(all_keys)) if p != rank]
    sh = [x for x in sh if x.shape[0]]
    if not sh:
        return np.zeros(my_keys.shape[0], bool)
    sh = np.unique(np.concatenate(sh), axis=0).astype(np.int64) // stride
    r = np.arange(-margin, margin + 1)
    off = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3)
    near = (sh[:, None, :] + off[None, :, :]).reshape(-1, 3)
    B = 1 << 20  # block coordinates of a 512^3 .. 4096^3 grid fit comfortably

    def code(k):
        k = k + B // 2
        return (k[:, 0] * B + k[:, 1]) * B + k[:, 2]

    return np.isin(code(my_keys.astype(np.int64) // stride), np.unique(code(near)))


class HaloExchange:
    """Ghost-block partial-sum exchange.  `lookup(keys[n,3]) -> local block numbers` , `pack(blocks_t, nb, buf)`,
    `unpack_add(blocks_t, nb, buf)` are supplied by the caller (HIP kernels in production: zs_rocm_query__bht /
    zs_rocm_mpm_halo_pack / zs_rocm_mpm_halo_unpack)."""

    def __init__(self, dist, rank, world_size, my_keys, lookup, make_index_tensor, make_buffer, block_floats, all_keys=None):
        self.dist, self.rank, self.world = dist, rank, world_size
        self.peers = []  # (peer, offset, nb) into the concatenated block list / buffers
        if world_size == 1:
            return
        if all_keys is None:  # every rank learns every rank's key list
            all_keys = gather_block_keys(dist, world_size, my_keys, make_buffer(1).device)
        found = []
        for p in range(world_size):
            if p == rank:
                continue
            sk = shared_keys(my_keys, all_keys[p])
            if sk.shape[0] == 0:
                continue
            local = lookup(sk)
            assert (local >= 0).all()
            found.append((p, local.astype(np.int32)))
        # one concatenated block list / send buffer / receive buffer: ONE pack launch and ONE unpack launch per exchange,
        # the per-peer messages are contiguous slices of the two buffers
        total = sum(x[1].shape[0] for x in found)
        self.total_blocks = total
        self.block_floats = block_floats
        if total:
            self.blocks_all = make_index_tensor(np.concatenate([x[1] for x in found]))
            self.sendbuf = make_buffer(total * block_floats)
            self.recvbuf = make_buffer(total * block_floats)
            off = 0
            for p, local in found:
                nb = local.shape[0]
                self.peers.append((p, off, nb))
                off += nb
        self.bytes_per_exchange = total * block_floats * 4

    def exchange(self, pack, unpack_add):
        """pack(blocks, nb, buf): buf[i] = grid[blocks[i]];  unpack_add(blocks, nb, buf): grid[blocks[i]] += buf[i] with
        ATOMIC adds (a corner block shared with several peers is listed once per peer)."""
        if not self.peers:
            return
        d = self.dist
        bf = self.block_floats
        pack(self.blocks_all, self.total_blocks, self.sendbuf)
        if getattr(self, "_ops", None) is None:  # the message list is fixed for the life of the partition
            self._ops = []
            for p, off, nb in self.peers:
                self._ops.append(d.P2POp(d.isend, self.sendbuf[off * bf:(off + nb) * bf], p))
                self._ops.append(d.P2POp(d.irecv, self.recvbuf[off * bf:(off + nb) * bf], p))
        for w in d.batch_isend_irecv(self._ops):
            w.wait()
        unpack_add(self.blocks_all, self.total_blocks, self.recvbuf)

    def exchange_native(self, comm, pol, grid, side, chn0=0, nchn=7):
        """the same exchange through zs_rocm_dist_halo_exchange: pack kernel, grouped ncclSend / ncclRecv, atomic unpack-add, all
        enqueued on `pol`'s stream by the library (no Python between the three)"""
        if not self.peers:
            return
        import ctypes as C
        from ._lib import lib
        if getattr(self, "_native_args", None) is None:
            n = len(self.peers)
            self._native_args = ((C.c_int * n)(*[p for p, _, _ in self.peers]), (C.c_size_t * n)(*[o for _, o, _ in self.peers]),
                                 (C.c_size_t * n)(*[c for _, _, c in self.peers]), n)
        pr, po, pc, n = self._native_args
        if lib().zs_rocm_dist_halo_exchange(comm._h, pol.handle, grid.data_ptr(), int(side), chn0, nchn, self.blocks_all.data_ptr(),
                                            self.total_blocks, n, pr, po, pc, self.sendbuf.data_ptr(), self.recvbuf.data_ptr()) != 0:
            raise RuntimeError("zs_rocm_dist_halo_exchange failed")


def halo_plan_from_keys(all_keys, rank):
    """zs_rocm_halo_plan_from_keys (host part of the native halo plan, no GPU needed): list of (peer, offset, count) and the
    concatenated local block numbers, from every rank's [n,3] key list"""
    import ctypes as C
    from ._lib import lib
    world = len(all_keys)
    counts = (C.c_size_t * world)(*[int(k.shape[0]) for k in all_keys])
    cat = np.ascontiguousarray(np.concatenate([np.asarray(k, np.int32).reshape(-1, 3) for k in all_keys]), dtype=np.int32)
    npeers = C.c_int(0)
    pr, po, pc = (C.c_int * max(world, 1))(), (C.c_size_t * max(world, 1))(), (C.c_size_t * max(world, 1))()
    total = lib().zs_rocm_halo_plan_from_keys(cat.ctypes.data, counts, world, rank, C.byref(npeers), None, None, None, None)
    blocks = np.zeros(max(total, 1), np.int32)
    lib().zs_rocm_halo_plan_from_keys(cat.ctypes.data, counts, world, rank, C.byref(npeers), pr, po, pc, blocks.ctypes.data)
    return [(int(pr[k]), int(po[k]), int(pc[k])) for k in range(npeers.value)], blocks[:total]


class NativeHaloPlan:
    """zs_rocm_halo_plan: key all-gather (RCCL), plan and exchange buffers all inside libzsrocm.so; same face as HaloExchange"""

    def __init__(self, comm, pol, table, nblocks, side):
        from ._lib import lib
        L = lib()
        keys = table.view().activeKeys if hasattr(table, "view") else table
        self._h = L.zs_rocm_dist_halo_plan_create(comm._h, pol.handle, keys, nblocks, side)
        if not self._h:
            raise RuntimeError("zs_rocm_dist_halo_plan_create failed")
        self.total_blocks = L.zs_rocm_dist_halo_plan_blocks(self._h)
        self.bytes_per_exchange = L.zs_rocm_dist_halo_plan_bytes(self._h)
        self.peers = [None] * L.zs_rocm_dist_halo_plan_npeers(self._h)  # (only their number is needed on this side)

    @classmethod
    def from_lists(cls, comm, side, peers, blocks):
        """plan from explicit lists: peers = [(rank, offset, count)], blocks = int32 array of local block numbers (zs_rocm_dist_halo_plan_from_lists)"""
        import ctypes as C
        import numpy as np
        from ._lib import lib
        L = lib()
        self = cls.__new__(cls)
        npeers = len(peers)
        pr = (C.c_int * max(npeers, 1))(*[p[0] for p in peers])
        po = (C.c_size_t * max(npeers, 1))(*[p[1] for p in peers])
        pc = (C.c_size_t * max(npeers, 1))(*[p[2] for p in peers])
        b = np.ascontiguousarray(blocks, np.int32)
        self._h = L.zs_rocm_dist_halo_plan_from_lists(comm._h, int(side), npeers, pr, po, pc, b.ctypes.data_as(C.POINTER(C.c_int)), b.shape[0])
        if not self._h:
            raise RuntimeError("zs_rocm_dist_halo_plan_from_lists failed")
        self.total_blocks = L.zs_rocm_dist_halo_plan_blocks(self._h)
        self.bytes_per_exchange = L.zs_rocm_dist_halo_plan_bytes(self._h)
        self.peers = [None] * npeers
        return self

    def __del__(self):
        try:
            from ._lib import lib
            if self._h:
                lib().zs_rocm_dist_halo_plan_destroy(self._h)
        except Exception:
            pass

    def exchange_native(self, comm, pol, grid, side, chn0=0, nchn=7):
        from ._lib import lib
        if lib().zs_rocm_dist_halo_plan_exchange(self._h, comm._h, pol.handle, grid.data_ptr(), chn0, nchn) != 0:
            raise RuntimeError("zs_rocm_dist_halo_plan_exchange failed")


class NativeComm:
    """Thin mirror of zs_rocm_dist (zpc_amd/csrc/dist.hip): the RCCL communicator of this process.  The unique id is created on
    rank 0 and handed round by whatever the launcher offers -- here a torch.distributed broadcast; a C++ host passes the bytes
    itself (file, MPI, environment)."""

    def __init__(self, rank=0, world=1, device=0, dist=None, bcast_device=None):
        import ctypes as C
        import torch
        from ._lib import lib
        L = lib()
        nb = L.zs_rocm_dist_unique_id_bytes()
        buf = (C.c_ubyte * nb)()
        if rank == 0:
            if L.zs_rocm_dist_unique_id(buf) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
        if world > 1:
            t = torch.tensor(list(buf), dtype=torch.uint8, device=bcast_device or torch.device("cuda", device))
            dist.broadcast(t, 0)
            buf = (C.c_ubyte * nb)(*t.cpu().tolist())
        self.rank, self.world = rank, world
        self._h = L.zs_rocm_dist_create(rank, world, buf, device)
        if not self._h:
            raise RuntimeError("zs_rocm_dist_create (ncclCommInitRank) failed")

    def __del__(self):
        try:
            from ._lib import lib
            if self._h:
                lib().zs_rocm_dist_destroy(self._h)
        except Exception:
            pass

    def allreduce(self, pol, t, op="sum"):
        """in place, on the policy's stream; float32 or int64 tensors"""
        import torch
        from ._lib import lib
        code = {"sum": 0, "max": 1, "min": 2}[op]
        f = lib().zs_rocm_dist_allreduce_f32 if t.dtype == torch.float32 else lib().zs_rocm_dist_allreduce_i64
        if f(self._h, pol.handle, t.data_ptr(), t.numel(), code) != 0:
            raise RuntimeError("RCCL allreduce failed")

    def alltoall_counts(self, pol, send):
        import torch
        from ._lib import lib
        recv = torch.empty_like(send)
        if lib().zs_rocm_dist_alltoall_i64(self._h, pol.handle, send.data_ptr(), recv.data_ptr()) != 0:
            raise RuntimeError("RCCL counts all-to-all failed")
        return recv

    def alltoallv(self, pol, send, send_counts, recv, recv_counts):
        """float32 buffers; counts in floats per rank (python lists)"""
        import ctypes as C
        from ._lib import lib
        w = self.world
        so, ro, a, b = [0], [0], 0, 0
        for r in range(w - 1):
            a += send_counts[r]
            b += recv_counts[r]
            so.append(a)
            ro.append(b)
        A = lambda v: (C.c_size_t * w)(*[int(x) for x in v])
        if lib().zs_rocm_dist_alltoallv_f32(self._h, pol.handle, send.data_ptr(), A(send_counts), A(so), recv.data_ptr(), A(recv_counts),
                                            A(ro)) != 0:
            raise RuntimeError("RCCL all-to-all failed")

    def barrier(self, pol):
        from ._lib import lib
        lib().zs_rocm_dist_barrier(self._h, pol.handle)


def migrate_particles(mt, pol, dist, rank, world_size, glo, ghi, align, to_comm=None, from_comm=None, comm=None):
    """Move every particle to the rank that owns the cell it is in now.  `mt` is a MpmTransfer whose full
    particle state is in memory (after g2p / g2p2g(write_all=True)).  Device work is done by libzsrocm kernels:
    owner classification, a stable radix partition of the particle ids by destination, AoSoA -> AoS row gather into the send
    buffer, AoSoA compaction of the kept rows and AoS -> AoSoA scatter of the received rows; torch.distributed moves the
    bytes (all_to_all_single with uneven splits = ncclSend/ncclRecv groups over the xGMI links).
    Returns (n_sent, n_received).  After it the caller rebuilds partition, halo lists and bins."""
    import ctypes as C
    import torch
    from ._lib import lib
    from .primitives import radix_sort_pair
    L = lib()
    n, nchn, lw, dev = mt.n, mt.nchn, mt.L, mt.device
    if world_size == 1:
        return 0, 0
    dims = split_dims(world_size)
    owner = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    lo3, hi3, d3 = (C.c_int * 3)(*glo), (C.c_int * 3)(*ghi), (C.c_int * 3)(*dims)
    L.zs_rocm_mpm_owner_rank(pol.handle, mt._port("x"), n, mt.params.dx, lo3, hi3, d3, int(align), owner.data_ptr())
    ids = torch.arange(max(n, 1), dtype=torch.int32, device=dev)
    so, order = torch.empty_like(owner), torch.empty_like(ids)
    bits = max(1, (world_size - 1).bit_length())
    if n:
        radix_sort_pair(pol, owner, ids, so, order, n=n, sbit=0, ebit=bits)  # stable: ids ascend inside a destination
    cnt_dev = torch.empty(world_size, dtype=torch.int32, device=dev)
    L.zs_rocm_mpm_owner_counts(pol.handle, owner.data_ptr(), n, world_size, cnt_dev.data_ptr())  # (the library's own count, on the policy's stream)
    pol.syncCtx()
    counts = cnt_dev.cpu().tolist()
    starts = [0]
    for c in counts:
        starts.append(starts[-1] + c)
    keep = order[starts[rank]:starts[rank + 1]]
    leave = torch.cat([order[starts[p]:starts[p + 1]] for p in range(world_size) if p != rank]) if n else order[:0]
    n_keep, n_leave = int(keep.numel()), int(leave.numel())
    send_rows = [counts[p] if p != rank else 0 for p in range(world_size)]
    sendbuf = torch.empty(max(n_leave, 1) * nchn, dtype=torch.float32, device=dev)
    if n_leave:
        leave = leave.contiguous()
        L.zs_rocm_tv_gather_rows_f32(pol.handle, mt.buf.data_ptr(), leave.data_ptr(), n_leave, nchn, lw, sendbuf.data_ptr())
    pol.syncCtx()
    to_comm = to_comm or (lambda t: t)
    from_comm = from_comm or (lambda t: t)
    if comm is not None:  # RCCL through the C ABI (zs_rocm_dist_alltoall_i64 / _alltoallv_f32)
        sc = torch.tensor(send_rows, dtype=torch.int64, device=dev)
        rc = comm.alltoall_counts(pol, sc)
        pol.syncCtx()
        recv_rows = [int(x) for x in rc.cpu().tolist()]
        n_recv = sum(recv_rows)
        recvbuf = torch.empty(max(n_recv, 1) * nchn, dtype=torch.float32, device=dev)
        comm.alltoallv(pol, sendbuf, [r * nchn for r in send_rows], recvbuf, [r * nchn for r in recv_rows])
        pol.syncCtx()
    else:
        sc = to_comm(torch.tensor(send_rows, dtype=torch.int64, device=dev))
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc)
        recv_rows = [int(x) for x in rc.cpu().tolist()]
        n_recv = sum(recv_rows)
        recvbuf = to_comm(torch.empty(max(n_recv, 1) * nchn, dtype=torch.float32, device=dev))
        dist.all_to_all_single(recvbuf[: n_recv * nchn], to_comm(sendbuf)[: n_leave * nchn], [r * nchn for r in recv_rows],
                               [r * nchn for r in send_rows])
        recvbuf = from_comm(recvbuf)
    n_new = n_keep + n_recv
    tiles = (n_new + lw - 1) // lw
    newbuf = torch.zeros(max(tiles, 1) * lw * nchn, dtype=torch.float32, device=dev)
    if n_keep:
        keep = keep.contiguous()
        L.zs_rocm_tv_gather_f32(pol.handle, mt.buf.data_ptr(), newbuf.data_ptr(), n_keep, nchn, lw, keep.data_ptr())
    if n_recv:
        L.zs_rocm_tv_scatter_rows_f32(pol.handle, recvbuf.data_ptr(), n_recv, nchn, lw, newbuf.data_ptr(), n_keep)
    pol.syncCtx()
    mt.set_particles(newbuf[: tiles * lw * nchn] if tiles else newbuf[:0], n_new)
    return n_leave, n_recv
