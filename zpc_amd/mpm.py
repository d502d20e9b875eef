"""Host-side driver of the MPM particle<->grid hot path: the sequence a downstream zpc application runs per
sub-step (sparsity/partition -> grid reset -> P2G -> grid update -> G2P), expressed over the
C ABI of libzsrocm.  torch only owns device memory and streams here; every kernel is hand-written HIP.

Particle storage is one TileVector<f32, L> (AoSoA) with channels
    m:1  x:3  v:3  C:9  F:9  [logJp:1]
(the attribute set of zs::Particles, geometry/Structurefree.hpp:21-237, in the TileVector layout of
container/TileVector.hpp:108); the grid is TileVector<f32, side^3> with channels {m:1, v:3, rhs:3}
(simulation/mpm/Simulator.cpp:116-122) over blocks keyed in a bht<int,3,int,16>.
"""
import ctypes as C

import torch

from ._lib import lib, Port, Particles, MpmParams, SlotStorage, MpmStep
from .containers import Bht

FIXED_COROTATED, DRUCKER_PRAGER, VONMISES_FIXED_COROTATED, NACC = 0, 1, 2, 3  # ConstitutiveModelConfig members with F
EQUATION_OF_STATE = 4  # the fluid member: particles carry J (1 channel) where the solids carry F (9 channels)
HAS_LOGJP = (DRUCKER_PRAGER, NACC)


class MpmTransfer:
    def __init__(self, pol, n, dx, dt, model=FIXED_COROTATED, side=4, lane_width=64, E=5e4, nu=0.4, volume=1.0,
                 cohesion=0.0, beta=1.0, yield_surface=0.816496580927726 * 2.0 * 0.5 / (3.0 - 0.5), vol_correction=True,
                 device="cuda", key_is_origin=False, aos=False, cache_stress=False, yield_stress=240e6, xi=0.8, friction_angle=45.0,
                 hardening=True, bulk=4e4, viscosity=0.0):
        self.pol, self.n, self.L, self.side = pol, int(n), int(lane_width), int(side)
        self.device = torch.device(device)
        self.model = model
        self.fluid = model == EQUATION_OF_STATE
        self.nF = 1 if self.fluid else 9        # channels of the deformation state: J or F
        self.nchn = 16 + self.nF + (1 if model in HAS_LOGJP else 0)
        self.off = {"m": 0, "x": 1, "v": 4, "C": 7, "F": 16, "logJp": 25}
        # cache_stress: 6 extra channels "PF" hold the symmetric P F^T vol {xx, xy, xz, yy, yz, zz}, written by G2P (and update_stress),
        # read by P2G
        self.cache_stress = bool(cache_stress)
        if self.cache_stress and lib().zs_rocm_mpm_stress_channels() != 6:
            raise RuntimeError("libzsrocm expects %d channels in particles.stress, this mirror allocates 6" % lib().zs_rocm_mpm_stress_channels())
        if self.cache_stress:
            self.off["PF"] = self.nchn
            self.nchn += 6
        self.aos = bool(aos)  # AoS storage (the zs::Particles / zs::Vector<vec<T,N>> form) instead of the AoSoA TileVector
        if self.aos:
            self.L = 1
        self.tiles = (self.n + self.L - 1) // self.L
        self.buf = torch.zeros(self.tiles * self.L * self.nchn, dtype=torch.float32, device=self.device)
        self.buf2 = None  # second buffer for re-binning (ping-pong)
        self.drift_flag = None  # device status words of zs_rocm_mpm_g2p2g_range: [0] split-launch margin violated, [1] exact-path count
        self.drift_tripped = False
        self.outside_tripped = False
        self.key_is_origin = bool(key_is_origin)  # SparseGrid convention: partition keys are block origins (multiples of side)
        self.kstride = side if key_is_origin else 1
        # VonMisesFixedCorotatedConfig::yieldStress; NACCConfig::xi, Msqr() from the friction angle, hardeningOn (beta shared)
        self.params = MpmParams(model, dx, dt, volume, E, nu, cohesion, beta, yield_surface, int(vol_correction), side,
                                int(self.key_is_origin), yield_stress, xi, lib().zs_rocm_nacc_msqr(friction_angle), int(hardening),
                                bulk, viscosity)
        self.table = None
        self.grid = None
        self.nblocks = 0
        self.order = self.bin_start = self.cell_count = self.nbr = None
        self.binned = False
        self.slotted = False   # slotted storage (zs_rocm_mpm_slot_particles): self.buf holds nbins * K tiles, cell_mask says which
        self.n_slots = 0
        self.K = 0
        # run-level record of the slotted storage's status words: folded in by check_slots() (which unslot() calls), so that a
        # re-partition (slot() starts a fresh status buffer) can never discard a latched flag or a lost-mover count
        self.slot_record = {"sent": 0, "homed": 0, "flags": [0] * 5, "periods": 0, "edge_periods": 0, "log": []}

    def _zero(self, t):
        """clear device memory ON THE POLICY'S STREAM (torch's zero_() would run on torch's current stream, which is ordered with
        the policy's stream only while the policy uses the null stream)"""
        lib().zs_rocm_memset(self.pol.handle, t.data_ptr(), 0, t.numel() * t.element_size())

    # ------------------------------------------------------------------ particle access
    def _port(self, name, buf=None):
        buf = self.buf if buf is None else buf
        bits = self.L.bit_length() - 1  # AoS: L == 1 -> numTileBits = tileMask = 0, component stride 1 (GenericIterator.hpp:71-72)
        return Port(buf.data_ptr() + self.off[name] * self.L * 4, 0, bits, self.L - 1, self.nchn)

    def particles(self, buf=None):
        null = Port(None, 0, 0, 0, 1)
        return Particles(self._port("m", buf), self._port("x", buf), self._port("v", buf), self._port("C", buf), self._port("F", buf),
                         self._port("logJp", buf) if self.model in HAS_LOGJP else null,
                         self._port("PF", buf) if self.cache_stress else null, self.n_slots if self.slotted else self.n)

    def set_particles(self, buf, n):
        """Adopt a new AoSoA particle buffer (after an inter-rank migration): the partition and the bins are void."""
        self.n = int(n)
        self.tiles = (self.n + self.L - 1) // self.L
        assert buf.numel() == self.tiles * self.L * self.nchn
        self.buf, self.buf2 = buf, None
        self.order = self.bin_start = self.cell_count = None
        self.binned = False

    def update_stress(self):
        """particles.PF := model(F, logJp) * vol (first step / after host edits of F); no-op without cache_stress."""
        lib().zs_rocm_mpm_update_stress(self.pol.handle, C.byref(self.params), self.particles())

    def upload(self, mass, pos, vel, Cm, F, logJp=None):
        """AoS host/device arrays -> AoSoA particle buffer (zs_rocm_tv_from_aos_f32)."""
        cols = [torch.as_tensor(mass, dtype=torch.float32).reshape(self.n, 1), torch.as_tensor(pos, dtype=torch.float32).reshape(self.n, 3),
                torch.as_tensor(vel, dtype=torch.float32).reshape(self.n, 3), torch.as_tensor(Cm, dtype=torch.float32).reshape(self.n, 9),
                torch.as_tensor(F, dtype=torch.float32).reshape(self.n, self.nF)]   # F, or J for the fluid
        if self.model in HAS_LOGJP:
            lj = torch.zeros(self.n) if logJp is None else torch.as_tensor(logJp, dtype=torch.float32)
            cols.append(lj.reshape(self.n, 1))
        if self.cache_stress:
            cols.append(torch.zeros(self.n, 6))
        aos = torch.cat([c.to(self.device) for c in cols], dim=1).contiguous()
        lib().zs_rocm_tv_from_aos_f32(self.pol.handle, aos.data_ptr(), self.n, self.nchn, self.L, self.buf.data_ptr())
        self.pol.syncCtx()
        self.binned = False

    def download(self):
        buf = self.buf
        if self.slotted:  # the occupied slots, in slot order
            buf, cnt = self._compact_copy()
            assert cnt == self.n, "slotted storage holds %d particles, expected %d" % (cnt, self.n)
        aos = torch.empty(self.n, self.nchn, dtype=torch.float32, device=self.device)
        lib().zs_rocm_tv_to_aos_f32(self.pol.handle, buf.data_ptr(), self.n, self.nchn, self.L, aos.data_ptr())
        self.pol.syncCtx()
        a = aos.cpu().numpy()
        out = {"m": a[:, 0].copy(), "x": a[:, 1:4].copy(), "v": a[:, 4:7].copy(), "C": a[:, 7:16].copy(),
               "J" if self.fluid else "F": a[:, 16:16 + self.nF].copy()}
        if self.model in HAS_LOGJP:
            out["logJp"] = a[:, 25].copy()
        return out

    # ------------------------------------------------------------------ partition (SparsityCompute.tpp:5-24)
    def build_partition(self, expected_blocks, margin=0, order=None, axes=None):
        """ComputeSparsity + EnlargeSparsity{0, 2} (the reference's partition); margin = m enlarges by m more blocks on every side
        (lo = -m, hi = 2 + m): room for the particles to travel m blocks before the partition has to be rebuilt.

        order = how the blocks are numbered (the reference leaves it to the race of the inserting threads):
          "insertion"         the table's own dense indices (the race)
          None / "holders_lex" the blocks that hold particles in lexicographic key order, the apron blocks behind them (in the order
                              of their race: they hold no particle, the per-block kernels leave them at once)
          "lex"               every block in lexicographic key order
          "morton"            every block along the Z-order curve
        axes (holders_lex / lex): the key components from most to least significant, default (0, 1, 2); the last one changes fastest
        along the numbering.  The fused block kernel runs 1.3 % faster with the cloud's longest axis last (bench.py passes that).
        The numbering changes no result beyond the order of the atomic sums, only which workgroups run side by side
        (profiles/r06_p2g.md, section 4b)."""
        import os
        order = os.environ.get("ZS_ROCM_CANONICAL_PARTITION") or order or "holders_lex"   # (the variable: tools/r06_order.sh's A/B)
        if order not in ("insertion", "holders_lex", "lex", "morton"):
            raise ValueError(f"build_partition: unknown block order {order!r}")
        self.table = Bht(3, int(expected_blocks))
        L = lib()
        L.zs_rocm_mpm_compute_sparsity(self.pol.handle, self.table.handle, self._port("x"), self.n, self.params.dx, self.side,
                                       int(self.key_is_origin))
        n_holders = 0
        if order == "holders_lex":  # (the apron blocks are inserted after this and keep the indices behind the holders)
            self.pol.syncCtx()
            self.table.canonicalize(self.pol, axes)
            self.pol.syncCtx()
            n_holders = self.table.size()
        if os.environ.get("ZS_ROCM_HOLDER_ORDER"):   # measurement only (tools/r06_order2.sh)
            self._measurement_holder_order(os.environ["ZS_ROCM_HOLDER_ORDER"])
        m = int(margin)
        self.partition_margin = m   # (repartition_slotted() keeps the same travel room unless told otherwise)
        lo, hi = (C.c_int * 3)(-m, -m, -m), (C.c_int * 3)(2 + m, 2 + m, 2 + m)
        L.zs_rocm_mpm_enlarge_sparsity(self.pol.handle, self.table.handle, lo, hi, self.kstride)
        self.pol.syncCtx()
        if order == "holders_lex" and not os.environ.get("ZS_ROCM_APRON_RACE"):   # the apron blocks in key order too: the whole numbering is reproducible
            self.table.canonicalize(self.pol, axes, first=n_holders)                    # (the variable: A/B of profiles/r06_p2g.md, section 4b)
            self.pol.syncCtx()
        if order == "morton":
            self.table.order_morton(self.pol)
            self.pol.syncCtx()
        elif order == "lex":
            self.table.canonicalize(self.pol, axes)
            self.pol.syncCtx()
        self.block_order, self.block_axes = order, (tuple(axes) if axes is not None else None)
        self.nblocks = self.table.size()
        self.slotted = False
        nc = self.side ** 3
        self.grid = torch.zeros(self.nblocks * 7 * nc, dtype=torch.float32, device=self.device)
        self.nbr = torch.empty(self.nblocks * 8, dtype=torch.int32, device=self.device)
        L.zs_rocm_mpm_build_neighbors(self.pol.handle, self.table.handle, self.nbr.data_ptr(), self.kstride)
        self.binned = False
        return self.nblocks

    def _measurement_holder_order(self, xp):
        """MEASUREMENT ONLY (profiles/r06_block_order_*.txt): renumber the blocks that hold particles (the table right after ComputeSparsity).
        "120" = key axis 1 most significant, then 2, then 0; "120:2,4,4" = the same inside and across tiles of 2 x 4 x 4 blocks; "m" = Z-order
        curve; suffix "s17": position p holds the (17 p mod n)-th holder of that order; suffix "x": XCD k (workgroup number mod 8) walks the
        k-th contiguous eighth."""
        import ctypes
        self.pol.syncCtx()
        n0 = self.table.size()
        v = self.table.view()
        k = torch.empty(n0 * 3, dtype=torch.int32, device=self.device)
        ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(v.activeKeys), ctypes.c_size_t(n0 * 12), 3)
        k = k.view(n0, 3).to(torch.int64) // self.kstride
        k = k - k.min(dim=0).values
        if xp == "m":
            code = torch.zeros(n0, dtype=torch.int64, device=self.device)
            for bit in range(16):
                for d in range(3):
                    code |= ((k[:, d] >> bit) & 1) << (3 * bit + (2 - d))
        else:
            ax = [int(c) for c in xp[:3]]
            t = [int(c) for c in xp.rstrip("x").split("s")[0].split(":")[1].split(",")] if ":" in xp else [1, 1, 1]
            code = torch.zeros(n0, dtype=torch.int64, device=self.device)
            for d in ax:
                code = code * 4096 + k[:, d] // t[d]
            for d in ax:
                code = code * t[d] + k[:, d] % t[d]
        perm = torch.argsort(code).to(torch.int32).contiguous()
        if "s" in xp:   # "021s17": position p holds the (17 p mod n)-th holder of the sorted order (n made coprime by dropping a tail)
            st = int(xp.split("s")[1])
            import math
            nn = n0
            while math.gcd(nn, st) != 1:
                nn -= 1
            pos = (torch.arange(nn, device=self.device, dtype=torch.int64) * st) % nn
            perm = torch.cat([perm[pos], perm[nn:]]).contiguous()
        if xp.endswith("x"):   # XCD k (workgroup number mod 8) walks the k-th contiguous eighth of the sorted holders
            q = n0 // 8
            pos = torch.arange(8 * q, device=self.device)
            perm = torch.cat([perm[(pos % 8) * q + pos // 8], perm[8 * q:]]).contiguous()
        torch.cuda.synchronize()
        self.table.reorder(self.pol, perm.data_ptr(), scatter=False)
        self.pol.syncCtx()

    def adopt_partition(self, active_keys_ptr, nblocks):
        """Use a partition numbered elsewhere -- the `_activeKeys` of a zs::HashTable<i32,3,int> built by
        partition_for_particles (the reference's Grids path) -- block i of the grid = active_keys[i]."""
        L = lib()
        self.table = Bht(3, int(nblocks))
        self.table.assign(self.pol, active_keys_ptr, int(nblocks))
        self.nblocks = int(nblocks)
        nc = self.side ** 3
        self.grid = torch.zeros(self.nblocks * 7 * nc, dtype=torch.float32, device=self.device)
        self.nbr = torch.empty(self.nblocks * 8, dtype=torch.int32, device=self.device)
        L.zs_rocm_mpm_build_neighbors(self.pol.handle, self.table.handle, self.nbr.data_ptr(), self.kstride)
        self.binned = False
        return self.nblocks

    def rebin(self, inputs_only=False):
        """particle -> block binning + physical reorder of the AoSoA buffer (count / scan / distribute).
        inputs_only: carry only m, x, F (or J) and logJp -- everything a fused G2P2G step reads; v, C and the cached stress are
        outputs of that step (recomputed from the grid), so between fused steps they need not be moved (14 instead of 32
        channels for the sand column).  Do not use it before p2g() / g2p() or before reading v, C, stress."""
        L = lib()
        if self.order is None or self.order.numel() != self.n:
            self.order = torch.empty(self.n, dtype=torch.int32, device=self.device)
        self.nbins = self.nblocks * (self.side // 4) ** 3  # a bin = 4x4x4 cells = one wavefront
        self.bin_start = torch.empty(self.nbins + 1, dtype=torch.int32, device=self.device)
        self.cell_count = torch.empty(self.nbins * 64, dtype=torch.int32, device=self.device)
        L.zs_rocm_mpm_bin_particles(self.pol.handle, self.table.handle, self._port("x"), self.n, self.params.dx, self.side,
                                    int(self.key_is_origin), self.order.data_ptr(), self.bin_start.data_ptr(), self.cell_count.data_ptr())
        if self.buf2 is None:
            self.buf2 = torch.empty_like(self.buf)
        if inputs_only:
            mask = 0xF | (((1 << self.nF) - 1) << self.off["F"]) | ((1 << self.off["logJp"]) if self.model in HAS_LOGJP else 0)
            L.zs_rocm_tv_gather_channels_f32(self.pol.handle, self.buf.data_ptr(), self.buf2.data_ptr(), self.n, self.nchn, self.L,
                                             self.order.data_ptr(), mask)
        else:
            L.zs_rocm_tv_gather_f32(self.pol.handle, self.buf.data_ptr(), self.buf2.data_ptr(), self.n, self.nchn, self.L,
                                    self.order.data_ptr())
        self.pol.syncCtx()
        self.buf, self.buf2 = self.buf2, self.buf
        self.binned = True
        if self.drift_flag is not None:
            flags = self.drift_flag.cpu()
            self.drift_tripped = self.drift_tripped or bool(flags[0])  # latched: a re-bin must not erase them
            self.outside_tripped = self.outside_tripped or bool(flags[2])
            self._zero(self.drift_flag)

    # ------------------------------------------------------------------ slotted storage (zpc_amd/csrc/mpm_slotted.hip)
    SLOT_FLAG_NAMES = ["outbox full", "a cell is full (K)", "mass or a mover for a block outside the partition",
                       "early warning: a particle lives next to the partition's edge", "a particle was not stored under its cell"]

    def slot(self, K=24, outbox_cap=128):
        """compact particle buffer -> slotted storage (bins x K rounds x 64 lanes, one tile row per (bin, round)): the form the
        fused step keeps valid by itself while particles move (no re-bins).  Needs the partition; lane width 64."""
        assert self.L == 64 and not self.aos and self.table is not None and not self.slotted
        L = lib()
        self.K = int(K)
        self.outbox_cap = int(outbox_cap)   # movers one bin can send per step (a bin holds 512 particles at 8 per cell)
        self.nbins = self.nblocks * (self.side // 4) ** 3
        self.n_slots = self.nbins * self.K * 64
        sbuf = torch.empty(self.nbins * self.K * 64 * self.nchn, dtype=torch.float32, device=self.device)
        self.cell_mask = torch.empty(self.nbins * 64, dtype=torch.int32, device=self.device)
        self.slot_status = torch.zeros(8 + 2 * 256, dtype=torch.int32, device=self.device)  # ZS_ROCM_SLOT_STATUS_WORDS
        self.nbr27 = torch.empty(self.nblocks * 27, dtype=torch.int32, device=self.device)
        L.zs_rocm_mpm_build_neighbors27(self.pol.handle, self.table.handle, self.nbr27.data_ptr(), self.kstride)
        # blocks from which a particle could reach the partition's edge in one more block of travel (status word [3]: re-partition soon)
        self.block_edge = torch.empty(max(self.nblocks, 1), dtype=torch.uint8, device=self.device)
        L.zs_rocm_mpm_partition_edge(self.pol.handle, self.table.handle, self.block_edge.data_ptr(), self.kstride, -1, 2)
        rc = L.zs_rocm_mpm_slot_particles(self.pol.handle, self.table.handle, self._port("x"), self.n, self.params.dx, self.side,
                                          int(self.key_is_origin), self.K, self.buf.data_ptr(), sbuf.data_ptr(), self.nchn,
                                          self.cell_mask.data_ptr(), self.slot_status.data_ptr())
        if rc != 0:
            raise RuntimeError("zs_rocm_mpm_slot_particles refused its arguments")
        self.mover_count = torch.zeros(L.zs_rocm_mpm_slot_outbox_bytes(self.nbins, self.outbox_cap, 0) // 4, dtype=torch.int32, device=self.device)
        self.mover_dest = torch.zeros(L.zs_rocm_mpm_slot_outbox_bytes(self.nbins, self.outbox_cap, 1) // 4, dtype=torch.int32, device=self.device)  # claim words: zero between steps
        self.mover_rec = torch.empty(L.zs_rocm_mpm_slot_outbox_bytes(self.nbins, self.outbox_cap, 2) // 4, dtype=torch.float32, device=self.device)
        self.pol.syncCtx()
        st = self.slot_status.cpu().numpy()
        if st[1] or st[2]:
            raise RuntimeError("slot(): %s" % ("a cell holds more than K = %d particles" % self.K if st[1] else "particles outside the partition"))
        self.buf, self.buf2 = sbuf, None
        self.slotted, self.binned = True, False
        self.order = self.bin_start = self.cell_count = None
        self._edge_host = self._edge_event = None
        self.slot_storage = SlotStorage(self.cell_mask.data_ptr(), self.K, self.nbr.data_ptr(), self.nbr27.data_ptr(), self.mover_count.data_ptr(),
                                        self.mover_dest.data_ptr(), self.mover_rec.data_ptr(), self.outbox_cap, self.slot_status.data_ptr(),
                                        self.block_edge.data_ptr())

    def _backing(self, name, numel, dtype, zero=False, spare=False):
        """a view of `numel` elements of a grow-only backing tensor kept under `name`: a re-partition every ~100 steps must not pay a
        hipMalloc of tens of GB each time (0.5-1 s for the 43 GB slot buffer of the 64 Mi-particle column)"""
        pool = self.__dict__.setdefault("_pool", {})
        t = pool.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(int(numel * (1.25 if spare else 1.0)) + 64, dtype=dtype, device=self.device)
            pool[name] = t
        v = t[:numel]
        if zero:
            self._zero(v)
        return v

    def reserve_repartition_buffers(self, growth=1.25):
        """allocate, outside any timed region, what repartition_slotted() will need: the second slot buffer (the two are used in turns)
        and the per-partition arrays, sized for a partition `growth` times today's"""
        assert self.slotted
        n = int(self.buf.numel() * growth)
        pool = self.__dict__.setdefault("_pool", {})
        if "slots_alt" not in pool or pool["slots_alt"].numel() < n:
            pool["slots_alt"] = torch.empty(n, dtype=torch.float32, device=self.device)
        self.pol.syncCtx()

    def repartition_slotted(self, margin=None, strict=True):
        """(margin: None = the travel room build_partition() was given.  After this call only the slots the occupancy words select are defined in
        self.buf -- the spare buffer is not cleared: include/zs_rocm.h, zs_rocm_mpm_reslot.)
        Re-partition IN PLACE (zs_rocm_mpm_slot_compute_sparsity + enlarge + zs_rocm_mpm_reslot): the new partition is the reference's
        ComputeSparsity + EnlargeSparsity over the cells that hold particles (taken from the occupancy words: no particle is read), every
        populated bin moves as whole tile rows to its block's new number, the grid of node velocities (self.grid: what the next fused step
        gathers from) is carried over.  Single rank only (particles do not change owner).  Returns the new number of blocks."""
        assert self.slotted and self.L == 64
        L = lib()
        self.check_slots(strict=strict)                      # fold the period's status words into slot_record first
        old_table, old_nblocks = self.table, self.nblocks
        m = int(getattr(self, "partition_margin", 0) if margin is None else margin)   # default: the travel room build_partition() gave
        # capacity: the enlarged partition has at most (3 + 2 m)^3 blocks per populated block (a bound, never reached by a bulk of particles);
        # a cloud that keeps its shape needs about what it had.  Twice the old count + slack, and the table says if that was not enough.
        new_table = Bht(3, max(2 * int(old_nblocks) + 64, 4096))
        L.zs_rocm_mpm_slot_compute_sparsity(self.pol.handle, old_table.handle, self.cell_mask.data_ptr(), old_nblocks, self.side,
                                            int(self.key_is_origin), new_table.handle)
        if getattr(self, "block_order", "insertion") == "holders_lex":   # the numbering build_partition() was asked for
            self.pol.syncCtx()
            new_table.canonicalize(self.pol, getattr(self, "block_axes", None))
            self.pol.syncCtx()
            n_holders = new_table.size()
        lo, hi = (C.c_int * 3)(-m, -m, -m), (C.c_int * 3)(2 + m, 2 + m, 2 + m)
        L.zs_rocm_mpm_enlarge_sparsity(self.pol.handle, new_table.handle, lo, hi, self.kstride)
        self.pol.syncCtx()
        if getattr(self, "block_order", "insertion") == "holders_lex":
            new_table.canonicalize(self.pol, getattr(self, "block_axes", None), first=n_holders)
            self.pol.syncCtx()
        nb = new_table.size()
        if not new_table.success() or nb == 0:   # nothing of this object has been replaced yet: the caller can still unslot() and re-partition the long way
            raise RuntimeError("repartition_slotted: the new partition did not fit its table (%d blocks, capacity for %d)"
                               % (nb, max(2 * int(old_nblocks) + 64, 4096)))
        bpb = (self.side // 4) ** 3
        nbins = nb * bpb
        nc = self.side ** 3
        # the two slot buffers are used in turns (the one being left becomes the spare)
        pool = self.__dict__.setdefault("_pool", {})
        need = nbins * self.K * 64 * self.nchn
        alt = pool.get("slots_alt")
        if alt is None or alt.numel() < need:
            alt = torch.empty(int(need * 1.1), dtype=torch.float32, device=self.device)
        sbuf = alt[:need]
        # the buffer being left is whatever self.buf is a view of NOW (after an unslot() / slot() cycle "slots_cur" would name a buffer that is
        # no longer in use, and the live one would be dropped instead of parked)
        old_store = self.buf._base if self.buf._base is not None else self.buf
        flip = self.__dict__.get("_flip", 0) ^ 1
        self._flip = flip
        mask = self._backing("mask%d" % flip, nbins * 64, torch.int32)
        grid = self._backing("grid%d" % flip, nb * 7 * nc, torch.float32)
        if L.zs_rocm_mpm_reslot(self.pol.handle, old_table.handle, new_table.handle, self.side, self.K, self.nchn, self.buf.data_ptr(),
                                sbuf.data_ptr(), self.cell_mask.data_ptr(), mask.data_ptr(), self.grid.data_ptr(), grid.data_ptr(),
                                self.slot_status.data_ptr()) != 0:
            raise RuntimeError("zs_rocm_mpm_reslot refused its arguments")
        self.pol.syncCtx()                                   # the old buffers are read until here
        pool["slots_cur"], pool["slots_alt"] = alt, old_store
        self.table, self.nblocks, self.nbins, self.n_slots = new_table, nb, nbins, nbins * self.K * 64
        self.buf, self.cell_mask, self.grid = sbuf, mask, grid
        self.grid2 = self._backing("gridB%d" % flip, nb * 7 * nc, torch.float32, zero=True)
        self.nbr = self._backing("nbr", nb * 8, torch.int32, spare=True)
        L.zs_rocm_mpm_build_neighbors(self.pol.handle, self.table.handle, self.nbr.data_ptr(), self.kstride)
        self.nbr27 = self._backing("nbr27", nb * 27, torch.int32, spare=True)
        L.zs_rocm_mpm_build_neighbors27(self.pol.handle, self.table.handle, self.nbr27.data_ptr(), self.kstride)
        self.block_edge = self._backing("edge", max(nb, 1), torch.uint8, spare=True)
        L.zs_rocm_mpm_partition_edge(self.pol.handle, self.table.handle, self.block_edge.data_ptr(), self.kstride, -1, 2)
        self.mover_count = self._backing("mcount", L.zs_rocm_mpm_slot_outbox_bytes(nbins, self.outbox_cap, 0) // 4, torch.int32, zero=True, spare=True)
        self.mover_dest = self._backing("mdest", L.zs_rocm_mpm_slot_outbox_bytes(nbins, self.outbox_cap, 1) // 4, torch.int32, zero=True, spare=True)  # claim words: zero between steps
        self.mover_rec = self._backing("mrec", L.zs_rocm_mpm_slot_outbox_bytes(nbins, self.outbox_cap, 2) // 4, torch.float32, spare=True)
        self._edge_host = self._edge_event = None
        self.slot_storage = SlotStorage(self.cell_mask.data_ptr(), self.K, self.nbr.data_ptr(), self.nbr27.data_ptr(), self.mover_count.data_ptr(),
                                        self.mover_dest.data_ptr(), self.mover_rec.data_ptr(), self.outbox_cap, self.slot_status.data_ptr(),
                                        self.block_edge.data_ptr())
        self.pol.syncCtx()
        if int(self.slot_status[2].item()):
            raise RuntimeError("repartition_slotted: a populated block is missing from the new partition")
        return nb

    def _compact_copy(self):
        """(compact TileVector buffer of the occupied slots in slot order, particle count)"""
        L = lib()
        slots = torch.empty(self.n_slots, dtype=torch.int32, device=self.device)
        cnt = int(L.zs_rocm_mpm_slot_list(self.pol.handle, self.cell_mask.data_ptr(), self.nbins, self.K, slots.data_ptr()))
        tiles = (cnt + 63) // 64
        out = torch.zeros(max(tiles, 1) * 64 * self.nchn, dtype=torch.float32, device=self.device)
        if cnt:
            L.zs_rocm_tv_gather_f32(self.pol.handle, self.buf.data_ptr(), out.data_ptr(), cnt, self.nchn, 64, slots.data_ptr())
        self.pol.syncCtx()
        return out, cnt

    def unslot(self, strict=True):
        """slotted storage -> compact buffer (before a re-partition / migration).  The period's status words are folded into
        slot_record first (strict: a raised flag other than the early warning, or sent != re-homed, raises); the particle count must
        come out as it went in -- the slotted step never drops a particle."""
        assert self.slotted
        self.check_slots(strict=strict)
        out, cnt = self._compact_copy()
        if cnt != self.n:
            raise RuntimeError("slotted storage holds %d particles, %d went in: particles were lost" % (cnt, self.n))
        self.slotted = False
        self.tiles = (cnt + 63) // 64
        self.buf, self.buf2 = out, None
        self.binned = False

    def check_slots(self, strict=True):
        """Fold the status words of the slotted step into slot_record (and clear them on the device); strict: raise if a capacity
        overflowed, a storage invariant broke or a mover was not re-homed.  Returns the period's first 8 words ([5], [6]: movers
        sent / re-homed).  Word [3] (early warning: re-partition soon) never raises -- see repartition_requested()."""
        # the steps in flight (on the policy's stream, which need not be torch's current one) must have finished before the words are
        # read AND before they are zeroed: a flag or a mover count written in between would be erased without reaching slot_record
        self.pol.syncCtx()
        st = [int(v) for v in self.slot_status.cpu().numpy()]
        st[5], st[6] = sum(st[8:264]), sum(st[264:520])  # the counters are spread over 256 words each
        self._zero(self.slot_status)
        rec = self.slot_record
        rec["sent"] += st[5]
        rec["homed"] += st[6]
        rec["periods"] += 1
        rec["edge_periods"] += int(bool(st[3]))
        for k in range(5):
            rec["flags"][k] += int(bool(st[k]))
        bad = [self.SLOT_FLAG_NAMES[k] for k in (0, 1, 2, 4) if st[k]]
        if st[5] != st[6]:
            bad.append("%d movers sent, %d re-homed" % (st[5], st[6]))
        if bad:
            rec["log"].append((rec["periods"], bad))
            if strict:
                raise RuntimeError("slotted G2P2G: " + "; ".join(bad))
        return st[:8]

    def poll_repartition(self, reduce=None):
        """Closed-loop re-partition trigger without a stall on the stream: returns the answer of the PREVIOUS call's copy of the status
        words -- True if [3] (a particle lives in a block next to the partition's edge) or an overflow flag ([0], [1], [2], [4]) was
        set -- and starts a new asynchronous device -> pinned-host copy behind the steps enqueued so far.  The answer lags one polling
        interval and is the same on every call sequence (it waits for the previous copy's event, which has normally long fired), so
        N ranks that poll at the same steps decide alike; `reduce(t)`: all-reduce(max) the float tensor t in place on the policy's
        stream (multi-rank).  The warning is raised a whole block (side cells of travel) before a particle can run out of partition:
        poll at least every side / 4 / (cells per step) steps."""
        ready = False
        if self._edge_event is not None:
            self._edge_event.synchronize()
            ready = bool(self._edge_host[:5].any())
        if self._edge_host is None:
            self._edge_host = torch.zeros(8, dtype=torch.float32).pin_memory()
            self._edge_event = torch.cuda.Event()
        sp = self.pol.getStream()  # the copy must be ordered behind the steps: the policy's stream
        with torch.cuda.stream(torch.cuda.ExternalStream(sp) if sp else torch.cuda.default_stream(self.device)):
            w = self.slot_status[:8].float()
            if reduce is not None:
                reduce(w)
            self._edge_host.copy_(w, non_blocking=True)
            self._edge_event.record()
        return ready

    def repartition_requested(self):
        """synchronous form of poll_repartition(): reads the status words now"""
        w = self.slot_status[:5].cpu()
        return bool(w.any())

    # ------------------------------------------------------------------ one sub-step
    def clear_grid(self):
        self._zero(self.grid)  # TileVector::reset(pol, 0) (TileVector.hpp:636-640)

    def p2g(self, binned=None):
        binned = self.binned if binned is None else binned
        bs = self.bin_start.data_ptr() if binned else None
        cc = self.cell_count.data_ptr() if binned else None
        nb = self.nbr.data_ptr() if binned else None
        lib().zs_rocm_mpm_p2g(self.pol.handle, C.byref(self.params), self.particles(), self.table.handle, self.grid.data_ptr(),
                              self.nblocks, bs, cc, nb)

    def grid_update(self, extf=(0.0, 0.0, 0.0), max_vel=None):
        e = (C.c_float * 3)(*extf)
        lib().zs_rocm_mpm_grid_update(self.pol.handle, C.byref(self.params), self.grid.data_ptr(), self.nblocks, e,
                                      max_vel.data_ptr() if max_vel is not None else None)

    def apply_boundary(self, collider):
        """ApplyBoundaryConditionOnGridBlocks (simulation/grid/GridOp.hpp:111-164): collider.resolveCollision on every grid node
        with mass; call after grid_update.  `collider`: make_collider(...)."""
        lib().zs_rocm_mpm_apply_boundary(self.pol.handle, C.byref(self.params), self.table.handle, self.grid.data_ptr(), self.nblocks,
                                         C.byref(collider))

    def g2p(self, binned=None):
        binned = self.binned if binned is None else binned
        bs = self.bin_start.data_ptr() if binned else None
        cc = self.cell_count.data_ptr() if binned else None
        nb = self.nbr.data_ptr() if binned else None
        lib().zs_rocm_mpm_g2p(self.pol.handle, C.byref(self.params), self.particles(), self.table.handle, self.grid.data_ptr(),
                              self.nblocks, bs, cc, nb)

    # ------------------------------------------------------------------ gather-style transfers (P2C2G.hpp / G2C2P.hpp)
    def build_buckets(self, displacement=0.0, dense=False):
        """IndexBuckets of cell size dx over the current positions (index_buckets_for_particles, displacement 0): bucket = the cell
        that contains the particle, which is what P2C2GTransfer's 27-bucket walk expects."""
        from .containers import IndexBuckets
        if getattr(self, "buckets", None) is None:
            self.buckets = IndexBuckets()   # rebuilt in place every step: the table and the arrays are reused
        if dense:   # buckets over the partition's own cells: no hash table to fill (a time loop rebuilds the buckets every step)
            assert displacement == 0.0
            self.buckets.build_for_partition(self.pol, self._port("x"), self.n, self.params.dx, self.table.handle, self.side,
                                             self.key_is_origin)
            return self.buckets
        # table sized for the occupied cells, which the partition bounds (the reference sizes it for one cell per particle)
        cells = min(self.n, self.nblocks * self.side ** 3) if getattr(self, "nblocks", 0) else 0
        self.buckets.build(self.pol, self._port("x"), self.n, self.params.dx, displacement=displacement, expected_cells=cells)
        return self.buckets

    def p2c2g(self, kind=0):
        """kind 0 P2C2GTransfer, 1 P2C2GTransferMomentum, 2 P2C2GTransferForce; ADDS into grid channels m, mv."""
        if lib().zs_rocm_mpm_p2c2g(self.pol.handle, C.byref(self.params), self.particles(), self.buckets._h, self.table.handle,
                                   self.grid.data_ptr(), self.nblocks, int(kind)) != 0:
            raise RuntimeError("zs_rocm_mpm_p2c2g refused its arguments")

    def g2c2p(self, fused=True):
        """PreG2C2PTransfer + G2C2PTransfer + PostG2C2PTransfer: v, B (= C) from the grid, then F (or J) and x advance.
        fused: one pass over the particles (zs_rocm_mpm_g2c2p_step) instead of the three calls; same bits."""
        L = lib()
        if fused:
            if L.zs_rocm_mpm_g2c2p_step(self.pol.handle, C.byref(self.params), self.particles(), self.table.handle, self.grid.data_ptr(),
                                        self.nblocks) != 0:
                raise RuntimeError("zs_rocm_mpm_g2c2p_step refused its arguments")
            self.binned = False
            return
        L.zs_rocm_mpm_pre_g2c2p(self.pol.handle, self.particles())
        if L.zs_rocm_mpm_g2c2p(self.pol.handle, C.byref(self.params), self.particles(), None, self.table.handle, self.grid.data_ptr(),
                               self.nblocks) != 0:
            raise RuntimeError("zs_rocm_mpm_g2c2p refused its arguments")
        L.zs_rocm_mpm_post_g2c2p(self.pol.handle, C.byref(self.params), self.particles())
        self.binned = False

    def g2p2g(self, write_all=False, split=None, between=None, reorder=False):
        """Fused G2P (from self.grid) + P2G (into a zeroed second grid, which then becomes self.grid): zs_rocm_mpm_g2p2g.
        Needs cache_stress=True and binned particles.  split=k: blocks [0, k) are launched first, `between()` is called (the
        caller records an event there and starts the ghost exchange of the NEW self.grid on another stream), then blocks
        [k, nblocks) are launched (zs_rocm_mpm_g2p2g_range).
        reorder=True: the particles are binned anew by their current positions (count / scan / distribute only) and the step
        itself carries them into that order -- inputs read from the old buffer through the permutation, results stored to the
        other buffer (zs_rocm_mpm_g2p2g_reorder_range): a re-bin without the separate reorder pass."""
        if self.slotted:
            if reorder:
                raise ValueError("slotted storage: no re-ordering steps (the step keeps the order itself)")
            if getattr(self, "grid2", None) is None or self.grid2.numel() != self.grid.numel():
                self.grid2 = torch.zeros_like(self.grid)
            else:
                self._zero(self.grid2)
            src, dst = self.grid, self.grid2
            self.grid, self.grid2 = dst, src  # `between` sees the grid being accumulated as self.grid
            ranges = [(0, self.nblocks)] if not split else [(0, int(split)), (int(split), self.nblocks)]
            for k, (b0, b1) in enumerate(ranges):
                rc = lib().zs_rocm_mpm_g2p2g_slots(self.pol.handle, C.byref(self.params), self.particles(), self.table.handle, src.data_ptr(),
                                                   dst.data_ptr(), self.nblocks, C.byref(self.slot_storage), int(write_all),
                                                   b0, b1, int(k == len(ranges) - 1))
                if rc != 0:
                    raise RuntimeError("zs_rocm_mpm_g2p2g_slotted refused the call")
                if k == 0 and between is not None:
                    between()
            return
        if not (self.cache_stress and self.binned):
            raise RuntimeError("g2p2g needs cache_stress=True and rebin()")
        src_buf = None
        if reorder and write_all:
            raise ValueError("a re-ordering step cannot also materialise v, C, stress (write_all)")
        if reorder:
            L = lib()
            if self.order is None or self.order.numel() != self.n:
                self.order = torch.empty(self.n, dtype=torch.int32, device=self.device)
            L.zs_rocm_mpm_bin_particles(self.pol.handle, self.table.handle, self._port("x"), self.n, self.params.dx, self.side,
                                        int(self.key_is_origin), self.order.data_ptr(), self.bin_start.data_ptr(),
                                        self.cell_count.data_ptr())
            if self.buf2 is None:
                self.buf2 = torch.empty_like(self.buf)
            src_buf, self.buf, self.buf2 = self.buf, self.buf2, self.buf  # results go to the other buffer
        if getattr(self, "grid2", None) is None or self.grid2.numel() != self.grid.numel():
            self.grid2 = torch.zeros_like(self.grid)
        else:
            self._zero(self.grid2)
        if self.drift_flag is None:
            self.drift_flag = torch.zeros(3, dtype=torch.int32, device=self.device)
        ranges = [(0, self.nblocks)] if not split else [(0, int(split)), (int(split), self.nblocks)]
        src, dst = self.grid, self.grid2
        self.grid, self.grid2 = dst, src  # `between` sees the grid being accumulated as self.grid
        for k, (b0, b1) in enumerate(ranges):
            if src_buf is None:
                rc = lib().zs_rocm_mpm_g2p2g_range(self.pol.handle, C.byref(self.params), self.particles(), self.table.handle, src.data_ptr(),
                                                   dst.data_ptr(), self.nblocks, self.bin_start.data_ptr(), self.cell_count.data_ptr(),
                                                   self.nbr.data_ptr(), int(write_all), b0, b1, self.drift_flag.data_ptr())
            else:
                rc = lib().zs_rocm_mpm_g2p2g_reorder_range(self.pol.handle, C.byref(self.params), self.particles(), self.particles(src_buf),
                                                           self.order.data_ptr(), self.table.handle, src.data_ptr(), dst.data_ptr(),
                                                           self.nblocks, self.bin_start.data_ptr(), self.cell_count.data_ptr(),
                                                           self.nbr.data_ptr(), int(write_all), b0, b1, self.drift_flag.data_ptr())
            if rc != 0:
                raise RuntimeError("zs_rocm_mpm_g2p2g refused the call")
            if k == 0 and between is not None:
                between()

    def step_slotted(self, extf=(0.0, 0.0, 0.0), max_vel=None, write_all=False, n_boundary=0, comm=None, plan=None, comm_pol=None,
                     collider=None, halo_grid=None, events=None, breakdown=None, halo_channels=7, range_schedule=0, handover_snapshot=None):
        """One whole sub-step on slotted storage behind ONE C-ABI call (zs_rocm_mpm_step_slotted): second grid := 0, fused G2P2G over the
        boundary blocks [0, n_boundary) then the interior, ghost-block exchange of `plan` on comm_pol's stream overlapping the interior,
        grid update (+ collider), CFL allreduce(max) of max_vel.  The grids swap: self.grid is the new one afterwards.
        comm: NativeComm, plan: NativeHaloPlan (both None on a single rank)."""
        assert self.slotted
        if getattr(self, "grid2", None) is None or self.grid2.numel() != self.grid.numel():
            self.grid2 = torch.empty_like(self.grid)
        src, dst = self.grid, self.grid2
        a = getattr(self, "_step_args", None)
        if a is None:
            a = self._step_args = MpmStep()
        a.params = C.addressof(self.params)
        a.particles = self.particles()
        a.table = self.table.handle
        a.gridA, a.gridB = src.data_ptr(), dst.data_ptr()
        a.nblocks = self.nblocks
        a.storage = C.addressof(self.slot_storage)
        a.writeAll = int(write_all)
        a.extf = (C.c_float * 3)(*extf)
        a.maxVelSqr = max_vel.data_ptr() if max_vel is not None else None
        a.collider = C.addressof(collider) if collider is not None else None
        a.nBoundary = int(n_boundary)
        a.dist = comm._h if comm is not None else None
        a.plan = plan._h if plan is not None else None
        a.commPolicy = comm_pol.handle if comm_pol is not None else None
        a.haloGrid = halo_grid.data_ptr() if halo_grid is not None else None
        a.haloChannels = int(halo_channels)   # 4: only {m, mv} of the ghost blocks travel (all a step reads of them); 7: the rhs channels too
        a.rangeSchedule = int(range_schedule)   # RANGES_IN_TURN / _SIDE_BY_SIDE / _ONE_LAUNCH (include/zs_rocm.h, zs_rocm_mpm_step.rangeSchedule)
        a.handoverSnapshot = handover_snapshot.data_ptr() if handover_snapshot is not None else None   # (test hook)
        a.evTransferBegin, a.evTransferEnd = (events[0], events[1]) if events is not None else (None, None)   # raw hipEvent_t (HipEvents)
        a.evBreakdown = breakdown if breakdown is not None else None   # (C.c_void_p * ZS_ROCM_STEP_EVENTS) of raw hipEvent_t (StepBreakdown.next())
        if getattr(self, "_poisoned", False):
            raise RuntimeError("this MpmTransfer is poisoned: an earlier zs_rocm_mpm_step_slotted failed after committing the particle side of its step")
        if lib().zs_rocm_mpm_step_slotted(self.pol.handle, C.byref(a)) != 0:
            # (include/zs_rocm.h: the storage has been committed, the grids have not -- the particles are one step ahead; neither a retry nor a
            # continuation is valid)
            self._poisoned = True
            raise RuntimeError("zs_rocm_mpm_step_slotted failed: the step is half applied (particles advanced, grids not): this object cannot be stepped again")
        self.grid, self.grid2 = dst, src

    def margin_violated(self):
        """True if, since construction, an exact-path particle was ever farther than one bin from its bin during a fused step
        (the overlapped multi-GPU exchange is then not valid: see zs_rocm_mpm_g2p2g_range)."""
        return self.drift_tripped or (self.drift_flag is not None and bool(self.drift_flag[0].item()))

    def left_partition(self):
        """True if a fused step ever dropped a contribution because a particle's stencil reached a block that is not in the
        partition: rebuild the partition (build_partition / re-map) more often."""
        return self.outside_tripped or (self.drift_flag is not None and bool(self.drift_flag[2].item()))

    def exact_path_particles(self, reset=True):
        """Particles the fused steps since the last call handled on the exact path (they left their cell after the last
        re-bin); synchronises the stream.  The re-bin trigger: the exact path costs ~50x the binned one per particle."""
        if self.drift_flag is None:
            return 0
        c = int(self.drift_flag[1].item())
        if reset:
            self._zero(self.drift_flag[1:2])  # the flags [0], [2] stay
        return c

    def reorder_partition(self, first_mask):
        """Renumber the blocks so that those with first_mask[i] != 0 come first (stable); returns their count.  Call between
        build_partition() and rebin()."""
        import numpy as np
        keys = self.active_keys()
        m = np.asarray(first_mask).astype(bool)
        order = np.concatenate([np.nonzero(m)[0], np.nonzero(~m)[0]])
        pk = torch.from_numpy(np.ascontiguousarray(keys[order])).to(self.device)
        self.adopt_partition(pk.data_ptr(), keys.shape[0])
        self.pol.syncCtx()
        return int(m.sum())

    def active_keys(self):
        """[nblocks, 3] int32 block keys in block-number order (host copy of the table's activeKeys)."""
        import ctypes
        v = self.table.view()
        nb = self.nblocks
        keys = torch.empty(max(nb, 1) * 3, dtype=torch.int32, device=self.device)
        self.pol.syncCtx()
        if nb:
            ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(v.activeKeys), ctypes.c_size_t(nb * 12), 3)
        return keys.cpu().numpy()[: nb * 3].reshape(nb, 3)

    def grid_by_key(self):
        """{(bx,by,bz): ndarray[7, side^3]} -- for comparisons that must not depend on block numbering."""
        v = self.table.view()
        nb = self.nblocks
        keys = torch.empty(nb * 3, dtype=torch.int32, device=self.device)
        import ctypes
        # activeKeys is a device pointer owned by the table: copy through hipMemcpy via torch's from-pointer-free path
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy(ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(v.activeKeys), ctypes.c_size_t(nb * 12), 3)
        k = keys.cpu().numpy().reshape(nb, 3)
        g = self.grid.cpu().numpy().reshape(nb, 7, self.side ** 3)
        return {tuple(int(x) for x in k[i]): g[i] for i in range(nb)}


class HipEvents:
    """raw hipEvent_t pairs for timing stretches the library enqueues itself (zs_rocm_mpm_step.evTransferBegin / End)"""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.pairs = []

    def pair(self):
        e = [C.c_void_p(), C.c_void_p()]
        for x in e:
            if self.hip.hipEventCreate(C.byref(x)) != 0:
                raise RuntimeError("hipEventCreate failed")
        self.pairs.append(e)
        return e[0], e[1]

    def elapsed_ms(self):
        """[ms] of every pair (the stream must have been synchronised)"""
        out = []
        for a, b in self.pairs:
            ms = C.c_float(0)
            if self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0:
                out.append(ms.value)
        return out

    def __del__(self):
        try:
            for a, b in self.pairs:
                self.hip.hipEventDestroy(a)
                self.hip.hipEventDestroy(b)
        except Exception:
            pass


RANGES_IN_TURN, RANGES_SIDE_BY_SIDE, RANGES_ONE_LAUNCH = 0, 1, 2   # ZS_ROCM_RANGES_*: zs_rocm_mpm_step.rangeSchedule


class StepBreakdown:
    """zs_rocm_mpm_step.evBreakdown: ZS_ROCM_STEP_EVENTS raw hipEvent_t per recorded step; after a synchronisation, the mean time of every
    stretch of a rank's step (ms): where a multi-GPU step's time goes (boundary range, interior range, the exchange on the side stream,
    waiting for it, grid update, CFL allreduce)."""
    NEV = 8
    STRETCHES = (("boundary_range_ms", 0, 1), ("interior_range_ms", 1, 2), ("wait_for_exchange_ms", 2, 5), ("grid_update_ms", 5, 6),
                 ("cfl_allreduce_ms", 6, 7), ("exchange_side_stream_ms", 3, 4), ("step_ms", 0, 7),
                 ("both_ranges_ms", 0, 2))   # (ranges side by side: [0,1] and [0,2] overlap, interior_range_ms = [1,2] is only the rest)

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.steps = []

    def next(self):
        arr = (C.c_void_p * self.NEV)()
        for k in range(self.NEV):
            e = C.c_void_p()
            if self.hip.hipEventCreate(C.byref(e)) != 0:
                raise RuntimeError("hipEventCreate failed")
            arr[k] = e
        self.steps.append(arr)
        return arr

    def summary(self):
        out = {}
        for name, i, j in self.STRETCHES:
            v = []
            for arr in self.steps:
                ms = C.c_float(0)
                if self.hip.hipEventElapsedTime(C.byref(ms), C.c_void_p(arr[i]), C.c_void_p(arr[j])) == 0:
                    v.append(ms.value)
            out[name] = (sum(v) / len(v)) if v else None
        return out

    def __del__(self):
        try:
            for arr in self.steps:
                for k in range(self.NEV):
                    self.hip.hipEventDestroy(C.c_void_p(arr[k]))
        except Exception:
            pass


PLANE, CUBOID, SPHERE, CYLINDER = 0, 1, 2, 3   # analytic_geometry_e members GeneralBoundary holds (geometry/Collider.h:246-252)
STICKY, SLIP, SEPARATE = 0, 1, 2               # collider_e (geometry/Collider.h:8)


def make_collider(geometry, ctype, param, s=1.0, dsdt=0.0, R=None, omega=(0, 0, 0), b=(0, 0, 0), dbdt=(0, 0, 0)):
    """zs::Collider<AnalyticLevelSet<geometry, f32, 3>>{levelset(param), ctype} with setTranslation / setRotation / scale.
    param: plane (origin xyz, normal xyz), cuboid (min xyz, max xyz), sphere (centre xyz, radius), cylinder (bottom xyz, radius,
    length, axis)."""
    from ._lib import Collider
    c = Collider()
    p = [float(v) for v in param]
    lib().zs_rocm_collider_init(C.byref(c), int(geometry), int(ctype), (C.c_float * len(p))(*p), len(p))
    c.s, c.dsdt = float(s), float(dsdt)
    if R is not None:
        c.R = (C.c_float * 9)(*[float(v) for v in (R.reshape(-1) if hasattr(R, "reshape") else R)])
    c.omega = (C.c_float * 3)(*[float(v) for v in omega])
    c.b = (C.c_float * 3)(*[float(v) for v in b])
    c.dbdt = (C.c_float * 3)(*[float(v) for v in dbdt])
    return c
