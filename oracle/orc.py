"""ctypes wrapper of the CPU parity checker (oracle/libzpc_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by tests/ (through
tests/util.py), by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py -- never by the product path (zpc_amd/)."""
import ctypes as C

import numpy as np


def ptr(a, ct=None):
    return a.ctypes.data_as(C.c_void_p)


class OrcMpmParams(C.Structure):
    _fields_ = [("model", C.c_int), ("dx", C.c_float), ("dt", C.c_float), ("volume", C.c_float), ("E", C.c_float),
                ("nu", C.c_float), ("cohesion", C.c_float), ("beta", C.c_float), ("yieldSurface", C.c_float),
                ("volCorrection", C.c_int), ("side", C.c_int), ("nthreads", C.c_int), ("yieldStress", C.c_float),
                ("xi", C.c_float), ("Msqr", C.c_float), ("hardeningOn", C.c_int), ("bulk", C.c_float), ("viscosity", C.c_float),
                ("hostVariant", C.c_int)]


YIELD_SURFACE = 0.816496580927726 * 2.0 * 0.5 / (3.0 - 0.5)  # DruckerPragerConfig default


class OracleMpm:
    """Drives oracle/mpm.c + oracle/bht.c: the CPU restatement of partition build, P2G, grid update, G2P."""

    def __init__(self, oracle, model, dx, dt, side, volume, E=5e4, nu=0.4, nthreads=1, cohesion=0.0, beta=1.0, yield_stress=240e6,
                 xi=0.8, friction_angle=45.0, hardening=True, bulk=4e4, viscosity=0.0, host_variant=0):
        self.o = oracle
        oracle.orc_nacc_msqr.restype = C.c_float
        self.p = OrcMpmParams(model, dx, dt, volume, E, nu, cohesion, beta, YIELD_SURFACE, 1, side, nthreads, yield_stress, xi,
                              oracle.orc_nacc_msqr(C.c_float(friction_angle)), int(hardening), bulk, viscosity, int(host_variant))
        self.side = side
        self.o.orc_bht_create.restype = C.c_void_p
        self.o.orc_bht_active_keys.restype = C.POINTER(C.c_int32)
        self.o.orc_bht_size.restype = C.c_int32
        self.table = None

    def build_partition(self, pos, expected_blocks):
        n = pos.shape[0]
        self.table = C.c_void_p(self.o.orc_bht_create(3, C.c_size_t(expected_blocks)))
        self.o.orc_mpm_build_partition(self.table, ptr(pos), C.c_size_t(n), C.c_float(self.p.dx), self.side)
        self.nblocks = self.o.orc_bht_size(self.table)
        keys = np.ctypeslib.as_array(self.o.orc_bht_active_keys(self.table), shape=(self.nblocks, 3)).copy()
        self.keys = keys
        self.grid = np.zeros((self.nblocks, 7, self.side ** 3), np.float32)
        return self.nblocks

    def adopt_partition(self, keys):
        """Number the partition by a given key list (block i = keys[i]) instead of building it from particle positions."""
        keys = np.ascontiguousarray(keys, np.int32)
        self.table = C.c_void_p(self.o.orc_bht_create(3, C.c_size_t(keys.shape[0])))
        ret = np.zeros(keys.shape[0], np.int32)
        self.o.orc_bht_insert_many(self.table, ptr(keys), C.c_size_t(keys.shape[0]), ptr(ret))
        assert np.array_equal(ret, np.arange(keys.shape[0]))
        self.nblocks = keys.shape[0]
        self.keys = keys.copy()
        self.grid = np.zeros((self.nblocks, 7, self.side ** 3), np.float32)
        return self.nblocks

    def p2g(self, mass, pos, vel, Cm, F, logJp=None):
        n = pos.shape[0]
        lj = np.zeros(n, np.float32) if logJp is None else logJp
        self.o.orc_mpm_p2g(C.byref(self.p), self.table, C.c_size_t(n), ptr(mass), ptr(pos), ptr(vel), ptr(Cm), ptr(F), ptr(lj), ptr(self.grid))
        return lj

    def grid_update(self, extf=(0.0, 0.0, 0.0)):
        e = (C.c_float * 3)(*extf)
        mx = C.c_float(0.0)
        self.o.orc_mpm_grid_update(C.byref(self.p), C.c_size_t(self.nblocks), ptr(self.grid), e, C.byref(mx))
        return mx.value

    def g2p(self, pos, vel, Cm, F):
        n = pos.shape[0]
        self.o.orc_mpm_g2p(C.byref(self.p), self.table, C.c_size_t(n), ptr(pos), ptr(vel), ptr(Cm), ptr(F), ptr(self.grid))

    # ---- gather-style transfers (oracle/mpm.c: orc_mpm_p2c2g / orc_mpm_g2c2p / orc_mpm_post_g2c2p)
    def build_buckets(self, pos, displacement=0.0):
        """IndexBuckets of cell size dx, displacement 0 (sequential policy: ascending ids per bucket)."""
        n = pos.shape[0]
        I32P = C.POINTER(C.c_int32)
        self.o.orc_index_buckets_for_particles.restype = C.c_void_p
        self._ib = (I32P(), I32P(), I32P())
        self.buckets = C.c_void_p(self.o.orc_index_buckets_for_particles(ptr(pos), C.c_size_t(n), C.c_float(self.p.dx), C.c_float(displacement),
                                                                         C.c_size_t(0), C.byref(self._ib[0]), C.byref(self._ib[1]),
                                                                         C.byref(self._ib[2])))

    def p2c2g(self, kind, mass, pos, vel, Bm, F, logJp=None):
        n = pos.shape[0]
        lj = np.zeros(n, np.float32) if logJp is None else logJp
        self.o.orc_mpm_p2c2g(C.byref(self.p), int(kind), self.table, self.buckets, self._ib[1], self._ib[2], C.c_size_t(n), ptr(mass),
                             ptr(pos), ptr(vel), ptr(Bm), ptr(F), ptr(lj), ptr(self.grid))
        return lj

    def g2c2p(self, pos, vel, Bm, F):
        """Pre + G2C2P + Post on AoS arrays, in place."""
        n = pos.shape[0]
        vel[:] = 0
        Bm[:] = 0
        self.o.orc_mpm_g2c2p(C.byref(self.p), self.table, self.buckets, self._ib[1], self._ib[2], ptr(pos), ptr(vel), ptr(Bm), ptr(self.grid))
        self.o.orc_mpm_post_g2c2p(C.byref(self.p), C.c_size_t(n), ptr(pos), ptr(vel), ptr(Bm), ptr(F))

    def grid_by_key(self):
        return {tuple(int(x) for x in self.keys[i]): self.grid[i] for i in range(self.nblocks)}

    def __del__(self):
        try:
            if self.table:
                self.o.orc_bht_destroy(self.table)
        except Exception:
            pass
