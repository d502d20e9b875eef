"""Shared helpers for the parity tests: seeded inputs (xorshift64*) and oracle wrappers."""
import ctypes as C

import numpy as np

SEED0 = 0x9E3779B97F4A7C15


def rng(config_id=0):
    return np.random.Generator(np.random.PCG64(SEED0 ^ config_id))


def xorshift64star(n, seed):
    """The synthetic-input generator named by SURVEY.md 8(d); returns n uint64."""
    out = np.empty(n, dtype=np.uint64)
    x = np.uint64(seed if seed else 1)
    m = np.uint64(0x2545F4914F6CDD1D)
    with np.errstate(over="ignore"):
        for i in range(n):
            x ^= x >> np.uint64(12)
            x ^= x << np.uint64(25)
            x ^= x >> np.uint64(27)
            out[i] = x * m
    return out


def ptr(a, ct=None):
    return a.ctypes.data_as(C.c_void_p)


def orc_call(oracle, name, *args):
    f = getattr(oracle, name)
    f.restype = None
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(a.ctypes.data_as(C.c_void_p))
        elif isinstance(a, (int, np.integer)):
            conv.append(C.c_size_t(int(a)) if int(a) >= 0 else C.c_int(int(a)))
        else:
            conv.append(a)
    return f(*conv)
