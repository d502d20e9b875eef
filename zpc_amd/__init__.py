"""zpc_amd -- MI355X (gfx950) backend for zpc's execution-policy hot path.

Python here is only the host-side mirror of the reference's ctypes-facing interface
(include/zensim/py_interop/): policies, typed primitives over iterators, container handles and the
MPM transfer entry points, all forwarding to the C ABI of libzsrocm.so (include/zs_rocm.h).
There is no CPU fallback: importing works without a GPU, every compute call needs the HIP library.
"""
from ._lib import lib, LIB_PATH, Port, Particles, MpmParams, BhtViewLite  # noqa: F401
from .policy import RocmExecutionPolicy, rocm_exec  # noqa: F401
from . import primitives  # noqa: F401
from .primitives import (reduce, exclusive_scan, inclusive_scan, radix_sort, radix_sort_pair,  # noqa: F401
                         merge_sort, merge_sort_pair,
                         plus, multiplies, getmin, getmax)
