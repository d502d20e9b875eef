"""The header-only C++ face (include/zensim_rocm/zs_rocm.hpp): a user TU mirroring the reference's own CUDA tests
(test/cuda/main.cu, test/cuda/basic.cu) compiled with hipcc against libzsrocm.so and run on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_face_program_runs():
    exe = os.path.join(ROOT, "zpc_amd", "lib", "test_cpp_face")
    if not os.path.exists(exe):
        from zpc_amd import build
        build.build_cpp_face_test()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"cpp face ok" in r.stdout, r.stdout.decode()


def test_grid_arena_matches_the_references_own_grid_arena(tmp_path):
    """SparseGridView::iArena / GridArena of the C++ face against tests/golden/grid_arena.npz -- outputs of the REFERENCE's GridArena
    (math/curve/InterpolationKernel.hpp:271-560: constructors, weights and derivatives of the six kernels, isample, minimum, maximum,
    weight, weightsGradient; collocated and staggered) instantiated over a dense box of the same values (oracle/ref_shim.cpp).  The
    fixture is flattened to a binary file for the compiled test program."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "grid_arena.npz"))
    cases, out = z["cases"].astype(np.int32), z["out"].astype(np.float32)
    blob = tmp_path / "grid_arena.bin"
    with open(blob, "wb") as f:
        np.array([cases.shape[0], z["X"].shape[0], int(z["ext"])] + [int(v) for v in z["lo"]], np.int32).tofile(f)
        np.array([float(z["dx"]), float(z["default"])], np.float32).tofile(f)
        cases.tofile(f)
        z["data"].astype(np.float32).tofile(f)
        z["X"].astype(np.float32).tofile(f)
        out.tofile(f)
    exe = os.path.join(ROOT, "zpc_amd", "lib", "test_cpp_face")
    if not os.path.exists(exe):
        from zpc_amd import build
        build.build_cpp_face_test()
    r = subprocess.run([exe, "--grid-arena", str(blob)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b" 0 mismatches" in r.stdout, r.stdout.decode()[-3000:]


def test_views_with_the_ofb_access_check_option():
    """ZS_ENABLE_OFB_ACCESS_CHECK=1: out-of-range view accesses yield the sentinel reference (tests/cpp/test_ofb.hip)"""
    exe = os.path.join(ROOT, "zpc_amd", "lib", "test_ofb")
    if not os.path.exists(exe):
        from zpc_amd import build
        build.build_ofb_test()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"ofb access checks: 0 failures" in r.stdout, r.stdout.decode()[-2000:]
