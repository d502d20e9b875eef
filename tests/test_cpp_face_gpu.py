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
