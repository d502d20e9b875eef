#!/bin/bash
# Derives tests/golden/abi_layout.json from the reference's py_interop headers (build container only: needs /root/reference and
# the fmt symlink `make -C oracle ref` sets up).  No reference source is copied; the output is a table of offsets.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle ref >/dev/null
mkdir -p oracle/_ref
g++ -std=c++17 -fpermissive -w -DFMT_HEADER_ONLY=1 -DZS_ENABLE_OPENMP=0 -DZS_ENABLE_CUDA=0 -DZS_ENABLE_MUSA=0 -DZS_ENABLE_ROCM=0 \
  -DZS_ENABLE_SYCL=0 -DZS_ENABLE_SYCL_ONEAPI=0 -DZS_ENABLE_SYCL_ACPP=0 -DZS_ENABLE_VULKAN=0 -DZS_ENABLE_JIT=0 \
  -DZS_ENABLE_OFB_ACCESS_CHECK=0 -DZS_ENABLE_SERIALIZATION=0 -DZS_ENABLE_OPENVDB=0 \
  -I oracle/_ref/shim -I ${REF:-/root/reference}/include tools/gen_abi_layout.cpp -o oracle/_ref/gen_abi_layout
oracle/_ref/gen_abi_layout > tests/golden/abi_layout.json
python3 -c "import json;d=json.load(open('tests/golden/abi_layout.json'));print(len(d),'structs')"
