#!/bin/bash
# ab_build.sh NAME "EXTRA FLAGS" a.hip b.hip ...: a measurement build of libzsrocm.so in zpc_amd/lib/ablate/libzsrocm_NAME.so with the
# named translation units recompiled with EXTRA FLAGS (all other objects are taken from the product build).  Select with ZS_ROCM_LIB=...
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; shift 2
mkdir -p zpc_amd/lib/ablate/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result -I include"
objs=$(ls zpc_amd/lib/obj/*.o)
for f in "$@"; do
  b=$(basename $f .hip)
  fl="$FLAGS"; case $b in mpm_g2p) fl="$fl -fno-slp-vectorize -DZS_PSTORE_NT";; mpm*) fl="$fl -fno-slp-vectorize";; lbvh|collider) fl="$fl -ffp-contract=off";; esac
  /opt/rocm/bin/hipcc $fl $extra -c zpc_amd/csrc/$b.hip -o zpc_amd/lib/ablate/$name/$b.o &
  objs=$(echo "$objs" | grep -v "/$b.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o zpc_amd/lib/ablate/libzsrocm_$name.so $objs zpc_amd/lib/ablate/$name/*.o -L/opt/rocm/lib -lrccl
echo built zpc_amd/lib/ablate/libzsrocm_$name.so
