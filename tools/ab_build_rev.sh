#!/bin/bash
# ab_build_rev.sh NAME REV "EXTRA FLAGS" a.hip ...: like ab_build.sh, but the named translation units (and csrc/*.hpp, include/) are taken from git revision REV
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2; extra=$3; shift 3
T=$(mktemp -d); git archive $rev zpc_amd/csrc include | tar -x -C $T
mkdir -p zpc_amd/lib/ablate/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result -I $T/include"
objs=$(ls zpc_amd/lib/obj/*.o)
for f in "$@"; do
  b=$(basename $f .hip)
  fl="$FLAGS"; case $b in mpm_g2p) fl="$fl -fno-slp-vectorize -DZS_PSTORE_NT";; mpm*) fl="$fl -fno-slp-vectorize";; lbvh|collider) fl="$fl -ffp-contract=off";; esac
  /opt/rocm/bin/hipcc $fl $extra -c $T/zpc_amd/csrc/$b.hip -o zpc_amd/lib/ablate/$name/$b.o &
  objs=$(echo "$objs" | grep -v "/$b.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o zpc_amd/lib/ablate/libzsrocm_$name.so $objs zpc_amd/lib/ablate/$name/*.o -L/opt/rocm/lib -lrccl
rm -rf $T
echo built zpc_amd/lib/ablate/libzsrocm_$name.so
