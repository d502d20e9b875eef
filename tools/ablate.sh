#!/bin/bash
# Measurement-only builds of libzsrocm.so with parts of the fused G2P2G kernel stubbed (-DZS_ABLATE_*), to read off the
# marginal cost of the constitutive update / the phase-2 accumulation / the gather.  Results are WRONG by construction; the
# libraries land in zpc_amd/lib/ablate/ and are selected with ZS_ROCM_LIB=... (announced on stdout by zpc_amd._lib).
#   tools/ablate.sh STRESS CONSUME GATHER "STRESS CONSUME"
set -e
cd "$(dirname "$0")/.."
mkdir -p zpc_amd/lib/ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result -I include -fno-slp-vectorize -DZS_FUSED_FAST_BUILD -DZS_ROCM_WITH_PERSIST"
for v in "$@"; do
  name=$(echo $v | tr ' ' '_')
  defs=""; case "$v" in *STRESS*|*CONSUME*|*GATHER*|*PROLOGUE*|*EPILOGUE*) defs="-DZS_ABLATE_FREEZE";; esac
  for d in $v; do if [ "$d" = PROBE ]; then defs="$defs -DZS_PROBE"; else defs="$defs -DZS_ABLATE_$d"; fi; done
  ( /opt/rocm/bin/hipcc $FLAGS $defs $EXTRA -c zpc_amd/csrc/mpm_fused8.hip -o zpc_amd/lib/ablate/mpm_fused8_$name.o
    objs=$(ls zpc_amd/lib/obj/*.o | grep -v mpm_fused8.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o zpc_amd/lib/ablate/libzsrocm_$name.so $objs zpc_amd/lib/ablate/mpm_fused8_$name.o
    echo built zpc_amd/lib/ablate/libzsrocm_$name.so ) &
done
wait
