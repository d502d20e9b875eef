#!/bin/bash
# Measurement-only builds of libzsrocm.so with parts of the slotted fused step stubbed (-DZS_X_*; results are WRONG by construction):
#   tools/ablate_slot.sh NOFINAL NOTICKET "NOFINAL NOTICKET" ...   -> zpc_amd/lib/ablate/libzsrocm_slot_<name>.so (select with ZS_ROCM_LIB)
set -e
cd "$(dirname "$0")/.."
mkdir -p zpc_amd/lib/ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result -I include -fno-slp-vectorize -DZS_SLOT_FAST_BUILD"
for v in "$@"; do
  name=$(echo $v | tr ' ' '_')
  defs=""; for d in $v; do if [ "$d" = PROBE ]; then defs="$defs -DZS_SLOT_PROBE"; elif [ "$d" != BASE ]; then defs="$defs -DZS_X_$d"; fi; done; defs=$(echo "$defs" | sed "s/-DZS_X_PRIO\([0-9]\)/-DZS_X_PRIO=\1/; s/-DZS_X_ARRQ\([0-9]\)/-DZS_SL_ARRQ=\1/")
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c zpc_amd/csrc/mpm_slotted.hip -o zpc_amd/lib/ablate/mpm_slotted_$name.o
    objs=$(ls zpc_amd/lib/obj/*.o | grep -v mpm_slotted.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o zpc_amd/lib/ablate/libzsrocm_slot_$name.so $objs zpc_amd/lib/ablate/mpm_slotted_$name.o -L/opt/rocm/lib -lrccl
    echo built zpc_amd/lib/ablate/libzsrocm_slot_$name.so ) &
done
wait
