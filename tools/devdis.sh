#!/bin/bash
# devdis.sh <object.o> [out.s]: disassemble the gfx950 code object inside a hipcc object file
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/p.fat
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/p.fat --output=$T/p.co --unbundle
$L/llvm-objdump -d $T/p.co > "${2:-/dev/stdout}"
rm -rf $T
