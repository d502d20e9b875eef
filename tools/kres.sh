#!/bin/bash
# kres.sh <object.o> <kernel name regex>: VGPRs / AGPRs / SGPRs / scratch / LDS of the matching kernels in a hipcc object file
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/p.fat
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/p.fat --output=$T/p.co --unbundle
$L/llvm-readelf --notes $T/p.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    if not re.search(sys.argv[1],name): continue
    g=lambda k: re.search(r'\.'+k+r':\s+(\d+)',blk).group(1)
    print(name[:110], 'vgpr',g('vgpr_count'),'agpr',blk.split()[0],'sgpr',g('sgpr_count'),'scratch',g('private_segment_fixed_size'),'lds',g('group_segment_fixed_size'),'spill',g('vgpr_spill_count'))
" "$2"
rm -rf $T
