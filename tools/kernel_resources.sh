#!/bin/bash
# Register / LDS / scratch use of the kernels in an object file built by hipcc (gfx950 code object in .hip_fatbin):
#   bash tools/kernel_resources.sh build/obj/env_kernels.o [name-regex]
set -e
F=${1:-build/obj/env_kernels.o}; PAT=${2:-.}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $F
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co > $T/notes.txt
PAT="$PAT" python3 - $T/notes.txt <<'P'
import sys, re, os
txt = open(sys.argv[1]).read()
pat = os.environ["PAT"]
for blk in re.split(r'\n\s*- \.agpr_count:', txt)[1:]:
    blk = '.agpr_count:' + blk
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if re.search(pat, name):
        print('%-60s vgpr %s agpr %s sgpr %s lds %s scratch %s spill_v %s spill_s %s' % (name[:60], g('vgpr_count'), g('agpr_count'), g('sgpr_count'),
              g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count'), g('sgpr_spill_count')))
P
rm -rf $T
