#!/bin/bash
# VGPR / SGPR / spill / LDS figures of the kernels of a built library whose (mangled) name matches a pattern:
#   tools/kernel_regs.sh orientedreppoints_amd/csrc/liborp_hip.so dcn_fwd_split
T=$(mktemp -d); cp "$1" $T/lib.so; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1
for f in lib.so.*gfx950*; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null; done) | python3 -c "
import sys,re
pat=sys.argv[1]
for b in sys.stdin.read().split('- .agpr_count')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',b) or [None,'?'])[1]
    if pat in g('name'): print(g('name')[:100],'vgpr',g('vgpr_count'),'agpr',b.split()[1] if b.split() else '?','spill',g('vgpr_spill_count'),'sgpr',g('sgpr_count'),'scratch',g('private_segment_fixed_size'))
" "$2"; rm -rf $T
