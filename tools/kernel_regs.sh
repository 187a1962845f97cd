#!/bin/bash
# registers / scratch of the kernels in an object file's gfx950 code object: bash tools/kernel_regs.sh mx_deepim_amd/csrc/wino.o [filter]
o=$(realpath $1); d=$(mktemp -d); cd $d
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $o > /dev/null 2>&1
co=$(ls ${o}.0.hipv4-amdgcn-amd-amdhsa--gfx950 2>/dev/null || ls $(dirname $o)/*.hipv4-amdgcn-amd-amdhsa--gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if len(sys.argv)>1 and sys.argv[1] not in name: continue
    print('%-90s agpr %s vgpr %s sgpr %s scratch %s lds %s spill_v %s' % (name[:90], blk.split()[0], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('vgpr_spill_count')))
" $2
rm -f ${o}.0.host-* ${o}.0.hipv4-*; rm -rf $d
