#!/bin/bash
# Registers / scratch / LDS of every kernel in a built object: tools/kernel_regs.sh coclr_amd/csrc/build/conv_igemm.o
set -e
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
    --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
rows = []
for blk in txt.split('- .agpr_count:')[1:]:
    ag = blk.split()[0]
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    vg = re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1)
    sg = re.search(r'\.sgpr_count:\s+(\d+)', blk).group(1)
    sp = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1)
    rows.append((name, vg, ag, sg, sp))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = n.replace('(anonymous namespace)::', '').split('(')[0]
    print('vgpr %3s agpr %3s sgpr %3s scratch %4s  %s' % (r[1], r[2], r[3], r[4], n))
"
rm -rf $tmp
