#!/bin/bash
# register budget of every kernel (VGPRs, spills, LDS): run after ANY kernel edit - a spilling hot loop costs more than any schedule gains
# usage: bash tools/check_spills.sh
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
for f in gemm attention tokenwise; do
  FF=""; [ $f = attention ] && FF="-fno-slp-vectorize"      # build.py FILE_FLAGS
  (cd $R/transfusion_pytorch_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC $FF -I../../include -c $f.hip -o $T/$f.o -save-temps=obj 2>/dev/null)
done
python3 - $T <<'PY'
import re, sys, glob
bad = 0
for path in sorted(glob.glob(sys.argv[1] + '/*gfx950.s')):
    txt = open(path).read()
    for m in re.finditer(r'\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.vgpr_count:\s*(\d+)\s*\n\s*\.vgpr_spill_count:\s*(\d+)', txt, re.S):
        lds, name, vg, sp = int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4))
        short = re.sub(r'^_ZN3tfx\d+', '', name)[:60]
        flag = '  <-- SPILLS' if sp else ''
        if sp: bad += 1
        if sp or vg > 128: print(f'{short:62s} vgpr {vg:3d} spill {sp:3d} lds {lds}{flag}')
print('kernels with spills:', bad)
PY
rm -rf $T
