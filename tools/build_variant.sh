#!/bin/bash
# build a second copy of the library for same-box A/B runs (TFX_LIB=<path>): tools/build_variant.sh <name> [git-rev|WORK] [extra hipcc flags...]
# output: transfusion_pytorch_amd/lib/libtfx_<name>.so (git-ignored, travels with gpurun)
set -e
NAME=$1; REV=${2:-WORK}; shift; shift || true
R=$(cd $(dirname $0)/.. && pwd)
W=$(mktemp -d)
mkdir -p $W/pkg/csrc $W/include          # the sources include "../../include/tfx.h" relative to csrc
if [ "$REV" = WORK ]; then
  cp $R/transfusion_pytorch_amd/csrc/* $W/pkg/csrc/; cp $R/include/tfx.h $W/include/
else
  git -C $R archive $REV transfusion_pytorch_amd/csrc include/tfx.h | tar -x -C $W                     # the whole directory: the generated asm loops (*.inc) are sources too
  cp $W/transfusion_pytorch_amd/csrc/* $W/pkg/csrc/
fi
OBJS=""
for s in gemm attention tokenwise decode collective runner; do
  [ -f $W/pkg/csrc/$s.hip ] || continue
  FF=""; [ $s = attention ] && FF="-fno-slp-vectorize"      # (every revision: the product build always passes it, build.py FILE_FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $FF "$@" -c $W/pkg/csrc/$s.hip -o $W/$s.o &
  OBJS="$OBJS $W/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o $R/transfusion_pytorch_amd/lib/libtfx_$NAME.so
rm -rf $W
echo built $R/transfusion_pytorch_amd/lib/libtfx_$NAME.so
