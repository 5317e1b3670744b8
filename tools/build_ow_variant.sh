#!/bin/bash
# a copy of the library whose one-wave-per-SIMD NT kernels run a schedule variant of tools/gen_nt_ow_loop.py (same-box A/B through TFX_LIB):
#   tools/build_ow_variant.sh <name> [generator options...]        e.g.  tools/build_ow_variant.sh nodma --nodma
# only gemm.hip is recompiled; the other objects come from the product build (python -m transfusion_pytorch_amd.build).
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
L=$R/transfusion_pytorch_amd/lib
W=$(mktemp -d)
mkdir -p $W/pkg/csrc $W/include
cp $R/transfusion_pytorch_amd/csrc/* $W/pkg/csrc/; cp $R/include/tfx.h $W/include/
python3 $R/tools/gen_nt_ow_loop.py --out $W/pkg/csrc/gemm_nt_ow_loop.inc "$@" > /dev/null
[ -n "$TN_GEN_ARGS" ] && python3 $R/tools/gen_tn_ow_loop.py --out $W/pkg/csrc/gemm_tn_ow_loop.inc $TN_GEN_ARGS > /dev/null      # (weight-gradient loop variants: TN_GEN_ARGS="--split 1 ...")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $TFX_HIPCC_EXTRA -c $W/pkg/csrc/gemm.hip -o $W/gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $W/gemm.o $L/attention.o $L/tokenwise.o $L/decode.o $L/collective.o $L/runner.o -ldl -o $L/libtfx_$NAME.so
rm -rf $W
echo built $L/libtfx_$NAME.so "$@"
