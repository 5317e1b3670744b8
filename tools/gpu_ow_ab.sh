#!/bin/bash
# same-box A/B of schedule variants of the one-wave-per-SIMD NT kernel: tools/gpu_ow_ab.sh <tag> <cases> <variant>...   (variant "pp" = the ping-pong kernel)
TAG=$1; CASES=$2; shift; shift
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}.txt
: > $OUT
tools/mfma_peak 2>/dev/null | grep -A2 "random" | head -3 >> $OUT
export TFX_NT_PP_MIN=1
for rep in 1 2; do
for v in "$@"; do
  if [ $v = pp ]; then L=transfusion_pytorch_amd/lib/libtfx_hip.so; OW=0; elif [ $v = base ]; then L=transfusion_pytorch_amd/lib/libtfx_hip.so; OW=2; else L=transfusion_pytorch_amd/lib/libtfx_$v.so; OW=2; fi
  TFX_LIB=$L TFX_NT_OW=$OW timeout 200 tools/ow_probe run $v "$CASES" 2>&1 | grep -v "^\[run" | sed "s/^/rep$rep /" >> $OUT
done
done
cat $OUT
for v in "$@"; do
  case $v in pp|nodma|nord) ;; *) tools/ow_probe cmp pp $v | grep -v MISSING | sed "s/^/$v /" | tee -a $OUT;; esac
done
