#!/bin/bash
# weight-gradient (TN) kernels on the stand-alone probe: old 8-wave kernel (TFX_TN_OW=0) against the one-wave-per-SIMD kernel (TFX_TN_OW=2: ragged tiles too)
mkdir -p gpurun_out; OUT=gpurun_out/${1:-own}_tn.txt; : > $OUT
for r in 1 2; do
TFX_TN_OW=0 timeout 300 tools/ow_probe tn old 2>&1 | sed "s/^/rep$r /" >> $OUT
TFX_TN_OW=2 timeout 300 tools/ow_probe tn new 2>&1 | sed "s/^/rep$r /" >> $OUT
done
timeout 120 tools/ow_probe tncmp old new >> $OUT 2>&1
cat $OUT
