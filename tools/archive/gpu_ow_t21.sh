#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow21.txt; : > $OUT
export TFX_NT_PP_MIN=1
for rep in 1 2; do for v in split3 sB sC sD; do
  OWP_REPS=600 TFX_LIB=transfusion_pytorch_amd/lib/libtfx_$v.so TFX_NT_OW=1 timeout 300 tools/ow_probe run $v n512k512,n512k2816,n1544k512,n1024k2752,sq4096 2>&1 | grep -v "^\[run" | awk -v s=$v -v r=$rep '{print "rep", r, s, $2, $(NF-5), $(NF-4), $(NF-3), $(NF-2), $NF}' | tee -a $OUT
done; done
