#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow22.txt; : > $OUT
for rep in 1 2; do for v in hip tnsplit; do
  OWP_REPS=400 TFX_LIB=transfusion_pytorch_amd/lib/libtfx_$v.so TFX_TN_OW=1 timeout 300 tools/ow_probe tn x_$v 2>&1 | grep "t_" | awk -v s=$v -v r=$rep '{print "rep", r, s, $2, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}' | tee -a $OUT
done; done
