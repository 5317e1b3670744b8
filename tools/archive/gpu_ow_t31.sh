#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow31.txt; : > $OUT
for rep in 1 2; do for rf in 0 0.5 1 1.5 2; do
  OWP_REPS=400 TFX_TN_RAMP=$rf TFX_TN_OW=1 timeout 300 tools/ow_probe tn rp$rf 2>&1 | grep "t_" | awk -v m=$rf -v r=$rep '{print "rep", r, "ramp", m, $2, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}' | tee -a $OUT
done; done
tools/ow_probe tncmp rp0 rp1 | tail -9 | tee -a $OUT
