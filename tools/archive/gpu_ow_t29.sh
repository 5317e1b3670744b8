#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow29.txt; : > $OUT
for rep in 1 2; do for rot in 0 1; do
  OWP_REPS=400 TFX_TN_ROT=$rot TFX_TN_OW=1 timeout 300 tools/ow_probe tn r$rot 2>&1 | grep "t_" | awk -v m=$rot -v r=$rep '{print "rep", r, "rot", m, $2, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}' | tee -a $OUT
done; done
tools/ow_probe tncmp r0 r1 | tail -3 | tee -a $OUT
