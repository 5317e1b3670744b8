#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow28.txt; : > $OUT
for M in 65536 32768 16384 8192; do
  OWP_TN_M=$M OWP_REPS=400 TFX_TN_OW=1 timeout 300 tools/ow_probe tn m$M 2>&1 | grep "t_" | awk -v m=$M '{print "M", m, $2, "splits", $14, "grid", $16, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}' | tee -a $OUT
done
