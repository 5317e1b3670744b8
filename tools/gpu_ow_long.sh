#!/bin/bash
# short bursts against the power-limited steady state: the same probe cases with 20 and with 1500 timed launches
mkdir -p gpurun_out; OUT=gpurun_out/${1:-owl}_long.txt; : > $OUT
export TFX_NT_PP_MIN=1
for reps in 20 1500; do
  for m in 0 1; do
    OWP_REPS=$reps TFX_TN_OW=$m timeout 300 tools/ow_probe tn tn$m 2>&1 | grep "t_2816\|t_1544\|t_1024" | sed "s/^/reps $reps TN_OW=$m /" >> $OUT
  done
  for m in 0 1; do
    OWP_REPS=$reps TFX_NT_OW=$m timeout 300 tools/ow_probe run nt$m n512k512,n512k2816,n1544k512,sq4096 2>&1 | grep -v "^\[run" | sed "s/^/reps $reps NT_OW=$m /" >> $OUT
  done
done
cut -c1-200 $OUT
