#!/bin/bash
# training-step A/B of the NT kernel switch on one box: tools/gpu_ow_step.sh <tag> <mode>...   (each mode = a TFX_NT_OW value; 2 rounds)
TAG=$1; shift
mkdir -p gpurun_out; OUT=gpurun_out/${TAG}.txt; : > $OUT
for r in 1 2; do for m in "$@"; do
  TFX_NT_OW=$m python bench.py --steps 10 --warmup 3 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/st.log 2>&1
  echo "round $r TFX_NT_OW=$m: $(python -c "import json;d=json.loads(open('/tmp/st.log').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms/step', round(d['value'],1), d['unit'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'])" 2>&1 | tail -1)" | tee -a $OUT
done; done
