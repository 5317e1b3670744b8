#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow40.txt; : > $OUT
for cfg in 3 4; do for m in "TFX_SIDE_STREAM=0" "TFX_SIDE_STREAM=1" "TFX_TN_GROUP=0"; do
  env $m python bench.py --config $cfg --steps 5 --warmup 2 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/st.log 2>&1
  echo "config $cfg [$m]: $(python -c "import json;d=json.loads(open('/tmp/st.log').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms/step', round(d['value'],1), d['unit'])" 2>&1 | tail -1)" | tee -a $OUT
done; done
