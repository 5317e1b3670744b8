#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow41.txt; : > $OUT
for rep in 1 2; do for cfg in 3 4; do for m in "TFX_SIDE_STREAM=0" "TFX_SIDE_STREAM=1"; do
  env $m python bench.py --config $cfg --steps 5 --warmup 2 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/st.log 2>&1
  echo "rep $rep config $cfg [$m]: $(python -c "import json;d=json.loads(open('/tmp/st.log').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms/step', round(d['value'],1), d['unit'])" 2>&1 | tail -1)" | tee -a $OUT
done; done; done
