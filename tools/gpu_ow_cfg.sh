#!/bin/bash
# configs 3 / 4 (and 2) with and without the one-wave-per-SIMD GEMM kernels, one box
TAG=$1; shift
mkdir -p gpurun_out; OUT=gpurun_out/${TAG}.txt; : > $OUT
for cfg in 3 4 2; do for m in "TFX_NT_OW=0 TFX_TN_OW=0" "TFX_NT_OW=1 TFX_TN_OW=1"; do
  env $m python bench.py --config $cfg --steps 5 --warmup 2 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/st.log 2>&1
  echo "config $cfg [$m]: $(python -c "import json;d=json.loads(open('/tmp/st.log').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms/step', round(d['value'],1), d['unit'])" 2>&1 | tail -1)" | tee -a $OUT
done; done
export TFX_NT_PP_MIN=1
for stg in 0 6000 12000; do
  OWP_REPS=600 TFX_PP_STAGGER=$stg TFX_NT_OW=1 timeout 300 tools/ow_probe run st$stg n512k512,n512k2816,n1544k512,n1024k1024 2>&1 | grep -v "^\[run" | awk -v s=$stg '{print "stagger", s, $2, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}' | tee -a $OUT
done
