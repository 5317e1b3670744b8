#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/ow20.txt; : > $OUT
for m in "TFX_NT_OW=1" "TFX_NT_OW=2"; do
  env $m python bench.py --config 3 --steps 5 --warmup 2 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/st.log 2>&1
  echo "config 3 [$m]: $(python -c "import json;d=json.loads(open('/tmp/st.log').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms/step')" 2>&1 | tail -1)" | tee -a $OUT
done
export TFX_NT_PP_MIN=1
for rep in 1 2; do for v in hip split3 r2; do
  OWP_REPS=600 TFX_LIB=transfusion_pytorch_amd/lib/libtfx_$v.so TFX_NT_OW=1 timeout 300 tools/ow_probe run $v n512k512,n512k2816,n1544k512,n1024k2752,sq4096 2>&1 | grep -v "^\[run" | awk -v s=$v -v r=$rep '{print "rep", r, s, $2, $(NF-5), $(NF-4), $(NF-3), $(NF-2), $NF}' | tee -a $OUT
done; done
