#!/bin/bash
# training-step A/B of environment settings on one box: tools/gpu_ow_step2.sh <tag> "<ENV=.. ENV=..>" "<...>" ...   (2 rounds)
TAG=$1; shift
mkdir -p gpurun_out; OUT=gpurun_out/${TAG}.txt; : > $OUT
for r in 1 2; do for m in "$@"; do
  env $m python bench.py --steps 10 --warmup 3 --family-steps 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/st.log 2>&1
  echo "round $r [$m]: $(python -c "import json;d=json.loads(open('/tmp/st.log').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms/step', round(d['value'],1), [ (f['kernel'][:12], round(f['ms_per_step'],2)) for f in d['roofline_by_family']])" 2>&1 | tail -1)" | tee -a $OUT
done; done
