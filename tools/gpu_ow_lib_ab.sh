#!/bin/bash
# training-step A/B of two library builds on one box: tools/gpu_ow_lib_ab.sh <tag> <libA> <libB> ... (names under transfusion_pytorch_amd/lib/libtfx_<name>.so; 2 rounds, family times)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; OUT=gpurun_out/${TAG}.txt; : > $OUT
for r in 1 2; do for v in "$@"; do
  TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$v.so python bench.py --steps 10 --warmup 3 --family-steps 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/st.log 2>&1
  echo "round $r [$v]: $(python -c "import json;d=json.loads(open('/tmp/st.log').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms/step', round(d['value'],1), [ (f['kernel'][:12], round(f['ms_per_step'],2)) for f in d['roofline_by_family']])" 2>&1 | tail -1)" | tee -a $OUT
done; done
