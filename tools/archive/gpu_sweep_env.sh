#!/bin/bash
# sweep one env knob on one box: bash tools/gpu_sweep_env.sh VAR v1 v2 ...   (two passes; prints ms/step and NT TFLOP/s)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VAR=$1; shift
for i in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],2), round(d['roofline']['achieved'],1))"
  done
done
