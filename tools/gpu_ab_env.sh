#!/bin/bash
# A/B of an env toggle on one box: bash tools/gpu_ab_env.sh VAR A B   (bench 3x each, alternating; prints ms/step and NT TFLOP/s)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do
  for v in "$2" "$3"; do
    env $1=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1=$v', round(d['ms_per_step'],2), round(d['roofline']['achieved'],1))"
  done
done
