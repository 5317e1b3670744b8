#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for i in 1 2 3; do
  for v in "TFX_PP_STAGGER=18000" "TFX_PP_STAGGER=12000"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2))"
  done
done
