#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for m in ring splitk; do
  echo "== TFX_SKINNY=$m"
  TFX_SKINNY=$m python tools/bench_sample.py 2>&1 | grep -v amdgpu
  TFX_SKINNY=$m python tools/bench_configs.py 5 2>&1 | grep -v amdgpu
done
