#!/bin/bash
# round-3 call I: all GPU tests; ragged steady state with / without the structure prefetch; big sampling golden detail
TAG=${1:-r03i}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest.log | head -30
python -m pytest tests/test_sampling_gpu.py -q -s -k "config5 or big" 2>&1 | grep "\[big\]" | cut -c1-200
python -m pytest tests/test_model_gpu.py -q -s -k "cfg3_1024 or canon512" 2>&1 | grep "argmax agreement\|gradients:" | cut -c1-200
for i in 1 2; do
  for v in "TFX_PREFETCH=1" "TFX_PREFETCH=0"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --ragged-steps 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k: round(d[k],2) for k in ('ms_per_step','host_ms_per_step','structure_miss_ms','ragged_ms_per_step')})"
  done
done
