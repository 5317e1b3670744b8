#!/bin/bash
# round-3 call M: wide-wave-tile TN kernels: tests, microbenchmark and step A/B (TFX_TN_TILE = 2: 256 x 256 / 8 waves, 1: 4-wave wide, 0: 128 x 128;
# TFX_TN_SPREAD = 1: DMA pieces between the MFMAs, 0: burst behind the barrier)
TAG=${1:-r03m}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for v in 2 1; do TFX_TN_TILE=$v python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm_tn or pull" > gpurun_out/${TAG}_pytest_k$v.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest_k$v.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest_k$v.log | head -20; done
for v in "TFX_TN_TILE=2 TFX_TN_SPREAD=1" "TFX_TN_TILE=1 TFX_TN_SPREAD=1" "TFX_TN_TILE=1 TFX_TN_SPREAD=0" "TFX_TN_TILE=0"; do echo "$v"; env $v python tools/bench_gemm.py tn 2>&1 | grep -v "splits= 64\|splits= 32"; done
for i in 1 2 3; do
  for v in "TFX_TN_TILE=2 TFX_TN_SPREAD=1" "TFX_TN_TILE=1 TFX_TN_SPREAD=1" "TFX_TN_TILE=1 TFX_TN_SPREAD=0" "TFX_TN_TILE=0"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
