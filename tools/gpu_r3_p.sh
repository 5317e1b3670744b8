#!/bin/bash
# round-3 call P: TN tilings: kernel tests, step A/B, in-step duration of every TN launch (by grid), one stream
TAG=${1:-r03p}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" "TFX_TN_TILE=0"; do env $v python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm_tn or pull" > gpurun_out/${TAG}_pytest_k.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest_k.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest_k.log | head -20; done
for i in 1 2 3; do
  for v in "TFX_TN_TILE=-1" "TFX_TN_TILE=2" "TFX_TN_TILE=0" "TFX_TN_TILE=1"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
for v in -1 2 0; do
(cd /tmp && TFX_TN_TILE=$v TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt$v -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
python tools/prof_summary.py /tmp/pt$v/p_kernel_trace.csv --steady --by-grid gemm_tn > gpurun_out/${TAG}_tile${v}_kernel_summary.txt; echo "TFX_TN_TILE=$v"; grep "total\|gemm_tn" gpurun_out/${TAG}_tile${v}_kernel_summary.txt
done
