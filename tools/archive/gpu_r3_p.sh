#!/bin/bash
# round-3 call P: TN kernels: kernel tests, microbenchmark, step A/B against a second library (TFX_LIB), in-step duration per TN launch
TAG=${1:-r03p}; BASE=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "TFX_TN_TILE=-1" "TFX_TN_TILE=2" "TFX_TN_TILE=0"; do env $v python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm_tn or pull" > gpurun_out/${TAG}_pytest_k.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest_k.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest_k.log | head -20; done
python tools/bench_gemm.py tn 2>&1 | grep "splits=  0\|splits= 16\|splits=  8"
for i in 1 2 3; do
  for v in "TFX_TN_TILE=-1" ${BASE:+"TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$BASE.so"}; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v'[:20], round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady --by-grid gemm_tn > gpurun_out/${TAG}_kernel_summary.txt; grep "total\|gemm_tn" gpurun_out/${TAG}_kernel_summary.txt | cut -c1-150
python -m pytest tests/test_model_gpu.py tests/test_f4b_gpu.py -m gpu -q -x > gpurun_out/${TAG}_pytest_m.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest_m.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest_m.log | head -20
