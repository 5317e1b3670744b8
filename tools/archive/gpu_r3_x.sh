#!/bin/bash
# round-3 call X: NT epilogue changes: GEMM tests, microbenchmark, step A/B against a second library, in-step duration per NT launch (by grid)
TAG=${1:-r03x}; BASE=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" > gpurun_out/${TAG}_pytest_k.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest_k.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest_k.log | head -20
for v in "A=1" ${BASE:+"TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$BASE.so"}; do echo "$v" | cut -c1-20; env $v python tools/bench_gemm.py 2>&1 | grep "^NT"; done
for i in 1 2 3; do
  for v in "A=1" ${BASE:+"TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$BASE.so"}; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v'[:20], round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady --by-grid gemm_nt > gpurun_out/${TAG}_kernel_summary.txt; grep "total\|gemm_nt" gpurun_out/${TAG}_kernel_summary.txt | cut -c1-150
