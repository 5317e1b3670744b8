#!/bin/bash
# round-3: GEGLU-backward epilogue with coalesced [a|g] loads: tests, in-step kernel time, step A/B against a second library
BASE=${1:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "golden" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do
  for v in "A=1" ${BASE:+"TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$BASE.so"}; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v'[:20], round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
for v in "A=1" ${BASE:+"TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$BASE.so"}; do
(cd /tmp && rm -rf /tmp/pt && env $v TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
echo "$v" | cut -c1-20; python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady | grep "total\|gemm_nt_pp_kernel<5>\|gemm_nt_pp_kernel<3>" | cut -c1-120
done
