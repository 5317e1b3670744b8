#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "pull or attnres" 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests/test_model_gpu.py tests/test_f4b_gpu.py -m gpu -q -x -k "golden" 2>&1 | grep -E "passed|failed" | tail -2
(cd /tmp && rm -rf /tmp/p4 && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p4 -o p -- python $R/tools/bench_configs.py 4 > /tmp/p4.log 2>&1)
python tools/prof_summary.py /tmp/p4/p_kernel_trace.csv --steady | grep "total\|pull" | cut -c1-120; grep "^config 4" /tmp/p4.log | cut -c1-100
(cd /tmp && rm -rf /tmp/pt && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady | grep "total\|pull" | cut -c1-120
