#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "losses" 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "golden" 2>&1 | grep -E "passed|failed" | tail -2
(cd /tmp && rm -rf /tmp/pt && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady | grep "total\|ce_k\|rmsnorm" | cut -c1-120
