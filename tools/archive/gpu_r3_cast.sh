#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "cast or plumbing" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_f4b_gpu.py -m gpu -q -x 2>&1 | tail -2
(cd /tmp && rm -rf /tmp/pt && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady | grep "total\|cast_batch\|adam_k"
(cd /tmp && rm -rf /tmp/p3 && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o p -- python $R/tools/bench_configs.py 3 > /tmp/p3.log 2>&1)
python tools/prof_summary.py /tmp/p3/p_kernel_trace.csv --steady | grep "total\|cast_batch\|adam_k"; grep "config 3" /tmp/p3.log
