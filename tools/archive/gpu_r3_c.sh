#!/bin/bash
# round-3 call C: pull-form kernel variants (TFX_PULL_VARIANT) - kernel tests, model parity, same-box A/B at config 2 and config 3
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "pull or fused or attnres or adaln or layer_end" > gpurun_out/${TAG}_pytest_k.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_k.log
for v in 2 3 4; do TFX_PULL_VARIANT=$v python -m pytest tests/test_kernels_gpu.py -q -x -k "pull" 2>&1 | tail -1; done
python -m pytest tests/test_model_gpu.py -q -x > gpurun_out/${TAG}_pytest_m.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_m.log
for i in 1 2; do
  for v in 0 1 2 3 4; do
    TFX_PULL_VARIANT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TFX_PULL_VARIANT=$v', round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
export TMPDIR=/tmp
for v in 0 2; do
(cd /tmp && TFX_PULL_VARIANT=$v TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p3$v -o p -- python $R/tools/bench_configs.py 3 > $R/gpurun_out/${TAG}_cfg3_v$v.log 2>&1)
python tools/prof_summary.py /tmp/p3$v/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg3_v${v}_kernel_summary.txt; grep "config 3" gpurun_out/${TAG}_cfg3_v$v.log; head -8 gpurun_out/${TAG}_cfg3_v${v}_kernel_summary.txt
done
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/p2.log 2>&1)
python tools/prof_summary.py /tmp/p2/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg2_kernel_summary.txt; head -20 gpurun_out/${TAG}_cfg2_kernel_summary.txt
