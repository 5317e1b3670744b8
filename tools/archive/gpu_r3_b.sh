#!/bin/bash
# round-3 call B: the pull-form AttentionResidual backward and the fused backward wrapper sides - kernel tests, model parity, same-box A/B
# of the environment switches, one-stream kernel summary.   usage (on the GPU box): bash tools/gpu_r3_b.sh <tag>
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "pull or fused or attnres or adaln or layer_end" > gpurun_out/${TAG}_pytest_k.log 2>&1; tail -15 gpurun_out/${TAG}_pytest_k.log
python -m pytest tests/test_model_gpu.py -q -x > gpurun_out/${TAG}_pytest_m.log 2>&1; tail -8 gpurun_out/${TAG}_pytest_m.log
for i in 1 2; do
  for v in "TFX_ATTNRES_PULL=1 TFX_BWD_FUSED=1" "TFX_ATTNRES_PULL=0 TFX_BWD_FUSED=0" "TFX_ATTNRES_PULL=1 TFX_BWD_FUSED=0" "TFX_ATTNRES_PULL=0 TFX_BWD_FUSED=1"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
export TMPDIR=/tmp
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/p2.log 2>&1)
python tools/prof_summary.py /tmp/p2/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg2_kernel_summary.txt; head -34 gpurun_out/${TAG}_cfg2_kernel_summary.txt
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o p -- python $R/tools/bench_configs.py 3 > $R/gpurun_out/${TAG}_cfg3.log 2>&1)
python tools/prof_summary.py /tmp/p3/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg3_kernel_summary.txt; grep "config 3" gpurun_out/${TAG}_cfg3.log; head -16 gpurun_out/${TAG}_cfg3_kernel_summary.txt
