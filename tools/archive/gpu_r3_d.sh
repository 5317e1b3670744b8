#!/bin/bash
# round-3 call D: LDS-DMA ring pull kernel + swizzled attention-backward tiles: tests, same-box A/B, config 3 summary
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "pull or fused or attnres or adaln or layer_end or attention" > gpurun_out/${TAG}_pytest_k.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest_k.log | tail -3; grep -n "^FAILED\|Error\|assert" gpurun_out/${TAG}_pytest_k.log | head -10
python -m pytest tests/test_model_gpu.py -q -x > gpurun_out/${TAG}_pytest_m.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest_m.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest_m.log | head -10
for i in 1 2 3; do
  for v in "TFX_PULL_VARIANT=0" "TFX_PULL_VARIANT=1" "TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_attnold.so"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v'[:40], round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
export TMPDIR=/tmp
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/p2.log 2>&1)
python tools/prof_summary.py /tmp/p2/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg2_kernel_summary.txt; head -24 gpurun_out/${TAG}_cfg2_kernel_summary.txt
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o p -- python $R/tools/bench_configs.py 3 > $R/gpurun_out/${TAG}_cfg3.log 2>&1)
python tools/prof_summary.py /tmp/p3/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg3_kernel_summary.txt; grep "config 3" gpurun_out/${TAG}_cfg3.log; head -14 gpurun_out/${TAG}_cfg3_kernel_summary.txt
