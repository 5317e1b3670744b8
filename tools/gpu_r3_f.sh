#!/bin/bash
# round-3 call F: all GPU tests; TN fragment pipelining A/B; one-stream kernel summaries
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest.log | head -20
for i in 1 2 3; do
  for v in "TFX_TN_PIPE=1" "TFX_TN_PIPE=0"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v'[:24], round(d['ms_per_step'],2), 'loss', d['loss'])"
  done
done
export TMPDIR=/tmp
for v in 1 0; do
(cd /tmp && TFX_TN_PIPE=$v TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2$v -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/p2.log 2>&1)
python tools/prof_summary.py /tmp/p2$v/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg2_pipe${v}_kernel_summary.txt; grep "total\|gemm_tn" gpurun_out/${TAG}_cfg2_pipe${v}_kernel_summary.txt
done
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o p -- python $R/tools/bench_configs.py 3 > $R/gpurun_out/${TAG}_cfg3.log 2>&1)
python tools/prof_summary.py /tmp/p3/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg3_kernel_summary.txt; grep "config 3" gpurun_out/${TAG}_cfg3.log; head -8 gpurun_out/${TAG}_cfg3_kernel_summary.txt
