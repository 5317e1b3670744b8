#!/bin/bash
# round-3 call A: GPU tests, bench line, one-stream kernel summaries of config 2 / 3 / 4 (rocprofv3 --kernel-trace)
# usage (on the GPU box): bash tools/gpu_r3_a.sh <tag>
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ragged-steps 0 --no-sample > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; cut -c1-330 gpurun_out/${TAG}_bench_line.json
export TMPDIR=/tmp
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/p2.log 2>&1)
python tools/prof_summary.py /tmp/p2/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg2_kernel_summary.txt; head -34 gpurun_out/${TAG}_cfg2_kernel_summary.txt
for C in 3 4; do
  (cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p$C -o p -- python $R/tools/bench_configs.py $C > $R/gpurun_out/${TAG}_cfg${C}.log 2>&1)
  python tools/prof_summary.py /tmp/p$C/p_kernel_trace.csv --steady > gpurun_out/${TAG}_cfg${C}_kernel_summary.txt
  python tools/prof_gaps.py /tmp/p$C/p_kernel_trace.csv --steps 2 > gpurun_out/${TAG}_cfg${C}_gaps.txt 2>&1
  tail -2 gpurun_out/${TAG}_cfg${C}.log; head -30 gpurun_out/${TAG}_cfg${C}_kernel_summary.txt; head -3 gpurun_out/${TAG}_cfg${C}_gaps.txt
done
