#!/bin/bash
# the config-2 part of tools/gpu_evidence_r3.sh: bench line + one-stream steady-state kernel summary and gaps
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/ev_${TAG}_bench_line.json 2> gpurun_out/ev_${TAG}_bench.err
(cd /tmp && rm -rf /tmp/ev_prof && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_prof -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/ev_prof.log 2>&1)
python tools/prof_summary.py /tmp/ev_prof/p_kernel_trace.csv --steady > gpurun_out/ev_${TAG}_cfg2_kernel_summary.txt
python tools/prof_gaps.py /tmp/ev_prof/p_kernel_trace.csv --steps 2 > gpurun_out/ev_${TAG}_cfg2_gaps.txt 2>&1
python -c "
import json; d=json.load(open('gpurun_out/ev_${TAG}_bench_line.json')); print({k: d.get(k) for k in ('value','ms_per_step','ragged_ms_per_step','structure_miss_ms')}, d['roofline']['achieved'])"
head -3 gpurun_out/ev_${TAG}_cfg2_kernel_summary.txt
