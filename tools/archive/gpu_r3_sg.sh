#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "TFX_PULL_GRID=0" "TFX_PULL_GRID=256" "TFX_PULL_GRID=512" "TFX_PULL_GRID=1024"; do
(cd /tmp && rm -rf /tmp/pt && env $v TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
echo "$v"; python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady --by-grid "pull" | grep "total\|pull_reg" | cut -c1-130
done
