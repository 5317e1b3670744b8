#!/bin/bash
# round-3 call W: TN split policy: in-step duration per TN launch (one stream) for a sweep of the fill target
TAG=${1:-r03w}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "TFX_TN_FILL=0.6 TFX_TN_FILL0=0.4" "TFX_TN_FILL=0.7 TFX_TN_FILL0=0.5" "TFX_TN_FILL=0.8 TFX_TN_FILL0=0.6" "TFX_TN_FILL=0.9 TFX_TN_FILL0=0.75" "TFX_TN_FILL=0.97 TFX_TN_FILL0=1.0"; do
(cd /tmp && rm -rf /tmp/pt && env $v TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/pt.log 2>&1)
echo "$v"; python tools/prof_summary.py /tmp/pt/p_kernel_trace.csv --steady --by-grid gemm_tn | grep "total\|gemm_tn" | cut -c1-150
done | tee gpurun_out/${TAG}_tn_splits.txt
