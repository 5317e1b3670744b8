#!/bin/bash
# A/B two builds of libtfx_hip.so inside ONE gpurun call (box-to-box variance is +-8 %): per-kernel ms/step for each.
# usage (on the GPU box): bash tools/ab.sh <libA.so> <libB.so> [kernel regex] [rounds]
A=$1; B=$2; RX=${3:-.}; N=${4:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for i in $(seq $N); do
  for L in $A $B; do
    rm -rf /tmp/prof_ab
    TFX_LIB=$R/$L rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ab -o p -- python $R/bench.py --steps 4 --warmup 2 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/ab.log 2>&1
    echo "== $L (round $i): $(python -c "import json;print(round(json.loads(open('/tmp/ab.log').read().strip().splitlines()[-1])['ms_per_step'],2))" 2>/dev/null) ms/step"
    python $R/tools/prof_summary.py /tmp/prof_ab/p_kernel_trace.csv --steps 6 | grep -E "$RX" | head -12
  done
done
