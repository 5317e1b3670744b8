#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/prof_s
  TFX_ATTN_ORDER=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/s.log 2>&1
  echo "== TFX_ATTN_ORDER=$v"
  python $R/tools/prof_summary.py /tmp/prof_s/p_kernel_trace.csv --steps 6 | grep "attn_\|total"
done
