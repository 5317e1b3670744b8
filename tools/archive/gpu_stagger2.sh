#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in 0 18000 0 18000; do
  rm -rf /tmp/prof_s
  TFX_PP_STAGGER=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/s.log 2>&1
  echo "== stagger $v"
  python $R/tools/prof_summary.py /tmp/prof_s/p_kernel_trace.csv --steps 6 | grep -v "at::" | head -22
done
