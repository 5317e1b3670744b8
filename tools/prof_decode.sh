#!/bin/bash
# rocprofv3 kernel trace of the KV-cached text decode (tools/bench_sample.py): kernel time per decode step vs wall time
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -o p -- python $R/tools/bench_sample.py --new 64 > $R/gpurun_out/prof_dec.log 2>&1
cd $R && tail -2 gpurun_out/prof_dec.log
python tools/prof_summary.py gpurun_out/prof_dec/p_kernel_trace.csv --steps 1 | head -40
