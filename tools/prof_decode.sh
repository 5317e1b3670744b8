#!/bin/bash
# rocprofv3 kernel trace of the decode twin: kernel time vs wall time.   bash tools/prof_decode.sh [sample|config5]
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
CMD="python $R/tools/bench_sample.py --new 64"
[ "$1" == "config5" ] && CMD="python $R/tools/bench_configs.py 5"
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dec -o p -- $CMD > $R/gpurun_out/prof_dec.log 2>&1
cd $R && grep -v "rocprofv3\|amdgpu" gpurun_out/prof_dec.log | tail -3
python tools/prof_summary.py /tmp/prof_dec/p_kernel_trace.csv --steps 1 | head -${2:-30}
