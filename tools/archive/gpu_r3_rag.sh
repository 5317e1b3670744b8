#!/bin/bash
# round-3: kernel time of RAGGED steps (every batch a new structure) next to the cached step
TAG=${1:-r03rag}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prg && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prg -o p -- python $R/tools/prof_ragged_host.py 12 > /tmp/prg.log 2>&1)
python tools/prof_summary.py /tmp/prg/p_kernel_trace.csv --steady --by-grid gemm_nt_pp > gpurun_out/${TAG}_ragged_kernel_summary.txt; head -14 gpurun_out/${TAG}_ragged_kernel_summary.txt | cut -c1-130

