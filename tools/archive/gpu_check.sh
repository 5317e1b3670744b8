#!/bin/bash
# one GPU round trip: all GPU tests, a rocprofv3 kernel summary of 4 bench steps (ONE stream: TFX_SIDE_STREAM=0, so every kernel's
# duration is its own - the same condition bench.py brackets the roofline kernels under), and a 10-step bench line (two streams)
# usage (on the GPU box): bash tools/gpu_check.sh <tag>
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
export TMPDIR=/tmp
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --ragged-steps 0 --no-sample > $R/gpurun_out/prof_$TAG.log 2>&1)
python tools/prof_summary.py gpurun_out/prof_$TAG/p_kernel_trace.csv --steps 4 | head -${2:-26}
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','host_ms_per_step')}, d['roofline']['achieved'])"
