#!/bin/bash
# round-3 call H: tests, layer-wise error of the depth-24 forward, host profile of the ragged steady state
TAG=${1:-r03h}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest.log | head -30
python tools/layer_error.py cfg3_1024 > gpurun_out/${TAG}_layer_error.txt 2>&1; tail -32 gpurun_out/${TAG}_layer_error.txt
python tools/prof_ragged_host.py 8 > gpurun_out/${TAG}_ragged_host.txt 2>&1; head -50 gpurun_out/${TAG}_ragged_host.txt | cut -c1-180
