#!/bin/bash
# round-3 call J: two-stream concurrency probe
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TFX_SIDE_STREAM=0 python tools/bench_two_streams.py 10 2>&1 | grep -v amdgpu.ids
TFX_SIDE_STREAM=0 python tools/bench_two_streams.py 10 2>&1 | grep -v amdgpu.ids
