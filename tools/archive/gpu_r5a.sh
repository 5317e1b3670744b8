#!/bin/bash
# round 5, call A: GPU tests, quick bench, same-box A/B of the round-4 library against the new one (per-kernel ms/step)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05a tests
bash tools/gpu_run.sh r05a quick
bash tools/ab.sh transfusion_pytorch_amd/lib/libtfx_r04.so transfusion_pytorch_amd/lib/libtfx_hip.so "gemm_nt_pp|pull|tn_wide|attn_" 1 2>&1 | tee gpurun_out/r05a_ab.txt
