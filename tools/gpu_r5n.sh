#!/bin/bash
# round 5, call N: balanced fragment reads in the ping-pong NT kernel (TFX_PP_BAL) - GEMM kernel tests + goldens, then A/B of the two builds with the family split,
# then the library yardstick on the new build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05n tests "gemm_nt or geglu or training_step_matches or canon512"
for r in 1 2; do AB_FAMILY_STEPS=3 TFX_AB="TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_bal0.so;TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_hip.so" bash tools/gpu_run.sh r05n ab 2>&1 | cut -c1-330; done | tee gpurun_out/r05n_ab.txt
python tools/bench_gemm_lib.py 2>/dev/null | grep "^NT" | tee gpurun_out/r05n_gemm_vs_lib.txt
