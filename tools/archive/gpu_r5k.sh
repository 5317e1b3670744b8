#!/bin/bash
# round 5, call K: trimmed arithmetic of the fused QK-norm / RoPE backward epilogue - its tests, then A/B of the previous build against the new one
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05k tests "fused_qk_norm_rope or training_step_matches"
for r in 1 2; do AB_FAMILY_STEPS=3 TFX_AB="TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_prev.so;TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_hip.so" bash tools/gpu_run.sh r05k ab 2>&1 | cut -c1-330; done | tee gpurun_out/r05k_ab.txt
