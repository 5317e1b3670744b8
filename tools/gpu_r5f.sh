#!/bin/bash
# round 5, call F: the QK-norm / RoPE backward fused into the attention-backward epilogues - kernel + end-to-end tests, then same-box A/B of the switch
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05f tests "fused_qk_norm_rope or attention or training_step_matches or canon512 or side_stream"
AB_FAMILY_STEPS=3 TFX_AB="TFX_ATTN_QKNR=0;TFX_ATTN_QKNR=1" bash tools/gpu_run.sh r05f ab 2>&1 | tee gpurun_out/r05f_ab.txt
