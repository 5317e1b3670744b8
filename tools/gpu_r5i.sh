#!/bin/bash
# round 5, call I: attention-backward preparation fused into the dQ kernel - attention kernel tests + end-to-end goldens, then a 2 x 2-round A/B of TFX_ATTN_PREP
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05i tests "attention or fused_qk_norm_rope or training_step_matches or canon512 or velocity or text_matches or modality_matches"
for r in 1 2; do AB_FAMILY_STEPS=3 TFX_AB="TFX_ATTN_PREP=0;TFX_ATTN_PREP=1" bash tools/gpu_run.sh r05i ab 2>&1 | cut -c1-330; done | tee gpurun_out/r05i_ab.txt
