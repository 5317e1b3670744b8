#!/bin/bash
# round 5, call H: fused QK-norm / RoPE backward - the fixed kernel test, then a 3-round A/B of the switch (0 = two launches, 1 = fused with atomics)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05h tests "fused_qk_norm_rope"
for r in 1 2; do AB_FAMILY_STEPS=0 TFX_AB="TFX_ATTN_QKNR=0;TFX_ATTN_QKNR=1" bash tools/gpu_run.sh r05h ab 2>&1 | cut -c1-170; done | tee gpurun_out/r05h_ab.txt
