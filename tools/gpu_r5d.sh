#!/bin/bash
# round 5, call D: env A/Bs on the current code - 512-column NT GEMMs on the 128 x 128 kernel (TFX_NT_PP_MIN=513), 512 x 512 weight gradients on 256 x 256 tiles
# (TFX_TN_TILE=2); kernel tests of what changed since call C (pull kernel with the bias partials)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05d tests "attnres or pull or training_step_matches or geglu"
AB_FAMILY_STEPS=3 TFX_AB="TFX_NT_PP_MIN=512;TFX_NT_PP_MIN=513;TFX_TN_TILE=2" bash tools/gpu_run.sh r05d ab 2>&1 | tee gpurun_out/r05d_ab.txt
