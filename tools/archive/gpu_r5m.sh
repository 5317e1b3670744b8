#!/bin/bash
# round 5, call M: printed errors of the fused-backward kernel test and of the side-stream noise floor; env A/Bs of old switches on the final code
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -s -k "attention_bwd_with_fused_qk_norm_rope_bwd or fused_qk_norm_rope_backward_equals" > gpurun_out/r05m_pytest.log 2>&1
grep -n "fused\|passed\|failed" gpurun_out/r05m_pytest.log | cut -c1-200 | head -60
AB_FAMILY_STEPS=0 TFX_AB="TFX_QKNR=0;TFX_QKNR=1;TFX_PULL_VARIANT=2;TFX_PULL_VARIANT=5" bash tools/gpu_run.sh r05m ab 2>&1 | cut -c1-200 | tee gpurun_out/r05m_ab.txt
