#!/bin/bash
# same-box A/B of the attention kernels across library builds: bash tools/gpu_attn_libs.sh <lib name> ...   (names as in lib/libtfx_<name>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rnd in 1 2; do for L in "$@"; do TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$L.so python tools/bench_attn.py 20 2>&1 | grep -v amdgpu.ids; done; done
