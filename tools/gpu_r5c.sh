#!/bin/bash
# round 5, call C: the whole GPU suite on the current code, env A/Bs (GELU grid vs polynomial in the GEGLU forward; de-phasing delay), the in-step GEMM shape table
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05c tests
TFX_AB="TFX_GELU_TABLE=1;TFX_GELU_TABLE=0;TFX_PP_STAGGER=8000;TFX_PP_STAGGER=18000;TFX_PP_STAGGER=24000" bash tools/gpu_run.sh r05c ab 2>&1 | tee gpurun_out/r05c_ab.txt
TFX_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > gpurun_out/r05c_shapes.json 2> gpurun_out/r05c_shapes.txt
grep "\[shape\]" gpurun_out/r05c_shapes.txt | head -30
