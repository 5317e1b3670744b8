#!/bin/bash
# round 5, final evidence (one GPU round trip): same-box A/B of the round-4 library against the final one (step time + every family), the round's evidence
# set (tools/gpu_evidence.sh: bench line with riders, one-stream kernel summaries of configs 2 / 3 / 4, HBM traffic and SQ counter passes), the in-step GEMM
# shape table, the library yardstick, the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05e tests; cp gpurun_out/parity_measured.json gpurun_out/r05e_parity_measured.json
AB_FAMILY_STEPS=3 TFX_AB="TFX_ATTN_QKNR=0;TFX_ATTN_QKNR=1" bash tools/gpu_run.sh r05e ab 2>&1 | tee gpurun_out/r05e_ab_qknr.txt
bash tools/gpu_evidence.sh r05e > gpurun_out/r05e_evidence.log 2>&1; tail -40 gpurun_out/r05e_evidence.log | cut -c1-260
TFX_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > gpurun_out/r05e_shapes.json 2> gpurun_out/r05e_shapes.err
grep "\[shape\]" gpurun_out/r05e_shapes.err > gpurun_out/r05e_shapes.txt
python tools/bench_gemm_lib.py > gpurun_out/r05e_gemm_vs_lib.txt 2>&1; tail -30 gpurun_out/r05e_gemm_vs_lib.txt
