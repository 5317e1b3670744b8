#!/bin/bash
# round 5, second session: final evidence in one GPU round trip.  Same-box A/B of the one-wave-per-SIMD GEMM kernels against the round's earlier kernels
# (TFX_NT_OW=0 TFX_TN_OW=0: ping-pong NT + 8-wave TN), the evidence set of tools/gpu_evidence.sh (bench line with riders, one-stream kernel summaries of
# configs 2 / 3 / 4, HBM traffic and SQ counter passes), the in-step GEMM shape table, the library yardstick, the stand-alone probe (burst + steady state,
# bit-identity against the ping-pong kernel), SQ counters of the new NT kernel, the GPU suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh r05f tests; cp gpurun_out/parity_measured.json gpurun_out/r05f_parity_measured.json
AB_FAMILY_STEPS=3 TFX_AB="TFX_NT_OW=0 TFX_TN_OW=0;TFX_NT_OW=1 TFX_TN_OW=1" bash tools/gpu_run.sh r05f ab 2>&1 | tee gpurun_out/r05f_ab_ow.txt
bash tools/gpu_ow_cfg.sh r05f_cfg_ab > /dev/null 2>&1
bash tools/gpu_evidence.sh r05f > gpurun_out/r05f_evidence.log 2>&1; tail -40 gpurun_out/r05f_evidence.log | cut -c1-260
TFX_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > gpurun_out/r05f_shapes.json 2> gpurun_out/r05f_shapes.err
grep "\[shape\]" gpurun_out/r05f_shapes.err > gpurun_out/r05f_shapes.txt
python tools/bench_gemm_lib.py > gpurun_out/r05f_gemm_vs_lib.txt 2>&1; tail -30 gpurun_out/r05f_gemm_vs_lib.txt
# stand-alone probe: every case on both kernel families, short bursts, bit comparison; then the steady state (600 launches per case)
export TFX_NT_PP_MIN=1
TFX_NT_OW=0 timeout 300 tools/ow_probe run pp > gpurun_out/r05f_probe_pp.txt 2>&1
TFX_NT_OW=2 timeout 300 tools/ow_probe run ow > gpurun_out/r05f_probe_ow.txt 2>&1
timeout 300 tools/ow_probe cmp pp ow > gpurun_out/r05f_probe_cmp.txt 2>&1; tail -2 gpurun_out/r05f_probe_cmp.txt
: > gpurun_out/r05f_probe_steady.txt
tools/mfma_peak 2>/dev/null | grep -A2 "random" | head -3 >> gpurun_out/r05f_probe_steady.txt
for m in 0 1; do
  OWP_REPS=600 TFX_NT_OW=$m timeout 300 tools/ow_probe run s$m n512k512,n512k1408,n512k2816,n1544k512,n1024k1024,n1024k2752,n5504k1024,sq4096 2>&1 | sed "s/^/NT_OW=$m /" >> gpurun_out/r05f_probe_steady.txt
  OWP_REPS=400 TFX_TN_OW=$m timeout 300 tools/ow_probe tn t$m 2>&1 | sed "s/^/TN_OW=$m /" >> gpurun_out/r05f_probe_steady.txt
done
timeout 120 tools/ow_probe tncmp t0 t1 >> gpurun_out/r05f_probe_steady.txt 2>&1
unset TFX_NT_PP_MIN
bash tools/gpu_ow_pmc.sh r05f_pmc sq4096 base > /dev/null 2>&1
bash tools/gpu_ow_pmc.sh r05f_pmc sq4096 pp > /dev/null 2>&1
bash tools/gpu_ow_pmc.sh r05f_pmc n512k512 base > /dev/null 2>&1
bash tools/gpu_ow_pmc.sh r05f_pmc n512k512 pp > /dev/null 2>&1
ls gpurun_out | grep r05f | head -60
