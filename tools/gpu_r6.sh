#!/bin/bash
# round 6: the evidence set in one GPU round trip - GPU suite, tools/gpu_evidence.sh (bench line with riders; one-stream kernel summaries + gaps of configs 2 / 3 / 4;
# HBM traffic and SQ counter passes), the in-step GEMM shape table, the library yardstick, the attention probe (byte comparison of the generated loops + timings),
# per-kernel times of the attention kernels, the config-5 decode kernel table, the decode weight-prefetch A/B + cold / warm weights probe.        bash tools/gpu_r6.sh [tag]
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run.sh $TAG tests; cp gpurun_out/parity_measured.json gpurun_out/${TAG}_parity_measured.json
bash tools/gpu_evidence.sh $TAG > gpurun_out/${TAG}_evidence.log 2>&1; tail -30 gpurun_out/${TAG}_evidence.log | cut -c1-260
TFX_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > gpurun_out/${TAG}_shapes.json 2> gpurun_out/${TAG}_shapes.err
grep "\[shape\]" gpurun_out/${TAG}_shapes.err > gpurun_out/${TAG}_shapes.txt
python tools/bench_gemm_lib.py > gpurun_out/${TAG}_gemm_vs_lib.txt 2>&1; tail -12 gpurun_out/${TAG}_gemm_vs_lib.txt
ATTNP_BWD=1 TFX_ATTN_BWD_PIPE=0 TFX_ATTN_ASM=0 timeout 300 tools/attn_probe run plain > gpurun_out/${TAG}_attn_probe.txt 2>&1
ATTNP_BWD=1 TFX_ATTN_BWD_PIPE=1 TFX_ATTN_ASM=1 timeout 300 tools/attn_probe run asm >> gpurun_out/${TAG}_attn_probe.txt 2>&1
timeout 120 tools/attn_probe cmp plain asm >> gpurun_out/${TAG}_attn_probe.txt 2>&1; tail -3 gpurun_out/${TAG}_attn_probe.txt
bash tools/gpu_attn6.sh TFX_ATTN_BWD_PIPE=0 TFX_ATTN_BWD_PIPE=1 > gpurun_out/${TAG}_attn_kernel_times.txt 2>&1; cat gpurun_out/${TAG}_attn_kernel_times.txt
AB_FAMILY_STEPS=3 TFX_AB="TFX_ATTN_BWD_PIPE=0;TFX_ATTN_BWD_PIPE=1" bash tools/gpu_run.sh $TAG ab 2>&1 | tee gpurun_out/${TAG}_ab_dq.txt
bash tools/gpu_decode_prof.sh > /dev/null 2>&1; cp gpurun_out/decode_kernels_cfg5.txt gpurun_out/${TAG}_decode_kernels_cfg5.txt
{ for pf in 0 1; do echo "== TFX_DECODE_PREFETCH=$pf"; TFX_DECODE_PREFETCH=$pf timeout 300 python tools/bench_decode_step.py 2>&1 | tail -2; done
  echo "== cold / warm weights (tools/decode_cold_probe.py)"; timeout 300 python tools/decode_cold_probe.py 64,256 2>&1 | grep "^M="; } > gpurun_out/${TAG}_decode_prefetch.txt
cat gpurun_out/${TAG}_decode_prefetch.txt
ls gpurun_out | grep $TAG | head -60
