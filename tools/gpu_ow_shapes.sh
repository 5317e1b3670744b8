#!/bin/bash
# per-GEMM-shape in-step timings under two TFX_NT_OW modes, one box: tools/gpu_ow_shapes.sh <tag> <mode>...
TAG=$1; shift
mkdir -p gpurun_out
for m in "$@"; do
  env TFX_BENCH_SHAPES=1 $m python bench.py --steps 6 --warmup 3 --family-steps 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/sh.log 2> /tmp/sh.err
  grep "^\[shape\]" /tmp/sh.err > "gpurun_out/${TAG}_mode_${m// /_}.txt"
  python -c "import json;d=json.loads(open('/tmp/sh.log').read().strip().splitlines()[-1]);print('mode $m', round(d['ms_per_step'],3),'ms/step', [ (f['kernel'][:12], round(f['ms_per_step'],2)) for f in d['roofline_by_family']])" | tee -a "gpurun_out/${TAG}_mode_${m// /_}.txt"
done
