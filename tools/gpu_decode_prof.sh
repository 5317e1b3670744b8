#!/bin/bash
# round 6: kernel table of one config-5 sample_many call (rocprofv3 kernel trace) + the GPU test of the decode keep with raw-kernel weight writes
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_dec && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dec -o p -- python $R/tools/bench_configs.py 5 > $R/gpurun_out/prof_dec.log 2>&1
cd $R && grep -v "rocprofv3\|amdgpu" gpurun_out/prof_dec.log | tail -4 > gpurun_out/decode_kernels_cfg5.txt
python tools/prof_summary.py $(find /tmp/prof_dec -name "*kernel_trace.csv" | head -1) --steps 1 | head -40 >> gpurun_out/decode_kernels_cfg5.txt
cat gpurun_out/decode_kernels_cfg5.txt
python -m pytest tests/test_sampling_gpu.py -x -q -m gpu -k "keeps_its_decode_plans" 2>&1 | tail -5
