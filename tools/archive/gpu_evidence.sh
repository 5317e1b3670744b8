#!/bin/bash
# round evidence in one GPU round trip: steady-state kernel summary (one stream), inter-kernel gaps, NT GEMM HBM traffic (PMC), attention SQ counters,
# the bench line, the sampling line.   usage (on the GPU box): bash tools/gpu_evidence.sh <tag>     -> gpurun_out/ev_<tag>_*.txt
TAG=${1:-r}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_prof -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample > /tmp/ev_prof.log 2>&1)
python tools/prof_summary.py /tmp/ev_prof/p_kernel_trace.csv --steady > gpurun_out/ev_${TAG}_kernel_summary.txt
python tools/prof_gaps.py /tmp/ev_prof/p_kernel_trace.csv --steps 2 | grep -A3 "step -2" > gpurun_out/ev_${TAG}_gaps.txt
bash tools/pmc_traffic.sh ev_$TAG gemm_nt > /dev/null 2>&1; cp gpurun_out/traffic_ev_$TAG.txt gpurun_out/ev_${TAG}_traffic_gemm_nt.txt
bash tools/pmc_attn.sh ev_$TAG attn > gpurun_out/ev_${TAG}_pmc_sq_attn.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/ev_${TAG}_bench_line.json 2> /dev/null
python bench.py --sample > gpurun_out/ev_${TAG}_sample_line.json 2> /dev/null
head -30 gpurun_out/ev_${TAG}_kernel_summary.txt; cat gpurun_out/ev_${TAG}_gaps.txt; cat gpurun_out/ev_${TAG}_traffic_gemm_nt.txt; cat gpurun_out/ev_${TAG}_pmc_sq_attn.txt | cut -c1-220; cut -c1-400 gpurun_out/ev_${TAG}_bench_line.json; cut -c1-300 gpurun_out/ev_${TAG}_sample_line.json
