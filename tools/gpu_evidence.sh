#!/bin/bash
# round evidence in one GPU round trip (rounds 3 and 4): the bench line (with ragged steady state, sample_many, CPU baseline), one-stream steady-state kernel
# summaries + gaps of config 2 / 3 / 4, NT GEMM HBM traffic (PMC), SQ counters of the attention and pull-form kernels.
# usage (on the GPU box): bash tools/gpu_evidence.sh <tag>     -> gpurun_out/ev_<tag>_*
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/ev_${TAG}_bench_line.json 2> gpurun_out/ev_${TAG}_bench.err; cut -c1-300 gpurun_out/ev_${TAG}_bench_line.json
(cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_prof -o p -- python $R/bench.py --steps 4 --warmup 2 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/ev_prof.log 2>&1)
python tools/prof_summary.py /tmp/ev_prof/p_kernel_trace.csv --steady > gpurun_out/ev_${TAG}_cfg2_kernel_summary.txt
python tools/prof_gaps.py /tmp/ev_prof/p_kernel_trace.csv --steps 2 > gpurun_out/ev_${TAG}_cfg2_gaps.txt 2>&1
for C in 3 4; do
  (cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_p$C -o p -- python $R/tools/bench_configs.py $C > $R/gpurun_out/ev_${TAG}_cfg${C}.log 2>&1)
  python tools/prof_summary.py /tmp/ev_p$C/p_kernel_trace.csv --steady > gpurun_out/ev_${TAG}_cfg${C}_kernel_summary.txt
  python tools/prof_gaps.py /tmp/ev_p$C/p_kernel_trace.csv --steps 2 > gpurun_out/ev_${TAG}_cfg${C}_gaps.txt 2>&1
  grep "config $C" gpurun_out/ev_${TAG}_cfg${C}.log
done
bash tools/pmc_traffic.sh ev_$TAG gemm_nt > /dev/null 2>&1; cp gpurun_out/traffic_ev_$TAG.txt gpurun_out/ev_${TAG}_traffic_gemm_nt.txt
bash tools/pmc_attn.sh ev_$TAG "attn|pull|adaln" > gpurun_out/ev_${TAG}_pmc_sq_attn_tokenwise.txt 2>&1
bash tools/pmc_traffic.sh evtw_$TAG "pull|adaln_pre|layer_end" > /dev/null 2>&1; cp gpurun_out/traffic_evtw_$TAG.txt gpurun_out/ev_${TAG}_traffic_tokenwise.txt
head -40 gpurun_out/ev_${TAG}_cfg2_kernel_summary.txt; head -4 gpurun_out/ev_${TAG}_cfg2_gaps.txt; cat gpurun_out/ev_${TAG}_traffic_gemm_nt.txt; cat gpurun_out/ev_${TAG}_traffic_tokenwise.txt; cut -c1-220 gpurun_out/ev_${TAG}_pmc_sq_attn_tokenwise.txt
python -c "
import json; d=json.load(open('gpurun_out/ev_${TAG}_bench_line.json')); print({k: d.get(k) for k in ('value','ms_per_step','ragged_ms_per_step','structure_miss_ms','host_ms_per_step')}); print(d['roofline']['achieved'], d['roofline']['frac']); print(d.get('sample_many', {}).get('config5_forced_s'), d.get('sample_many', {}).get('config5_free_s')); print(d.get('cpu_baseline'))"
