#!/bin/bash
# HBM traffic of one kernel family via rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md:
# the two do not fit one pass; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 -> corrected in bench.py).
# usage (on the GPU box): bash tools/pmc_traffic.sh <tag> [kernel regex]      -> gpurun_out/traffic_<tag>.txt
TAG=${1:-t}; RX=${2:-gemm_nt}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
: > $R/gpurun_out/traffic_$TAG.txt
for C in FETCH_SIZE WRITE_SIZE; do
  TFX_SIDE_STREAM=0 rocprofv3 --pmc $C --kernel-include-regex "$RX" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -o p -- python $R/bench.py --steps 1 --warmup 1 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$C/p_counter_collection.csv --steps 2 | tee -a $R/gpurun_out/traffic_$TAG.txt
  rm -f $R/gpurun_out/pmc_${TAG}_$C/p_counter_collection.csv
done
