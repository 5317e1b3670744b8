#!/bin/bash
# PMC counter passes restricted to a kernel-name regex (default: attention kernels); no trace domains besides kernel-trace
# usage (on the GPU box): bash tools/pmc_attn.sh <tag> [regex]
TAG=${1:-pmc}; RX=${2:-attn}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "$RX" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python $R/bench.py --steps 1 --warmup 1 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$i/p_counter_collection.csv --steps 2
  rm -f $R/gpurun_out/pmc_${TAG}_$i/p_counter_collection.csv
done
