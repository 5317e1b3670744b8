#!/bin/bash
# PMC counters of the ring-form pull kernel and layer_end_fwd at config 4 (dim 768 / depth 16, two modalities)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "pull_dma|layer_end" --output-format csv -d /tmp/pmc4_$i -o p -- python $R/tools/bench_configs.py 4 > /tmp/pmc4_$i.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc4_$i/p_counter_collection.csv --steps 1 | cut -c1-230
done
