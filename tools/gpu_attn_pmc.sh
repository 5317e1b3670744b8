#!/bin/bash
# L2 hit rate and memory-side fetch of the attention kernels per block order (rocprofv3 PMC on the stand-alone probe; counters only - no traces)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
O=$R/gpurun_out/attn_pmc.txt
: > $O
export TMPDIR=/tmp ATTNP_REPS=3 ATTNP_BWD=${ATTNP_BWD:-1}
cd /tmp
for order in ${ORDERS:-1 2}; do
  for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    tag=$(echo $C | tr ' ' '_')
    rm -rf /tmp/pmc_o
    TFX_ATTN_ORDER=$order TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_hip.so timeout 300 rocprofv3 --pmc $C --kernel-include-regex "attn_" --output-format csv -d /tmp/pmc_o -o p -- $R/tools/attn_probe run pmc bench > /tmp/pmc_o.log 2>&1
    echo "== order $order counters $C" >> $O
    python3 $R/tools/pmc_summary.py $(find /tmp/pmc_o -name "*counter_collection.csv" | head -1) --steps 1 >> $O 2>&1 || tail -5 /tmp/pmc_o.log >> $O
  done
done
cat $O
