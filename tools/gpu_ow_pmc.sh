#!/bin/bash
# SQ counters of the one-wave-per-SIMD NT kernel on the stand-alone probe: tools/gpu_ow_pmc.sh <tag> <case> <variant (base|pp|name)>
TAG=$1; CASE=$2; V=$3
export TMPDIR=/tmp TFX_NT_PP_MIN=1
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ $V = pp ]; then export TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_hip.so TFX_NT_OW=0; elif [ $V = base ]; then export TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_hip.so TFX_NT_OW=1; else export TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$V.so TFX_NT_OW=1; fi
OUT=$R/gpurun_out/${TAG}_${V}_${CASE}.txt
: > $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/pg_$i
  timeout 120 rocprofv3 --pmc $C --kernel-include-regex "gemm" --output-format csv -d /tmp/pg_$i -o p -- $R/tools/ow_probe run pmc $CASE > /tmp/pg_$i.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pg_$i/p_counter_collection.csv --steps 24 | cut -c1-260 >> $OUT
done
cat $OUT
