#!/bin/bash
# SQ counters of ONE NT shape on the hand-written kernel and on torch.matmul (hipBLASLt), same process: what the library's kernel does differently
# usage (on the GPU box): bash tools/pmc_gemm_vs_lib.sh <tag> [N K]      -> gpurun_out/<tag>_pmc_gemm_vs_lib.txt
# (SQ / GRBM counters only: a pass with TA_* / TCC_* / TCP_* counters on this workload did not return within 10 minutes on this pool)
TAG=${1:-pmc}; N=${2:-512}; K=${3:-2816}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
cat > /tmp/one_gemm.py <<PY
import sys, torch
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tools')
from transfusion_pytorch_amd import capi
from bench_gemm import st, dev, BF
M, N, K = 65536, $N, $K
A = torch.randn(M, K, device=dev).to(BF); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF); C = torch.empty(M, N, device=dev, dtype=BF)
a = capi.make_args('tfx_gemm_nt_args', A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_BF16'], C=C, ldc=N)
Bt = B.t()
for _ in range(10):
    capi.call('tfx_gemm_nt', a, st())
for _ in range(10):
    torch.matmul(A, Bt, out=C)
torch.cuda.synchronize()
PY
OUT=$R/gpurun_out/${TAG}_pmc_gemm_vs_lib.txt
echo "NT 65536 x $N x $K, 10 launches each: tfx_gemm_nt (gemm_nt_pp_kernel) vs torch.matmul" > $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_t -o p -- python /tmp/one_gemm.py > /tmp/pg_t.log 2>&1
grep -i "gemm\|Cijk\|Name" /tmp/pg_t/p_kernel_stats.csv | cut -c1-220 >> $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "gemm|Cijk" --output-format csv -d /tmp/pg_$i -o p -- python /tmp/one_gemm.py > /tmp/pg_$i.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pg_$i/p_counter_collection.csv --steps 10 | cut -c1-260 >> $OUT
done
cat $OUT
