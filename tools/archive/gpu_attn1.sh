#!/bin/bash
# round 6: bit-identity + timing of the generated attention main loops against the hipcc-scheduled kernels (stand-alone probe, no torch)
mkdir -p gpurun_out
O=gpurun_out/attn1.txt
: > $O
TFX_ATTN_ASM=1 timeout 60 tools/attn_probe run asms n256 >> $O 2>&1; echo "small rc $?" >> $O
TFX_ATTN_ASM=0 ATTNP_BWD=${ATTNP_BWD:-0} timeout 300 tools/attn_probe run ref >> $O 2>&1; echo "ref rc $?" >> $O
TFX_ATTN_ASM=1 ATTNP_BWD=${ATTNP_BWD:-0} timeout 300 tools/attn_probe run asm >> $O 2>&1; echo "asm rc $?" >> $O
timeout 120 tools/attn_probe cmp ref asm >> $O 2>&1
cat $O
