#!/bin/bash
# timing variants of the forward asm loop (wrong results by construction): where does a tile's time go
mkdir -p gpurun_out
O=gpurun_out/attn4.txt
: > $O
export ATTNP_REPS=200
for lib in hip NOLDS NOLDSDMA NOVALU NOMFMA VALUONLY; do
  echo "== lib $lib asm" >> $O
  TFX_LIB=transfusion_pytorch_amd/lib/libtfx_$lib.so TFX_ATTN_ASM=1 timeout 120 tools/attn_probe run x bench >> $O 2>&1
done
echo "== ref" >> $O
TFX_ATTN_ASM=0 timeout 120 tools/attn_probe run x bench >> $O 2>&1
cat $O
