#!/bin/bash
# block order A/B: 1 = (head, sample) fastest / tile rank slowest ; 2 = XCD-local (the tiles of a pair consecutive in one XCD's queue)
mkdir -p gpurun_out
O=gpurun_out/attn5.txt
: > $O
export ATTNP_REPS=100 ATTNP_BWD=1
for order in 1 2 0; do for asm in 0 1; do
  echo "== order $order asm $asm" >> $O
  TFX_ATTN_ORDER=$order TFX_ATTN_ASM=$asm timeout 200 tools/attn_probe run o${order}a${asm} n1000 bench cfg3 >> $O 2>&1
done; done
timeout 100 tools/attn_probe cmp o1a0 o2a1 >> $O 2>&1
cat $O
