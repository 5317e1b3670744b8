#!/bin/bash
# round 6: the software-pipelined attention backward against the plain-loop kernels: byte comparison + timings (stand-alone probe)
mkdir -p gpurun_out
O=gpurun_out/attn_bwd.txt
: > $O
export ATTNP_BWD=1 ATTNP_REPS=${ATTNP_REPS:-50}
TFX_ATTN_BWD_PIPE=2 timeout 60 tools/attn_probe run pipes n256 >> $O 2>&1; echo "small rc $?" >> $O
TFX_ATTN_BWD_PIPE=0 timeout 300 tools/attn_probe run plain >> $O 2>&1; echo "plain rc $?" >> $O
TFX_ATTN_BWD_PIPE=2 timeout 300 tools/attn_probe run pipe >> $O 2>&1; echo "pipe rc $?" >> $O
timeout 120 tools/attn_probe cmp plain pipe >> $O 2>&1
cat $O
