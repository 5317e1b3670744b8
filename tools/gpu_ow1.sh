#!/bin/bash
# round 5, session 2: first GPU trip of the one-wave-per-SIMD NT kernel: bit-identity against the ping-pong kernel + timings (stand-alone probe, no torch)
mkdir -p gpurun_out
export TFX_NT_PP_MIN=1
TFX_NT_OW=2 timeout 60 tools/ow_probe run ows k64 > gpurun_out/ow1_small.txt 2>&1
echo "small rc $?" >> gpurun_out/ow1_small.txt
TFX_NT_OW=0 timeout 300 tools/ow_probe run pp > gpurun_out/ow1_pp.txt 2>&1
echo "pp rc $?" >> gpurun_out/ow1_pp.txt
TFX_NT_OW=2 timeout 300 tools/ow_probe run ow > gpurun_out/ow1_ow.txt 2>&1
echo "ow rc $?" >> gpurun_out/ow1_ow.txt
timeout 300 tools/ow_probe cmp pp ow > gpurun_out/ow1_cmp.txt 2>&1
cat gpurun_out/ow1_small.txt gpurun_out/ow1_pp.txt gpurun_out/ow1_ow.txt gpurun_out/ow1_cmp.txt
