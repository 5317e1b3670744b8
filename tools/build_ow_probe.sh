#!/bin/bash
# builds tools/ow_probe (stand-alone tfx_gemm_nt probe, see tools/ow_probe.cpp)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/ow_probe.cpp -o tools/ow_probe -ldl
echo built tools/ow_probe
