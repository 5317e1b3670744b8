#!/bin/bash
# builds tools/attn_probe (stand-alone tfx_attn_fwd / tfx_attn_bwd probe, see tools/attn_probe.cpp)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/attn_probe.cpp -o tools/attn_probe -ldl
echo built tools/attn_probe
