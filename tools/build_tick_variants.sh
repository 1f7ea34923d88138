#!/bin/bash
# build variants of libaf_hip.so for A/B runs (tools/probe_tick_ab.py): usage tools/build_tick_variants.sh name "-DFLAG=.." [name flags ...]
cd "$(dirname "$0")/.." && mkdir -p alphafive_amd/_lib/variants
while [ $# -ge 2 ]; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I include -I alphafive_amd/csrc -ffp-contract=off $2 \
        -o alphafive_amd/_lib/variants/libaf_hip_$1.so alphafive_amd/csrc/af_engine.hip &
  shift 2
done
wait
ls -la alphafive_amd/_lib/variants/libaf_hip_*.so
