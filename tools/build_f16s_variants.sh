#!/bin/bash
# Build A/B variants of libaf_net.so (af_conv_f16s build-time switches) into alphafive_amd/_lib/variants/ for tools/probe_f16s_ab.py.
#   tools/build_f16s_variants.sh name1:"-DAF_F16S_STW=0" name2:"-DAF_F16S_STW=1 -DAF_F16S_NT_STORE=1" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p alphafive_amd/_lib/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I include -I alphafive_amd/csrc $flags \
      -o alphafive_amd/_lib/variants/libaf_net_$name.so alphafive_amd/csrc/af_net.hip alphafive_amd/csrc/af_conv_f16s.hip 2>&1 | grep -v "warning\|^$" || true ) &
done
wait
ls -la alphafive_amd/_lib/variants/
