#!/bin/bash
# power sensor + shader clock under: the split-operand forward, the bf16 tower, a dense hipBLASLt bf16 GEMM, an HBM copy, the bench
# usage (GPU box): bash tools/power_workloads.sh gpurun_out/<dir>
set -u
OUT=${1:-gpurun_out/power}; mkdir -p $OUT
PT="python tools/power_trace.py"
N=6000 $PT $OUT/forward_11x11.csv -- python tools/probe_net_min.py | tail -1 | tee $OUT/forward_11x11.json
N=1200 S=15 $PT $OUT/forward_15x15.csv -- python tools/probe_net_min.py | tail -1 | tee $OUT/forward_15x15.json
N=2000 BLOCKS=8 $PT $OUT/tower_bf16.csv -- python tools/probe_tower_min.py | tail -1 | tee $OUT/tower_bf16.json
$PT $OUT/gemm_bf16.csv -- python -c "
import torch
a=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16); b=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)
for _ in range(6000): c=a@b
torch.cuda.synchronize()" | tail -1 | tee $OUT/gemm_bf16.json
$PT $OUT/hbm_copy.csv -- python -c "
import torch
a=torch.empty(1<<30,device='cuda',dtype=torch.uint8); b=torch.empty_like(a)
for _ in range(20000): b.copy_(a)
torch.cuda.synchronize()" | tail -1 | tee $OUT/hbm_copy.json
$PT $OUT/bench.csv -- python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-pmc | tail -1 | tee $OUT/bench.json
