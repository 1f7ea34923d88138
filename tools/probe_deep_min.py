"""Lean probe for a kernel table of the whole config-5 forward (stem + 16 tower convolutions + heads + dense): a few forwards of the
8-block bf16 net at B positions.  usage: rocprofv3 --kernel-trace --stats -d out -o p -- python tools/probe_deep_min.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphafive_amd.network_deep import DeepResNet
B = int(os.environ.get("B", 8192))
net = DeepResNet(11, blocks=8, width=128, device="cuda")
pv = net.select_backend("hip", B)
x = (torch.rand((B, 3, 11, 11), device="cuda") < 0.2).float()
for _ in range(int(os.environ.get("N", 8))):
    pv(x)
torch.cuda.synchronize()
print("done")
