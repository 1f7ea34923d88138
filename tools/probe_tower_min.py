"""Lean probe for counter collection: a few launches of the bf16 tower kernels only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphafive_amd.network_deep import DeepResNet
B = int(os.environ.get("B", 8192))
net = DeepResNet(11, blocks=int(os.environ.get("BLOCKS", 2)), width=128, device="cuda")
net.select_backend("hip", B)
if os.environ.get("ABL"):
    from alphafive_amd import tower_hip
    tower_hip.tune(2, int(os.environ["ABL"]))          # profiling ablations of af_tower_conv (results wrong by design)
if os.environ.get("ENGINE"):
    from alphafive_amd import tower_hip
    tower_hip.tune(3, int(os.environ["ENGINE"]))       # 0 af_tower_conv, 2 af_tower_conv3, 3 (default) conv3 + conv
h0 = torch.randn((B, 128, 11, 11), device="cuda").bfloat16()
net._tower.load_nchw(h0)
for _ in range(int(os.environ.get("N", 2))): net._tower.forward(B)
torch.cuda.synchronize()
print("done")
