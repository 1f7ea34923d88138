"""Lean probe for counter collection: only the HIP net, a handful of forwards."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphafive_amd.network import ResNet
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(os.environ.get("B", 4096)); N = int(os.environ.get("N", 3)); S = int(os.environ.get("S", 11))
net = ResNet(S, device="cuda", seed=1)
if S == 11: net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
x = (torch.rand((B, 3, S, S), device="cuda") < 0.2).float()
if os.environ.get("BRANCH") is not None:
    from alphafive_amd import net_hip as _nh
    _nh.tune(4, int(os.environ["BRANCH"]))          # value branch on the side stream (1) or serialised on the main stream (0)
pv = net.select_backend("hip")
if os.environ.get("ABLBITS"):
    from alphafive_amd import net_hip as _nh2
    _nh2.tune(7, int(os.environ["ABLBITS"]))        # af_conv_f16s ablation bits (1 = no LDS-DMA after the first slabs, 2 = no stores)
if os.environ.get("MODE"):
    from alphafive_amd import net_hip
    net_hip.tune(0, int(os.environ["MODE"]))
for _ in range(N): pv(x)
torch.cuda.synchronize()
print("done")
