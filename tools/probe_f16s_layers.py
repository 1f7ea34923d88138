"""Per-layer times of the fp16 split-operand path: run under rocprofv3 --kernel-trace (value branch on the main stream
so that kernels do not overlap), then tools/rocpd_stats.py prints the per-kernel averages."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from alphafive_amd import net_hip
from alphafive_amd.network import ResNet
from test_gpu_net import _positions

net = ResNet(11, device="cuda")
net.load_npz(os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz"))
B = int(os.environ.get("B", 4096))
hn = net_hip.HipNet(net.variables, 11, B, "cuda")
xb = torch.from_numpy(_positions(11, B, seed=1)).cuda()
net_hip.tune(0, 5)
net_hip.tune(4, int(os.environ.get("BRANCH", 0)))
net_hip.tune(7, int(os.environ.get("ABLBITS", 0)))
for _ in range(int(os.environ.get("N", 30))):
    hn(xb)
torch.cuda.synchronize()
