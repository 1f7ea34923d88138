"""Is a forward bound by its per-CU serial chain or by the board's power budget?  The same per-CU work twice: B positions on all 256
CUs, and B/2 positions on 128 CUs (AF_F16S_NCU=128: the persistent launches use half the workgroups, every workgroup walks the
same number of positions).  A chain-bound kernel takes the same time both ways; a power-bound one is faster on half the chip (the
idle half's share of the budget lets the busy half clock higher).  Run under tools/power_trace.py.  Env: S (11 / 15), B (4096), N."""
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from alphafive_amd.network import ResNet                               # noqa: E402

S, B, N = int(os.environ.get("S", 15)), int(os.environ.get("B", 4096)), int(os.environ.get("N", 1500))
ncu = int(os.environ.get("AF_F16S_NCU", 256))
Bx = B * ncu // 256
net = ResNet(S, device="cuda", seed=1)
if S == 11:
    net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
x = (torch.rand((Bx, 3, S, S), device="cuda") < 0.2).float()
pv = net.select_backend("hip")
for _ in range(100):
    pv(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    pv(x)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"S": S, "cus": ncu, "positions": Bx, "ms_per_forward": e0.elapsed_time(e1) / N}))
