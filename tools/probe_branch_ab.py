"""Forward time at 4096 positions with the value branch on the side stream (production) and serialised on the main stream
(net_hip.tune(4, 0 / 1)): r3 final kernels 1.326 vs 1.331 ms."""
import os, sys, torch, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from alphafive_amd.network import ResNet
from alphafive_amd import net_hip
net = ResNet(11, device="cuda", seed=1); net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
x = (torch.rand((4096, 3, 11, 11), device="cuda") < 0.2).float()
pv = net.select_backend("hip")
for br in (1, 0, 1, 0):
    net_hip.tune(4, br)
    for _ in range(20): pv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): pv(x)
    e1.record(); torch.cuda.synchronize()
    print("branch on side stream" if br else "branch serialised   ", e0.elapsed_time(e1) / 200)
