"""What would the forward gain if its HBM traffic disappeared?  The forward at the power cap, with af_conv_f16s's profiling ablations
(results wrong by design): ABLBITS 2 = no activation stores, 1 = no slab loads after a workgroup's first position (operands stay
in LDS), 3 = both, 4 = loads from L2-hot addresses.  One NORMAL forward runs first, so every activation buffer holds real values:
the ablated passes then multiply the same kind of numbers (operand bit patterns change the board's power draw: a probe that feeds
zeros or garbage measures the clock, not the traffic).  Run under tools/power_trace.py.  Env: ABLBITS, N (4000), B (4096)."""
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from alphafive_amd import net_hip                                     # noqa: E402
from alphafive_amd.network import ResNet                               # noqa: E402

B, N, bits = int(os.environ.get("B", 4096)), int(os.environ.get("N", 4000)), int(os.environ.get("ABLBITS", 0))
net = ResNet(11, device="cuda", seed=1)
net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
x = (torch.rand((B, 3, 11, 11), device="cuda") < 0.2).float()
pv = net.select_backend("hip")
p0, v0 = pv(x)
p0, v0 = p0.clone(), v0.clone()
net_hip.tune(7, bits)
for _ in range(200):
    pv(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    p, v = pv(x)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"ablbits": bits, "ms_per_forward": e0.elapsed_time(e1) / N, "positions": B, "forwards": N,
                  "max_abs_dv_vs_normal": float((v - v0).abs().max())}))
