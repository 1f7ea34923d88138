"""Does the ORDER in which a layer walks the batch matter?  (r6)  Every activation tensor of the 11x11 forward at 4096 positions is
75-302 MB, the Infinity Cache 256 MB: a consumer that walks the batch in its producer's order starts on the data that left the
cache first.  af_net_tune(7, 0x2000) reverses the walk of the even layers (boustrophedon: a launch starts on what its producer
wrote last); 0x6000 reverses every layer (control: today's order, mirrored).  Same bits per position in every variant — checked.
Variants interleaved in one process, R rounds of N forwards each.  Env: B (4096), N (600), R (8), S (11)."""
import json
import os
import sys

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
from alphafive_amd import net_hip                                     # noqa: E402
from alphafive_amd.network import ResNet                               # noqa: E402

B, N, R = int(os.environ.get("B", 4096)), int(os.environ.get("N", 600)), int(os.environ.get("R", 8))
variants = [int(v, 0) for v in os.environ.get("VARIANTS", "0,0x2000,0x6000").split(",")]
net = ResNet(11, device="cuda", seed=1)
net.load_npz(os.path.join(R_, "tests/golden/alphaFive-6960.weights.npz"))
x = (torch.rand((B, 3, 11, 11), device="cuda") < 0.2).float()
pv = net.select_backend("hip")
p0, v0 = pv(x)
p0, v0 = p0.clone(), v0.clone()
for _ in range(300):
    pv(x)
ms = {v: [] for v in variants}
same = {}
for r in range(R):
    for v in variants:
        net_hip.tune(7, v)
        for _ in range(20):
            p, w = pv(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            p, w = pv(x)
        e1.record()
        torch.cuda.synchronize()
        ms[v].append(e0.elapsed_time(e1) / N)
        same[v] = bool(torch.equal(p, p0) and torch.equal(w, v0))
net_hip.tune(7, 0)
for v in variants:
    a = sorted(ms[v])
    print(json.dumps({"abl": hex(v), "ms_median": a[len(a) // 2], "ms_min": a[0], "ms_max": a[-1], "bit_identical": same[v],
                      "rounds": [round(t, 4) for t in ms[v]]}))
