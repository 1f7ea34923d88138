"""Batch-1..8 forward on 11x11: the single launch of dataflow roles (default) vs the nine dependent launches (af_net_tune(7, 2048)),
eager and inside a HIP graph of 16 forwards.  Env: BS (default "1,2,4,8"), N (2000)."""
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from alphafive_amd import net_hip                                     # noqa: E402
from alphafive_amd.network import ResNet                               # noqa: E402

N = int(os.environ.get("N", 2000))
net = ResNet(11, device="cuda")
net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
out = {}
for B in [int(b) for b in os.environ.get("BS", "1,2,4,8").split(",")]:
    h = net_hip.HipNet(net.variables, 11, 8, net.device)
    x = (torch.rand((B, 3, 11, 11), device="cuda") < 0.2).float()
    for name, bits in (("single_launch", 0), ("nine_launches", 2048)):
        net_hip.tune(7, bits)
        for _ in range(50):
            h(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            h(x)
        e1.record()
        torch.cuda.synchronize()
        eager = 1e3 * e0.elapsed_time(e1) / N
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(16):
                h(x)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(N // 16):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out["B%d_%s" % (B, name)] = {"eager_us": round(eager, 2), "graph_us": round(1e3 * e0.elapsed_time(e1) / (N // 16 * 16), 2)}
        del g
    net_hip.tune(7, 0)
    out["B%d_error_flag" % B] = h.small_forward_error()
    h.close()
print(json.dumps(out, indent=1))
