"""Arithmetic width next to speed: the forward at 4096 positions on every convolution path of af_net_tune(0, .)
— 5 = the default fp16 split-operand path (22 mantissa bits), 1 = the fp32 MFMA Winograd path (24 bits; r6_02 also has the removed 0 / 2 / 3 / 4) — timed with HIP events, and its
worst |dv| / |dp| against (a) the fp64 restatement on 512 positions and (b) ResNet.eval_torch (PyTorch-ROCm fp32 ops) on all of them."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from alphafive_amd import net_hip                                  # noqa: E402
from alphafive_amd.network import ResNet                           # noqa: E402
from oracle import net_fp64                                        # noqa: E402
from test_gpu_net import _positions                                # noqa: E402

B = int(os.environ.get("B", 4096))
net = ResNet(11, device="cuda")
net.load_npz(os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz"))
pv = net.select_backend("hip")
x = _positions(11, B, seed=1)
xt = torch.from_numpy(x).cuda()
p64, v64 = net_fp64.forward(net.variables, x[:512])
pt, vt = net.eval_torch(xt)
out = {"B": B, "torch_fp32_vs_fp64": {"dv": float(np.abs(vt[:512].cpu().numpy() - v64).max()), "dp": float(np.abs(pt[:512].cpu().numpy() - p64).max())}}
try:
    for mode in (5, 1):
        net_hip.tune(0, mode)
        for _ in range(3):
            p, v = pv(xt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            p, v = pv(xt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out["mode%d" % mode] = {"ms": ms, "dv_fp64": float(np.abs(v[:512].cpu().numpy() - v64).max()),
                                "dp_fp64": float(np.abs(p[:512].cpu().numpy() - p64).max()),
                                "dv_torch": float((v - vt).abs().max()), "dp_torch": float((p - pt).abs().max())}
finally:
    net_hip.tune(0, 5)
print(json.dumps(out, indent=1))
