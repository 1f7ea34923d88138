"""Where a wave of af_conv_f16s spends its cycles, per layer: build libaf_net_timing.so with -DAF_F16S_TIMING
(hipcc ... -DAF_F16S_TIMING -o alphafive_amd/_lib/libaf_net_timing.so csrc/af_net.hip csrc/af_conv_f16s.hip), then
AF_NET_LIB=.../libaf_net_timing.so python tools/probe_f16s_timing.py"""
import ctypes as C
import os
import sys

import numpy as np
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
net_hip.tune(4, 0)
net_hip.tune(7, int(os.environ.get("ABLBITS", 0)))           # af_conv_f16s ablation bits (2 = no stores, 1 = no LDS-DMA after the first slabs)
for _ in range(3):
    hn(xb)
torch.cuda.synchronize()
L = net_hip.lib()
buf = np.zeros((10, 512, 4, 9), np.uint64)
L.af_f16s_debug_cycles.argtypes = [C.POINTER(C.c_uint64)]
assert L.af_f16s_debug_cycles(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
names = ["L1 32->64", "L2 64->64+p32", "L3 64->128", "L4 128->128+p64", "L5 128->32 (+proj out)", "L6 32->32 (+proj in)", "L7 128->64",
         "L8 64->64+p128", "L9 64->32 (+proj out)", "L10 32->32 (+proj in)"]
print("%-24s %6s %9s | %% of wave cycles: %7s %7s %7s %7s %7s %7s" % ("layer", "pos/WG", "cyc/pos", "items", "vmcnt", "barrier", "exchg", "(xbar)", "epilog"))
for li in range(10):
    d = buf[li].reshape(-1, 9).astype(np.float64)
    d = d[d[:, 6] > 0]
    tot = d[:, 5].sum()
    print("%-24s %6.1f %9.0f | %25.1f %7.1f %7.1f %7.1f %7.1f %7.1f" % (names[li], d[:, 6].mean(), tot / d[:, 6].sum(),
          100 * d[:, 0].sum() / tot, 100 * d[:, 1].sum() / tot, 100 * d[:, 2].sum() / tot, 100 * d[:, 3].sum() / tot, 100 * d[:, 7].sum() / tot,
          100 * d[:, 4].sum() / tot))
    w = buf[li].reshape(-1, 4, 9).astype(np.float64)
    w = w[w[:, 0, 6] > 0]
    print("      cycles per position: items %.0f  vmcnt %.0f  barrier %.0f  exchange %.0f  epilogue %.0f" % tuple(
        d[:, q].sum() / d[:, 6].sum() for q in (0, 1, 2, 3, 4)))
    print("      per wave: items %s  epilogue %s  exch-barrier %s (cycles per position)" % tuple(
        np.round(w[:, :, q].sum(0) / w[:, :, 6].sum(0)).astype(int).tolist() for q in (0, 4, 7)))
    print("      start-up (kernel entry -> weights and first slabs in place): %.0f cycles" % (w[:, :, 8].mean()))
