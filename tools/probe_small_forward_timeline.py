"""Timeline of the single-launch small-batch forward (af_small_forward_f16s): per role, the device-wide 100 MHz clock at entry / wait
satisfied / body done / signalled / exit, relative to the first role's entry.  Needs the AF_DF_TIMING build:
    tools/build_f16s_variants.sh dftiming:"-DAF_DF_TIMING"
    AF_NET_LIB=alphafive_amd/_lib/variants/libaf_net_dftiming.so python tools/probe_small_forward_timeline.py
Env: B (1)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from alphafive_amd import net_hip                                     # noqa: E402
from alphafive_amd.network import ResNet                               # noqa: E402

B = int(os.environ.get("B", 1))
net = ResNet(11, device="cuda")
net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
h = net_hip.HipNet(net.variables, 11, 8, net.device)
x = (torch.rand((B, 3, 11, 11), device="cuda") < 0.2).float()
for _ in range(20):
    h(x)
torch.cuda.synchronize()
names = (["stem"] * B + ["L0 b1c1"] * 2 * B + ["L1 b1c2"] * 2 * B + ["L2 b2c1"] * 4 * B + ["L3 b2c2"] * 4 * B + ["P b4c1"] * 2 * B + ["V block3"] * B
         + ["Q b4c2"] * 2 * B + ["vfc"] + ["B5 block5"] * B + ["PF1", "PF0"])
acc = np.zeros((len(names), 5))
N = 50
for _ in range(N):
    h(x)
    w = np.zeros((256, 5), np.uint64)
    net_hip.lib().af_df_debug_wall(w.ctypes.data_as(C.c_void_p))
    w = w[:len(names)].astype(np.int64)
    acc += (w - w[:, 0].min()) * 0.01
acc /= N
print("%-12s %8s %8s %8s %8s %8s   (us after the first role's entry; mean of %d forwards, B = %d)" % ("role", "entry", "ready", "done", "signal", "exit", N, B))
for nm, r in zip(names, acc):
    print("%-12s %8.2f %8.2f %8.2f %8.2f %8.2f" % (nm, *r))
