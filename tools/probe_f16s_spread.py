"""Spread of workgroup finish times inside each af_conv_f16s launch (what a device-wide barrier between layers costs): the
-DAF_F16S_TIMING build stamps every workgroup's entry and exit with the device-wide 100 MHz clock.
    hipcc ... -DAF_F16S_TIMING -o alphafive_amd/_lib/libaf_net_timing.so csrc/af_net.hip csrc/af_conv_f16s.hip
    AF_NET_LIB=.../libaf_net_timing.so python tools/probe_f16s_spread.py"""
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
B = 4096
hn = net_hip.HipNet(net.variables, 11, B, "cuda")
xb = torch.from_numpy(_positions(11, B, seed=1)).cuda()
net_hip.tune(0, 5)
net_hip.tune(4, int(os.environ.get("BRANCH", 0)))
L = net_hip.lib()
L.af_f16s_debug_wall.argtypes = [C.POINTER(C.c_uint64)]
names = ["L1 32->64", "L2 64->64+p32", "L3 64->128", "L4 128->128+p64", "L5 128->32", "L6 32->32", "L7 128->64", "L8 64->64+p128",
         "L9 64->32", "L10 32->32"]
acc = []
for rep in range(int(os.environ.get("REPS", 6))):
    for _ in range(3):
        hn(xb)
    torch.cuda.synchronize()
    buf = np.zeros((10, 512, 2), np.uint64)
    assert L.af_f16s_debug_wall(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    acc.append(buf.astype(np.float64) * 0.01)          # us
print("%-16s %5s | %8s %8s %8s %8s | %8s %8s | %s" % ("layer", "WGs", "dur us", "mean end", "p10 end", "min end", "lost us", "lost %", "entry spread us"))
tot_d = tot_l = 0.0
for li in range(10):
    rows = []
    for b in acc:
        d = b[li]
        d = d[d[:, 1] > 0]
        t0 = d[:, 0].min()
        end = d[:, 1] - t0
        rows.append((len(d), end.max(), end.mean(), np.percentile(end, 10), end.min(), (d[:, 0] - t0).max()))
    r = np.median(np.array(rows), axis=0)
    lost = r[1] - r[2]
    tot_d += r[1]
    tot_l += lost
    print("%-16s %5d | %8.1f %8.1f %8.1f %8.1f | %8.1f %7.1f%% | %.1f" % (names[li], r[0], r[1], r[2], r[3], r[4], lost, 100 * lost / r[1], r[5]))
print("sum of launch durations %.1f us; sum of (last - mean finish) %.1f us = %.1f %%: what per-layer device-wide barriers cost against a "
      "barrier-free chain in which every workgroup carries its own positions through all layers" % (tot_d, tot_l, 100 * tot_l / tot_d))

# is a workgroup's lateness systematic (the same CUs / XCDs late in every layer: a barrier-free chain would not help) or random?
idx = [li for li in range(10) if (acc[0][li][:, 1] > 0).sum() == 256]
late = []
for li in idx:
    m = np.mean([b[li][:256, 1] - b[li][:256, 0].min() for b in acc], axis=0)
    late.append((m - m.mean()) / m.mean())
late = np.array(late)
print("relative lateness of workgroup b (mean over repetitions), correlation between layers:")
print(np.round(np.corrcoef(late), 2))
print("per XCD (workgroup id mod 8), mean relative lateness over the 256-workgroup layers, %:",
      np.round(100 * late.mean(0).reshape(32, 8).mean(0), 2).tolist())
chain = np.sum([np.mean([b[li][:256, 1] - b[li][:256, 0].min() for b in acc], axis=0) for li in idx], axis=0)
bar = sum(np.mean([(b[li][:256, 1] - b[li][:256, 0].min()).max() for b in acc]) for li in idx)
print("256-workgroup layers: sum of per-layer maxima %.1f us; max over workgroups of the per-workgroup sums %.1f us; mean of the sums %.1f us"
      % (bar, chain.max(), chain.mean()))
# same workgroup, same layer, different repetitions: how repeatable is the lateness?
li = idx[3] if len(idx) > 3 else idx[0]
e = np.array([b[li][:256, 1] - b[li][:256, 0].min() for b in acc])
print("layer %s: repetition-to-repetition correlation of workgroup finish times %.2f" % (names[li], np.corrcoef(e)[0, 1:].mean()))
