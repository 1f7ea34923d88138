"""Where af_policy_fc_f16s spends its 31 us: -DAF_PFC_TIMING build, per-wave stamps of the device clock.
    hipcc ... -DAF_PFC_TIMING -o alphafive_amd/_lib/variants/libaf_net_pfctiming.so csrc/af_net.hip csrc/af_conv_f16s.hip
    AF_NET_LIB=.../libaf_net_pfctiming.so python tools/probe_pfc_timing.py"""
import ctypes as C, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from alphafive_amd import net_hip
from alphafive_amd.network import ResNet
from test_gpu_net import _positions
net = ResNet(11, device="cuda"); net.load_npz(os.path.join(R, "tests", "golden", "alphaFive-6960.weights.npz"))
B = 4096
hn = net_hip.HipNet(net.variables, 11, B, "cuda")
xb = torch.from_numpy(_positions(11, B, seed=1)).cuda()
net_hip.tune(4, 0)
L = net_hip.lib(); L.af_pfc_debug_wall.argtypes = [C.POINTER(C.c_uint64)]
rows = []
for rep in range(6):
    for _ in range(3): hn(xb)
    torch.cuda.synchronize()
    buf = np.zeros((256, 8, 5), np.uint64)
    assert L.af_pfc_debug_wall(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    d = buf[:128].astype(np.float64) * 0.01
    t0 = d[:, :, 0].min()
    rows.append(np.stack([(d[:, :, i] - t0) for i in range(5)], -1))
r = np.median(np.array(rows), 0)           # [wg][wave][stamp] us since the first wave's entry
names = ["entry", "first operands in place", "K loop done", "K halves combined", "softmax + stores done"]
for i, n in enumerate(names):
    print("%-26s mean %6.2f us   min %6.2f   max %6.2f" % (n, r[:, :, i].mean(), r[:, :, i].min(), r[:, :, i].max()))
print("per wave, K loop: kh=0 waves %.2f us, kh=1 waves %.2f us" % ((r[:, :4, 2] - r[:, :4, 1]).mean(), (r[:, 4:, 2] - r[:, 4:, 1]).mean()))
