"""Start-up of an af_conv_f16s launch (kernel entry -> weights and first slabs in place), per layer, from the -DAF_F16S_TIMING build
(see tools/probe_f16s_timing.py for the build line): AF_NET_LIB=.../libaf_net_timing.so python tools/probe_f16s_ramp.py"""
import ctypes as C, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from alphafive_amd import net_hip
from alphafive_amd.network import ResNet
from test_gpu_net import _positions
net = ResNet(11, device="cuda"); net.load_npz(os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz"))
B = 4096
hn = net_hip.HipNet(net.variables, 11, B, "cuda")
xb = torch.from_numpy(_positions(11, B, seed=1)).cuda()
net_hip.tune(0, 5); net_hip.tune(4, 0)
for _ in range(3): hn(xb)
torch.cuda.synchronize()
L = net_hip.lib()
buf = np.zeros((10, 512, 4, 9), np.uint64)
L.af_f16s_debug_cycles.argtypes = [C.POINTER(C.c_uint64)]
assert L.af_f16s_debug_cycles(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
for li in range(10):
    d = buf[li].reshape(-1, 9).astype(np.float64); d = d[d[:, 6] > 0]
    print("L%d ramp cycles: mean %.0f  p10 %.0f p90 %.0f max %.0f | total per wave %.0f" % (li + 1, d[:, 8].mean(), np.percentile(d[:, 8], 10), np.percentile(d[:, 8], 90), d[:, 8].max(), d[:, 5].mean()))
