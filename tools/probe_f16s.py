"""GPU probe: the fp16 split-operand conv path (af_net_tune(0, 5)) layer by layer against a torch fp64 evaluation,
final outputs against oracle/net_fp64.py, and forward time vs the fp32 Winograd path."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from alphafive_amd import net_hip
from alphafive_amd.network import ResNet
from oracle import net_fp64
from test_gpu_net import _positions

W = os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz")
net = ResNet(11, device="cuda")
net.load_npz(W)
B = int(os.environ.get("B", 4096))
hn = net_hip.HipNet(net.variables, 11, B, "cuda")
x = _positions(11, 512)
xt = torch.from_numpy(x).cuda()
net_hip.tune(0, 5)
p, v = hn(xt)
torch.cuda.synchronize()
p, v = p.cpu().numpy().copy(), v.cpu().numpy().copy()

# reference activations in fp64 (torch CPU)
V = {k: torch.from_numpy(a).double() for k, a in net.variables.items()}
def conv(f, name, act):
    k = V[name + "/kernel"].permute(3, 2, 0, 1)
    y = F.conv2d(f, k, V[name + "/bias"], padding=k.shape[-1] // 2)
    return F.elu(y) if act else y
def res(f, name):
    g = conv(f, name + "_conv1", True)
    return g, F.elu(conv(f, name + "_res", False) + conv(g, name + "_conv2", False))
nb = 16
f0 = conv(torch.from_numpy(x[:nb]).double(), "bone/conv1", True)
g1, o1 = res(f0, "bone/block1")
g2, o2 = res(o1, "bone/block2")
g3, o3 = res(o2, "value/block3")
g4, o4 = res(o2, "policy/block4")
g5, o5 = res(o4, "policy/block5")
refs = [f0, g1, o1, g2, o2, g3, g4, o4, g5]
names = ["stem", "b1.conv1", "b1.out", "b2.conv1", "b2.out", "b3.conv1", "b4.conv1", "b4.out", "b5.conv1"]
L = net_hip.lib()
L.af_net_debug_activation.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
for i, (r, nm) in enumerate(zip(refs, names)):
    Cc = r.shape[1]
    buf = np.zeros((nb, Cc, 121), np.float32)
    rc = L.af_net_debug_activation(hn._h, i, nb, buf.ctypes.data_as(C.POINTER(C.c_float)))
    ref = r.numpy().reshape(nb, Cc, 121)
    err = np.abs(buf - ref)
    print("%-10s C=%3d rc=%d  max|err| %.3e  (max|ref| %.2f)  worst at %s" % (nm, Cc, rc, err.max(), np.abs(ref).max(),
          np.unravel_index(err.argmax(), err.shape)), flush=True)
NREF = int(os.environ.get("NREF", 96))
p64, v64 = net_fp64.forward(net.variables, x[:NREF])
print("final (%d positions): max|dv| %.3e max|dp| %.3e (bar 1e-5)" % (NREF, np.abs(v[:NREF] - v64).max(), np.abs(p[:NREF] - p64).max()), flush=True)
for mode, bits in ((1, 0),):
    net_hip.tune(0, mode); net_hip.tune(7, bits)
    pm, vm = hn(xt)
    pm, vm = pm.cpu().numpy(), vm.cpu().numpy()
    print("  conv path %d opt %d: max|dv| %.3e max|dp| %.3e" % (mode, bits, np.abs(vm[:NREF] - v64).max(), np.abs(pm[:NREF] - p64).max()), flush=True)
pt, vt = net.eval_device(xt)
print("  PyTorch-ROCm fp32 ops: max|dv| %.3e max|dp| %.3e" % (np.abs(vt.cpu().numpy()[:NREF] - v64).max(), np.abs(pt.cpu().numpy()[:NREF] - p64).max()), flush=True)
net_hip.tune(0, 5); net_hip.tune(7, 0)
# determinism
p2, v2 = hn(xt)
print("deterministic:", bool((p2.cpu().numpy() == p).all() and (v2.cpu().numpy() == v).all()))

# timing
xb = torch.from_numpy(_positions(11, B, seed=1)).cuda()
for mode, br in ((1, 1), (5, 1), (5, 0)):
    net_hip.tune(0, mode); net_hip.tune(4, br)
    for _ in range(3):
        hn(xb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hn(xb)
    e1.record()
    torch.cuda.synchronize()
    print("mode %d (value branch on side stream: %d): %.3f ms per forward of %d positions" % (mode, br, e0.elapsed_time(e1) / 20, B), flush=True)
net_hip.tune(4, 1)
if os.environ.get("ABL"):
    net_hip.tune(0, 5)
    for bits in (1, 2, 3):
        net_hip.tune(7, bits)
        for _ in range(2):
            hn(xb)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            hn(xb)
        e1.record()
        torch.cuda.synchronize()
        print("mode 5 ablation bits %d: %.3f ms" % (bits, e0.elapsed_time(e1) / 10), flush=True)
    net_hip.tune(7, 0)
