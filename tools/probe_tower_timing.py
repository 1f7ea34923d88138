"""Where a wave of af_tower_conv spends a position: -DAF_TOWER_TIMING build (engine 0), cycles per position of the MFMA loop, the wait
for the next position's planes, the barrier and the epilogue.
    hipcc ... -DAF_TOWER_TIMING -o alphafive_amd/_lib/variants/libaf_tower_timing.so alphafive_amd/csrc/af_tower_bf16.hip
    AF_TOWER_LIB=.../libaf_tower_timing.so python tools/probe_tower_timing.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from alphafive_amd import tower_hip
from alphafive_amd.network_deep import DeepResNet
B = 8192
net = DeepResNet(11, blocks=8, width=128, device="cuda")
net.select_backend("hip", B)
tower_hip.tune(3, 0)
tw = net._tower
tw.load_nchw(torch.randn((B, 128, 11, 11), device="cuda").bfloat16())
for _ in range(4):
    tw.forward(B)
torch.cuda.synchronize()
L = tower_hip.lib()
L.af_tower_debug_cycles.argtypes = [C.POINTER(C.c_uint64)]
buf = np.zeros((2, 256, 4, 5), np.uint64)
assert L.af_tower_debug_cycles(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
for kind, name in ((0, "first convolution (288 MFMAs)"), (1, "second convolution + projection (320 MFMAs)")):
    d = buf[kind].reshape(-1, 5).astype(np.float64)
    n = d[:, 4].sum()
    parts = d[:, :4].sum(0) / n
    print("%-44s cycles per position: MFMA loop %.0f | vmcnt wait %.0f | barrier %.0f | epilogue %.0f | total %.0f (MFMA issue alone: %d)" % (
        name, parts[0], parts[1], parts[2], parts[3], parts.sum(), (288 if kind == 0 else 320) * 32))
