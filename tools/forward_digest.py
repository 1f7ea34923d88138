"""CRC32 digests of the hand-written forward's outputs on fixed inputs (11x11 alphaFive-6960, 15x15 random init seed 16): the same
build must print the same digests on every box; tests/test_gpu_net.py pins them (a change of ANY bit of the arithmetic shows here
even where it stays inside the 1e-5 bar)."""
import os
import sys
import zlib

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from alphafive_amd.network import ResNet                               # noqa: E402
from test_gpu_net import _positions                                    # noqa: E402


def digests():
    out = {}
    for S, B, seed in ((11, 600, 3), (11, 5, 4), (15, 300, 5), (15, 3, 6)):
        net = ResNet(S, device="cuda", seed=S + 1)
        if S == 11:
            net.load_npz(os.path.join(R, "tests", "golden", "alphaFive-6960.weights.npz"))
        pv = net.select_backend("hip")
        p, v = pv(torch.from_numpy(_positions(S, B, seed=seed)).cuda())
        out["S%d_B%d" % (S, B)] = "%08x%08x" % (zlib.crc32(p.cpu().numpy().tobytes()), zlib.crc32(v.cpu().numpy().tobytes()))
        net.close()
    return out


if __name__ == "__main__":
    print(digests())
