"""How long does one optimiser step of alphafive_amd.train.Trainer take on the device, next to the rate at which the self-play
engine finishes episodes?  (main.py:62-70: four minibatches of config.batch_size = 512 per ACCEPTED episode.)
Measurement for SURVEY 8f rank 2 (a caller of the hot path); prints one JSON line.  Env: B (512), STEPS (60), SYNC (1: read the scalars every step)."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from alphafive_amd.train import Trainer                                # noqa: E402

B, STEPS = int(os.environ.get("B", 512)), int(os.environ.get("STEPS", 60))
dev = torch.device("cuda", 0)
with np.load(os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz")) as z:
    var = {k: z[k] for k in z.files}
tr = Trainer(var, 11, device=dev)
g = torch.Generator(device="cpu").manual_seed(0)
boards = (torch.rand((B, 3, 11, 11), generator=g) < 0.1).float().to(dev)
pol = torch.softmax(torch.randn((B, 121), generator=g), dim=1).to(dev)
val = (torch.randint(0, 2, (B,), generator=g).float() * 2 - 1).to(dev)
wts = torch.ones(B).to(dev)
for _ in range(5):
    tr.step(boards, wts, val, pol, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    m = tr.step(boards, wts, val, pol, 1e-3, metrics=os.environ.get("SYNC", "1") == "1")
torch.cuda.synchronize()
m = m or tr.step(boards, wts, val, pol, 1e-3)
dt = (time.perf_counter() - t0) / STEPS
flop = 3 * 118727264 * B                                               # forward + two backward passes of the same contraction
print(json.dumps({"batch": B, "sync_every_step": os.environ.get("SYNC", "1") == "1", "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "positions_per_s": B / dt,
                  "algorithmic_tflops": flop / dt / 1e12, "loss_total": m["total"],
                  "note": "PyTorch-ROCm autograd (MIOpen fp32 convolutions) + TF-style Adam; one host sync per step for the logged scalars"}))
