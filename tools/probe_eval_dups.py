"""How many of the positions the self-play loop sends to the net were sent before?  (decision probe, not a test)

Every game owns its tree (genData/player.py: one Player per worker), but the net is a pure function of the position and the
forward is batch-slot independent to the bit (tests/test_gpu_net.py), so an evaluation another game (or an earlier episode of the
same game) already paid for could be reused without changing a single tree.  This measures the ceiling of that at configs[1]:
hash the planes of every parked leaf of every tick in steady state and count repeats
  * inside one tick's batch,
  * against a window of the last W ticks,
  * against everything seen since the recording started (an unbounded store).
Env: G (4096), WARM (ticks, 1500), REC (ticks, 4000), BOARD (11), SIMS/UPPER (500/642).
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import make_cfg                                            # noqa: E402
from alphafive_amd.engine import SelfPlayEngine                        # noqa: E402
from alphafive_amd.network import ResNet                               # noqa: E402

G = int(os.environ.get("G", 4096))
WARM, REC = int(os.environ.get("WARM", 1500)), int(os.environ.get("REC", 4000))
cfg = make_cfg(int(os.environ.get("SIMS", 500)), int(os.environ.get("UPPER", 642)), int(os.environ.get("BOARD", 11)))
dev = torch.device("cuda", 0)
net = ResNet(cfg.board_size, device=dev, seed=0)
if cfg.board_size == 11:
    net.load_npz(os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz"))
pv = net.select_backend("hip")
sp = SelfPlayEngine(cfg, G, pv, device=0, seed=0)
t0 = time.time()
for _ in range(WARM // 50):
    sp.run_ticks(50)
    sp.check()
    sp.pop_raw(512)
torch.cuda.synchronize()
print(f"warm: {WARM} ticks in {time.time() - t0:.1f} s, progress {sp.progress()}", flush=True)

gen = torch.Generator(device="cpu").manual_seed(1234)
zob = torch.randint(-2 ** 62, 2 ** 62, (3 * cfg.board_size ** 2,), generator=gen, dtype=torch.int64).to(dev)
keys = torch.zeros((REC, G), dtype=torch.int64, device=dev)
live = torch.zeros((REC, G), dtype=torch.bool, device=dev)
stones = torch.zeros((REC, G), dtype=torch.int16, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
for t in range(REC):
    sp.engine.tick(sp.policy.data_ptr(), sp.value.data_ptr(), sp.planes.data_ptr(), stream)
    st = torch.from_numpy(sp.engine.status(stream).copy()).to(dev)
    pl = sp.planes.view(G, -1)
    keys[t] = ((pl > 0).to(torch.int64) * zob).sum(1)
    live[t] = st == 1                                                  # AF_STATUS_NEED_EVAL: the slot's planes are this tick's
    stones[t] = (sp.planes[:, :2] > 0).view(G, -1).sum(1).to(torch.int16)
    pv(sp.planes)
    sp.ticks += 1
    if (t + 1) % 50 == 0:
        sp.check()
        sp.pop_raw(512)
torch.cuda.synchronize()

k = keys[live]
tick_of = torch.arange(REC, device=dev).view(-1, 1).expand(REC, G)[live]
n = k.numel()
print(f"recorded {REC} ticks, {n} evaluations ({n / (REC * G):.4f} of the slots)")
# unbounded store: first occurrence of every key
order = torch.argsort(k, stable=True)
ks, ts = k[order], tick_of[order]
first = torch.ones(n, dtype=torch.bool, device=dev)
first[1:] = ks[1:] != ks[:-1]
print(f"unbounded store over the recording: {1 - first.sum().item() / n:.4f} repeats")
# same, counting only the second half of the recording (the store is warm by then)
half = ts >= REC // 2
print(f"  second half only (store warmed by the first): {1 - (first & half).sum().item() / half.sum().item():.4f} repeats")
# inside one tick
same_tick = torch.zeros(n, dtype=torch.bool, device=dev)
same_tick[1:] = (~first[1:]) & (ts[1:] == ts[:-1])
print(f"inside one tick's batch: {same_tick.sum().item() / n:.4f}")
# window of W ticks: distance to the previous occurrence (stable sort keeps tick order inside a key)
gap = torch.full((n,), 1 << 30, dtype=torch.int64, device=dev)
gap[1:] = torch.where(first[1:], gap[1:], ts[1:] - ts[:-1])
for W in (1, 4, 16, 64, 256, 1024):
    print(f"  previous occurrence within {W:5d} ticks: {(gap <= W).sum().item() / n:.4f}")
# by the number of stones on the evaluated board
sb = stones[live][order].to(torch.int64)
rep = ~first
print("stones on the leaf : share of evaluations : repeats among them (unbounded store)")
for lo, hi in ((0, 2), (2, 4), (4, 6), (6, 8), (8, 12), (12, 20), (20, 121)):
    m = (sb >= lo) & (sb < hi)
    c = m.sum().item()
    if c:
        print(f"  [{lo:3d},{hi:3d}) : {c / n:.4f} : {(rep & m).sum().item() / c:.4f}")
