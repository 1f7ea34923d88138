"""Timings of the callers of the hot path next to their host forms, one JSON line each:
  replay   utils.RandomStack.get_data(512) (host numpy, the reference's algorithm) vs DeviceRandomStack.get_data(512) (device ring)
  hand-off engine -> replay, device to device (af_replay_append_packed), episodes/s
  arena    alphafive_amd.arena.play_matches: N batched games of two weight sets, games/s and moves/s"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from alphafive_amd import utils                                       # noqa: E402
from alphafive_amd.replay import DeviceRandomStack                     # noqa: E402
from alphafive_amd.network import ResNet, random_variables             # noqa: E402
from alphafive_amd.engine import SelfPlayEngine                        # noqa: E402
from alphafive_amd import arena                                        # noqa: E402
from bench import make_cfg                                             # noqa: E402
from test_gpu_replay import _episodes                                  # noqa: E402

W = os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz")
random.seed(1)
np.random.seed(1)
eps = _episodes(11, 120, seed=3)
host, devs = utils.RandomStack(11, 2000), DeviceRandomStack(11, 2000, device=0)
for rec, res in eps:
    host.push(rec, res)
    devs.push(rec, res)
for name, st in (("host RandomStack (numpy)", host), ("DeviceRandomStack (HBM ring)", devs)):
    for _ in range(3):
        st.get_data(512)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        b = st.get_data(512)
    torch.cuda.synchronize()
    print(json.dumps({"row": "f1 get_data(512)", "impl": name, "ms": (time.perf_counter() - t0) / 30 * 1e3,
                      "positions_in_buffer": int(st._size()) if hasattr(st, "_size") else None}), flush=True)

cfg = make_cfg(60, 80, 11)
net = ResNet(11, device="cuda")
net.load_npz(W)
sp = SelfPlayEngine(cfg, 1024, net.select_backend("hip"), device=0, seed=0)
st = DeviceRandomStack(11, 200000, device=0)
pushed, t_push = 0, 0.0
t0 = time.perf_counter()
while pushed < 2000:
    sp.run_ticks(256)
    sp.check()
    t1 = time.perf_counter()
    for r in st.iter_push_packed(sp.post_episodes_device(256), 256, cfg.gamma):
        pushed += 1
    st.check()
    torch.cuda.synchronize()
    t_push += time.perf_counter() - t1
print(json.dumps({"row": "f1 hand-off engine -> replay, device to device", "episodes": pushed, "hand_off_s": t_push,
                  "episodes_per_s_of_hand_off_time": pushed / t_push, "whole_loop_s": time.perf_counter() - t0}), flush=True)
sp.close()

cfg = make_cfg(200, 260, 11)
n2 = ResNet(11, device="cuda")
n2.set_variables(random_variables(11, seed=2))
for G in (64, 512):
    t0 = time.perf_counter()
    r = arena.play_matches(cfg, net.select_backend("hip"), n2.select_backend("hip"), G, device=0)
    dt = time.perf_counter() - t0
    print(json.dumps({"row": "f4 arena (choose_best_player.py:38-60), 200 sims/move", "games": G, "s": dt, "games_per_s": G / dt,
                      "moves_per_s": float(sum(r["lengths"])) / dt, "wins": r["wins"], "draws": r["draws"]}), flush=True)
