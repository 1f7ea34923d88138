"""Where a wave of af_tick_kernel spends its cycles at steady state (config 2): build the profiling variant
   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DAF_TICK_TIMING -Iinclude -Ialphafive_amd/csrc -o alphafive_amd/_lib/libaf_hip_timing.so alphafive_amd/csrc/af_engine.hip
then AF_HIP_LIB=.../libaf_hip_timing.so python tools/probe_tick_timing.py   (TICKS ticks of warm-up, then SAMPLES launches are read back)"""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from alphafive_amd import engine as eng
from alphafive_amd.engine import SelfPlayEngine
from alphafive_amd.network import ResNet

cfg = types.SimpleNamespace(board_size=11, goal=5, simulation_per_step=500, upper_simulation_per_step=642, init_temp=1.2,
                            gamma=0.94, tau_decay_rate=0.94, tau_decay_rate_r=0.9, dirichlet_alpha=0.3, c_puct=5.0)
net = ResNet(11, device="cuda")
net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
G = 4096
sp = SelfPlayEngine(cfg, G, net.select_backend("hip"), seed=0)
sp.run_ticks(int(os.environ.get("TICKS", 2500)))
L = eng.lib()
L.af_engine_debug_tick_cycles.argtypes = [C.POINTER(C.c_uint64)]
names = ["state load", "consume (expand + backup)", "move boundary", "terminal test + lookup", "select", "park + state store", "total", "selects",
         "  select: rows + noise rounds", "  select: noise sum + normalise", "  select: scores", "  select: argmax, pick, prefetch, step"]
acc = np.zeros((0, 13))
for _ in range(int(os.environ.get("SAMPLES", 40))):
    sp.run_ticks(7)
    buf = np.zeros((8192, 13), np.uint64)
    assert L.af_engine_debug_tick_cycles(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    acc = np.concatenate([acc, buf[:G].astype(np.float64)])
tot = acc[:, 6]
print("waves sampled %d; mean total %.0f cycles, p50 %.0f, p99 %.0f, max %.0f; selects per launch %.2f" % (len(acc), tot.mean(), np.percentile(tot, 50), np.percentile(tot, 99), tot.max(), acc[:, 7].mean()))
for q in range(6):
    print("  %-28s %6.1f %% of wave cycles  (mean %.0f cycles per launch%s)" % (names[q], 100 * acc[:, q].sum() / tot.sum(), acc[:, q].mean(),
          (", %.0f per select" % (acc[:, q].sum() / acc[:, 7].sum())) if q in (3, 4) else ""))
for q in range(8, 12):
    print("  %-40s %6.1f %% of wave cycles  (%.0f cycles per select)" % (names[q], 100 * acc[:, q].sum() / tot.sum(), acc[:, q].sum() / acc[:, 7].sum()))
print("  rejection-loop iterations of a wave per select: %.2f (every lane draws its 2 cells one after the other: the wave runs until its slowest lane has both)" % (acc[:, 12].sum() / acc[:, 7].sum()))
slow = acc[tot >= np.percentile(tot, 99)]
print("slowest 1 %% of waves: selects %.1f, %.0f cycles; shares: " % (slow[:, 7].mean(), slow[:, 6].mean()) + ", ".join("%s %.0f %%" % (names[q].strip(), 100 * slow[:, q].sum() / slow[:, 6].sum()) for q in list(range(6)) + list(range(8, 12))))
print("slowest 1 %%: cycles per select: " + ", ".join("%s %.0f" % (names[q].strip(), slow[:, q].sum() / slow[:, 7].sum()) for q in (3, 4, 8, 9, 10, 11)))
