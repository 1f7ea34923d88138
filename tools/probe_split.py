"""Scratch probe: time split between the tree tick kernel and the net at config 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types
import numpy as np, torch
from alphafive_amd.engine import SelfPlayEngine
from alphafive_amd.network import ResNet

G = int(os.environ.get("G", 4096)); TICKS = int(os.environ.get("TICKS", 600))
cfg = types.SimpleNamespace(board_size=11, goal=5, simulation_per_step=500, upper_simulation_per_step=642, init_temp=1.2,
                            gamma=0.94, tau_decay_rate=0.94, tau_decay_rate_r=0.9, dirichlet_alpha=0.3, c_puct=5.0)
net = ResNet(11, device="cuda"); net.load_npz("tests/golden/alphaFive-6960.weights.npz")
sp = SelfPlayEngine(cfg, G, net.eval_device, seed=0)
for _ in range(20): sp.tick()
torch.cuda.synchronize()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(TICKS)]
stream = torch.cuda.current_stream().cuda_stream
t0 = time.time()
for i in range(TICKS):
    ev[i][0].record()
    sp.engine.tick(sp.policy.data_ptr(), sp.value.data_ptr(), sp.planes.data_ptr(), stream)
    ev[i][1].record()
    p, v = net.eval_device(sp.planes); sp.policy.copy_(p); sp.value.copy_(v)
    ev[i][2].record()
torch.cuda.synchronize(); wall = time.time() - t0
tt = np.array([e[0].elapsed_time(e[1]) for e in ev]); tn = np.array([e[1].elapsed_time(e[2]) for e in ev])
print(f"G={G} ticks={TICKS} wall={wall:.2f}s  tick kernel mean {tt.mean():.3f} ms (p50 {np.median(tt):.3f}, max {tt.max():.3f})  net mean {tn.mean():.3f} ms")
print("counters", sp.counters(), "progress", sp.progress())
print("net TFLOP/s:", G * 118.727264e6 / (tn.mean() * 1e-3) / 1e12)
