"""Scratch probe: tick-kernel time over the first ~700 ticks of config 2 (HIP net)."""
import os, sys, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from alphafive_amd.engine import SelfPlayEngine
from alphafive_amd.network import ResNet
cfg = types.SimpleNamespace(board_size=11, goal=5, simulation_per_step=500, upper_simulation_per_step=642, init_temp=1.2,
                            gamma=0.94, tau_decay_rate=0.94, tau_decay_rate_r=0.9, dirichlet_alpha=0.3, c_puct=5.0)
net = ResNet(11, device="cuda"); net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
pv = net.select_backend("hip")
sp = SelfPlayEngine(cfg, 4096, pv, seed=0)
T = int(os.environ.get("TICKS", 700))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(T)]
stream = torch.cuda.current_stream().cuda_stream
for i in range(T):
    ev[i][0].record()
    sp.engine.tick(sp.policy.data_ptr(), sp.value.data_ptr(), sp.planes.data_ptr(), stream)
    ev[i][1].record()
    p, v = pv(sp.planes); sp.policy.copy_(p); sp.value.copy_(v)
torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) for a, b in ev])
print(f"tick kernel ms: mean {t.mean():.4f}  first100 {t[:100].mean():.4f}  ticks 500-700 {t[500:].mean():.4f}  max {t.max():.4f}")
print(sp.counters())
