"""Lean probe for counter collection on the tick kernel + net: ~60 ticks of config 2."""
import os, sys, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from alphafive_amd.engine import SelfPlayEngine
from alphafive_amd.network import ResNet
cfg = types.SimpleNamespace(board_size=11, goal=5, simulation_per_step=500, upper_simulation_per_step=642, init_temp=1.2,
                            gamma=0.94, tau_decay_rate=0.94, tau_decay_rate_r=0.9, dirichlet_alpha=0.3, c_puct=5.0)
net = ResNet(11, device="cuda"); net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
sp = SelfPlayEngine(cfg, int(os.environ.get("G", 4096)), net.select_backend("hip"), seed=0)
sp.run_ticks(int(os.environ.get("TICKS", 60)))
torch.cuda.synchronize()
ct = sp.counters()
print("counters", ct)
