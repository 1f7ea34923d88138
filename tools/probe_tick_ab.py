"""A/B of af_tick_kernel builds at the steady-state mix of config 2: AF_HIP_LIB=<variant .so> python tools/probe_tick_ab.py
Games are aged 24 plies at 16 simulations per move (untimed, like bench.py's extra legs), rebuilt by 2 full-budget steps, then
TICKS ticks are timed with HIP events around the tick kernel alone.  Prints the mean / p50 launch time and a digest of the
engine's counters and progress: variants that claim to be bit-identical must print the same digest."""
import os, sys, types, zlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from alphafive_amd.engine import SelfPlayEngine
from alphafive_amd.network import ResNet
S = int(os.environ.get("S", 11))
sims, upper = (500, 642) if S == 11 else (800, 942)
cfg = types.SimpleNamespace(board_size=S, goal=5, simulation_per_step=sims, upper_simulation_per_step=upper, init_temp=1.2,
                            gamma=0.94, tau_decay_rate=0.94, tau_decay_rate_r=0.9, dirichlet_alpha=0.3, c_puct=5.0)
net = ResNet(S, device="cuda")
if S == 11:
    net.load_npz(os.path.join(R, "tests/golden/alphaFive-6960.weights.npz"))
pv = net.select_backend("hip")
G = int(os.environ.get("G", 4096))
sp = SelfPlayEngine(cfg, G, pv, seed=0)


def step(target):
    while sp.progress()[0] < target:
        sp.run_ticks(16)
    sp.check()
    sp.pop_raw(512)


sp.engine.set_simulations(16, 24)
t = 0
for _ in range(int(os.environ.get("AGE", 24))):
    t += G
    step(t)
sp.engine.set_simulations(sims, upper)
for _ in range(2):
    t += G
    step(t)
T = int(os.environ.get("TICKS", 600))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(T)]
stream = torch.cuda.current_stream().cuda_stream
for i in range(T):
    ev[i][0].record()
    sp.engine.tick(sp.policy.data_ptr(), sp.value.data_ptr(), sp.planes.data_ptr(), stream)
    ev[i][1].record()
    pv(sp.planes)
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in ev])
ct = sp.counters()
dig = zlib.crc32(repr(sorted(ct.items())).encode() + repr(sp.progress()).encode())
print("%-40s tick ms mean %.4f p50 %.4f max %.4f | selects/launch %.2f | digest %08x" % (
    os.path.basename(os.environ.get("AF_HIP_LIB", "libaf_hip.so")), ms.mean(), np.median(ms), ms.max(), 0.0, dig))
