"""Scratch probe: HIP net vs torch net (numerics + time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from alphafive_amd.network import ResNet
from oracle import net_fp64
S = int(os.environ.get("S", 11)); B = int(os.environ.get("B", 4096))
net = ResNet(S, device="cuda", seed=1)
if S == 11: net.load_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/alphaFive-6960.weights.npz"))
rng = np.random.RandomState(0)
x = np.zeros((B, 3, S, S), np.float32)
for b in range(B):
    n = rng.randint(0, S*S-1); cells = rng.permutation(S*S)[:n+1]
    x[b, 0].reshape(-1)[cells[0:n:2]] = 1; x[b, 1].reshape(-1)[cells[1:n:2]] = 1; x[b, 2].reshape(-1)[cells[n]] = 1
xt = torch.from_numpy(x).cuda()
pt, vt = net.eval_torch(xt)
pv = net.select_backend("hip")
ph, vh = pv(xt)
torch.cuda.synchronize()
print("hip vs torch: |dp| %.3e |dv| %.3e" % ((ph-pt).abs().max().item(), (vh-vt).abs().max().item()))
p64, v64 = net_fp64.forward(net.variables, x[:16])
print("hip vs fp64 : |dp| %.3e |dv| %.3e" % (np.abs(ph[:16].cpu().numpy()-p64).max(), np.abs(vh[:16].cpu().numpy()-v64).max()))
print("torch vs fp64: |dp| %.3e |dv| %.3e" % (np.abs(pt[:16].cpu().numpy()-p64).max(), np.abs(vt[:16].cpu().numpy()-v64).max()))
for name, fn in (("torch", net.eval_torch), ("hip", pv)):
    for _ in range(5): fn(xt)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): fn(xt)
    torch.cuda.synchronize(); dt = (time.time()-t0)/20
    print(f"{name}: {dt*1e3:.3f} ms/forward  {B*118.727264e6/dt/1e12 if S==11 else 0:.1f} TFLOP/s")
if os.environ.get("TUNES"):
    from alphafive_amd import net_hip
    for spec in os.environ["TUNES"].split(";"):
        net_hip.tune(0, 1); net_hip.tune(3, 0); net_hip.tune(4, 1); net_hip.tune(1, 1); net_hip.tune(6, 256)
        for kv in spec.split(","):
            if kv: net_hip.tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
        for _ in range(5): pv(xt)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): pv(xt)
        torch.cuda.synchronize(); dt = (time.time()-t0)/20
        ph2, vh2 = pv(xt)
        print(f"tune[{spec}]: {dt*1e3:.3f} ms  {B*118.727264e6/dt/1e12:.1f} TF  |dp| {(ph2-pt).abs().max().item():.2e}")
