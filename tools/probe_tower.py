"""Scratch probe: hand-written bf16 tower vs PyTorch ops (numerics + time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from alphafive_amd.network_deep import DeepResNet
B = int(os.environ.get("B", 8192)); NB = int(os.environ.get("BLOCKS", 8))
net = DeepResNet(11, blocks=NB, width=128, device="cuda")
g = torch.Generator().manual_seed(5)
for blk in net.tower:
    for k in ("res", "c1", "c2"):
        blk[k] = (blk[k][0], (torch.randn(128, generator=g) * 0.1).to("cuda", torch.bfloat16))
x = (torch.rand((B, 3, 11, 11), device="cuda") < 0.2).float()
pv = net.select_backend("hip", B)
tw = net._tower
h0 = F.elu(F.conv2d(x.to(torch.bfloat16), net.stem[0], net.stem[1], padding=2))
# fp32 reference with bf16 rounding at the same points (after each ELU)
def ref32(h):
    h = h.float()
    for blk in net.tower:
        w = {k: (blk[k][0].float(), blk[k][1].float()) for k in blk}
        gg = F.elu(F.conv2d(h, *w["c1"], padding=1)).bfloat16().float()
        h = F.elu(F.conv2d(h, *w["res"]) + F.conv2d(gg, *w["c2"], padding=1)).bfloat16().float()
    return h
n = min(B, 64)
tw.load_nchw(h0); tw.forward(B); out = tw.store_nchw(B).float()
r32 = ref32(h0[:n]); rt = net.tower_reference(h0[:n]).float()
print("hip vs fp32-ref: max|d| %.4f  mean|d| %.6f  max|ref| %.3f" % ((out[:n]-r32).abs().max().item(), (out[:n]-r32).abs().mean().item(), r32.abs().max().item()))
print("torch-bf16 vs fp32-ref: max|d| %.4f mean|d| %.6f" % ((rt-r32).abs().max().item(), (rt-r32).abs().mean().item()))
print("borders still zero:", float(tw.x.float().abs().sum() - tw._xin.float().abs().sum()) == 0.0)
p1, v1 = pv(x); p2, v2 = net.eval_device(x); p3, v3 = net.eval_hip_torch_ends(x)
# stem / heads kernels vs torch ops on the same data
tw.stem(x); xs = tw.store_nchw(B).float(); print("stem |d| %.4f (max %.3f)" % ((xs - h0.float()).abs().max().item(), h0.float().abs().max().item()))
print("hip-full vs hip-with-torch-ends: policy |d| %.2e value |d| %.2e" % ((p1-p3).abs().max().item(), (v1-v3).abs().max().item()))
print("policy |d| %.2e value |d| %.2e" % ((p1-p2).abs().max().item(), (v1-v2).abs().max().item()))
for name, fn in (("tower hip", lambda: tw.forward(B)), ("tower torch", lambda: net.tower_reference(h0)), ("net hip", lambda: pv(x)), ("net hip + torch ends", lambda: net.eval_hip_torch_ends(x)), ("stem hip", lambda: tw.stem(x)), ("heads hip", lambda: tw.heads(B)), ("net torch", lambda: net.eval_device(x))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.time()-t0)/10
    fl = tw.flops_per_position if name.startswith("tower") else net.flops_per_position()
    print(f"{name}: {dt*1e3:.3f} ms  {B*fl/dt/1e12:.1f} TFLOP/s")
if os.environ.get("TUNES"):
    from alphafive_amd import tower_hip
    for spec in os.environ["TUNES"].split(";"):
        tower_hip.tune(0, 0); tower_hip.tune(1, 0); tower_hip.tune(2, 0)
        for kv in spec.split(","):
            if kv: tower_hip.tune(int(kv.split(":")[0]), int(kv.split(":")[1]))
        for _ in range(3): tw.forward(B)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): tw.forward(B)
        torch.cuda.synchronize(); dt = (time.time()-t0)/10
        tw.load_nchw(h0); tw.forward(B); o2 = tw.store_nchw(B).float()
        print(f"tune[{spec}]: {dt*1e3:.3f} ms  {B*tw.flops_per_position/dt/1e12:.1f} TF  max|d| {(o2[:n]-r32).abs().max().item():.4f}")
