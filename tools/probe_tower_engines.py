"""A/B of the bf16 tower's convolution kernels (af_tower_tune key 3): 0 = af_tower_conv, 3 = conv3 for a block's first convolution (default), 4 = af_tower_persist (one launch), 2 = af_tower_conv3 (epilogue under the other
tile pair's MFMAs).  8 blocks, B positions: outputs compared with each other and with the fp32 reference at B = 300, then the
time of a tower pass at B = 8192 (interleaved rounds).  usage: python tools/probe_tower_engines.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import torch.nn.functional as F
from alphafive_amd import tower_hip
from alphafive_amd.network_deep import DeepResNet

ENG = [int(x) for x in os.environ.get("ENGINES", "0,2").split(",")]
net = DeepResNet(11, blocks=8, width=128, device="cuda", seed=3)
g = torch.Generator().manual_seed(5)
for blk in net.tower:
    for k in ("res", "c1", "c2"):
        blk[k] = (blk[k][0], (torch.randn(128, generator=g) * 0.1).to("cuda", torch.bfloat16))
B = 300
net.select_backend("hip", 8192)
tw = net._tower
h0 = (torch.randn((B, 128, 11, 11), generator=g) * 0.5).to("cuda", torch.bfloat16)
ref = h0.float()
for blk in net.tower:
    w = {k: (blk[k][0].float(), blk[k][1].float()) for k in blk}
    mid = F.elu(F.conv2d(ref, *w["c1"], padding=1)).bfloat16().float()
    ref = F.elu(F.conv2d(ref, *w["res"]) + F.conv2d(mid, *w["c2"], padding=1)).bfloat16().float()
outs = {}
for e in ENG:
    tower_hip.tune(3, e)
    tw.load_nchw(h0); tw.forward(B)
    outs[e] = tw.store_nchw(B).float().clone()
    tw.load_nchw(h0); tw.forward(B)
    rep = torch.equal(tw.store_nchw(B).float(), outs[e])
    err = (outs[e] - ref).abs()
    S = 11
    zero_ok = all(float(buf[:B, :, :S].float().abs().sum()) == 0.0 and float(buf[:B, :, S + S * S:].float().abs().sum()) == 0.0 for buf in (tw.x, tw.g))
    print("engine %d: err vs fp32 reference mean %.3e max %.3e | repeatable %s | zero borders intact %s | finite %s" % (
        e, err.mean().item(), err.max().item(), rep, zero_ok, bool(torch.isfinite(outs[e]).all())))
for e in ENG[1:]:
    d = (outs[ENG[0]] - outs[e]).abs()
    print("engine %d vs %d: max |diff| %.3e, identical elements %.2f %%" % (ENG[0], e, d.max().item(), 100.0 * (d == 0).float().mean().item()))
B = 8192
hb = (torch.randn((B, 128, 11, 11), generator=g) * 0.5).to("cuda", torch.bfloat16)
tw.load_nchw(hb)
res = {e: [] for e in ENG}
for rnd in range(int(os.environ.get("ROUNDS", 4))):
    for e in ENG:
        tower_hip.tune(3, e)
        for _ in range(3):
            tw.forward(B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            tw.forward(B)
        e1.record(); torch.cuda.synchronize()
        res[e].append(e0.elapsed_time(e1) / 10)
flop = 2 * 8 * (128 * 128 * 9 * 2 + 128 * 128) * 121 * B
for e in ENG:
    ms = min(res[e])
    print("engine %d: tower pass %.3f ms (rounds: %s) = %.0f TFLOP/s = %.3f of the 2.5 PFLOP/s bf16 peak" % (e, ms, " ".join("%.3f" % x for x in res[e]), flop / ms / 1e9, flop / ms / 1e9 / 2500))
tower_hip.tune(3, 3)
