"""A/B of libaf_net.so build variants (tools/build_f16s_variants.sh): for every alphafive_amd/_lib/variants/libaf_net_<name>.so, in
its own process: forward time at B positions (HIP events over N forwards, value branch on the side stream as in production),
per-layer times when LAYERS=1 (value branch on the main stream, events around each launch are not available through the C ABI,
so the whole forward only), accuracy against the fp64 restatement on 48 positions and a determinism / slot-permutation check.

    python tools/probe_f16s_ab.py [name ...]          # default: every variant found
"""
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(REPO, "alphafive_amd", "_lib", "variants")


def child():
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import numpy as np
    import torch
    from alphafive_amd import net_hip
    from alphafive_amd.network import ResNet
    from oracle import net_fp64
    from test_gpu_net import _positions
    B, N, S = int(os.environ.get("B", 4096)), int(os.environ.get("N", 60)), int(os.environ.get("S", 11))
    net = ResNet(S, device="cuda", seed=2)
    if S == 11:
        net.load_npz(os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz"))
    hn = net_hip.HipNet(net.variables, S, B, "cuda")
    x = _positions(S, B, seed=1)
    xb = torch.from_numpy(x).cuda()
    if os.environ.get("BRANCH") is not None:
        net_hip.tune(4, int(os.environ["BRANCH"]))           # value branch on its side stream (1, default) or serialised on the main stream (0)
    if os.environ.get("ABLBITS"):
        net_hip.tune(7, int(os.environ["ABLBITS"]))          # af_conv_f16s ablation / A-B bits (16 = the VALU stem)
    for _ in range(10):
        hn(xb)
    torch.cuda.synchronize()
    times = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            hn(xb)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / N)
    p, v = (t.clone() for t in hn(xb))
    p2, v2 = hn(xb)
    det = bool(torch.equal(p, p2) and torch.equal(v, v2))
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    pp, vp = hn(xb[perm].contiguous())
    slot = bool(torch.equal(pp, p[perm]) and torch.equal(vp, v[perm]))
    # many repetitions compared bit for bit: a wait that is too loose shows up as a rare mismatch
    bad = 0
    for _ in range(int(os.environ.get("REPS", 200))):
        q, w = hn(xb)
        bad += int(not (torch.equal(q, p) and torch.equal(w, v)))
    p64, v64 = net_fp64.forward(net.variables, x[:48])
    import hashlib
    digest = hashlib.blake2b(p.cpu().numpy().tobytes() + v.cpu().numpy().tobytes(), digest_size=8).hexdigest()    # variants: same bits?
    out = {"ms": min(times), "ms_all": times, "digest": digest, "deterministic": det, "slot_independent": slot, "mismatching_repeats": bad,
           "dv": float(np.abs(v[:48].cpu().numpy() - v64).max()), "dp": float(np.abs(p[:48].cpu().numpy() - p64).max())}
    print("RESULT " + json.dumps(out), flush=True)


def main():
    names = sys.argv[1:] or sorted(os.path.basename(f)[len("libaf_net_"):-3] for f in glob.glob(os.path.join(VAR, "libaf_net_*.so")))
    for rnd in range(int(os.environ.get("ROUNDS", 2))):                 # interleaved rounds: box-to-box and thermal drift show up
        for name in names:
            lib = os.path.join(VAR, "libaf_net_%s.so" % name) if name != "default" else os.path.join(REPO, "alphafive_amd", "_lib", "libaf_net.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, AF_NET_LIB=lib),
                               capture_output=True, text=True, timeout=900)
            res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            print("%-28s round %d  %s" % (name, rnd, res[-1][7:] if res else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])), flush=True)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
