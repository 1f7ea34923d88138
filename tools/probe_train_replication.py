#!/usr/bin/env python
"""Does the whole loop — self-play on the engine, RandomStack, the training step — reproduce what TensorFlow logged while the
reference trained?  main.py:57-76 from the shipped alphaFive-6960 checkpoint with the reference's settings (542 / 642 sims, buffer
12,000, 4 minibatches of 512 per accepted episode, lr 1e-3, gamma 0.94): the scalars of main.py:41-45 that train_loop logs per step
(x_entropy_loss, value_loss, entropy of the 4th minibatch, as main.py:64-69 does) are compared with
tests/golden/tf_train_scalars.npz (the reference's own TensorBoard log around step 6960).  The TF numbers are TRAINING-batch losses
of a net that sees every buffered position ~80 times, so fresh self-play of the same weights scores worse (tools/probe_tf_scalars.py)
and only a run that trains on its own buffer can be compared with them.
env: G (concurrent games, 128), STEPS (1500), LENGTH (12000), SEED."""
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, make_cfg  # noqa: E402
from alphafive_amd import train  # noqa: E402
from alphafive_amd.engine import SelfPlayEngine  # noqa: E402
from alphafive_amd.network import ResNet  # noqa: E402
from alphafive_amd.replay import DeviceRandomStack  # noqa: E402


def main():
    import random
    G = int(os.environ.get("G", 128))
    steps = int(os.environ.get("STEPS", 1500))
    length = int(os.environ.get("LENGTH", 12000))
    seed = int(os.environ.get("SEED", 1))
    cfg = make_cfg(buffer_size=length, get_lr=lambda step: 1e-3, ckpt_path=tempfile.mkdtemp(prefix="af_ckpt_"))
    random.seed(seed)
    np.random.seed(seed)
    net = ResNet(11, device="cuda")
    net.load_npz(os.path.join(GOLDEN, "alphaFive-6960.weights.npz"))
    sp = SelfPlayEngine(cfg, G, net.select_backend("hip"), device=0, seed=seed)
    stack = DeviceRandomStack(11, length, device=0)
    trainer = train.Trainer(net.variables, 11, device="cuda")
    rows, lens = [], []
    t0 = time.time()

    def log(line):
        # "step: %d, xcross_loss: %0.3f, mse: %0.3f, entropy: %0.3f"
        parts = line.replace(",", "").split()
        rows.append((int(parts[1]), float(parts[3]), float(parts[5]), float(parts[7]), stack.data_len[-1], time.time() - t0))
        if len(rows) % 100 == 0:
            r = np.array(rows[-100:])
            print("steps %5d..%5d  x_entropy %.3f  value_loss %.3f  entropy %.3f  episode_len %.2f  (%.0f s)" % (
                r[0, 0], r[-1, 0], r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].mean(), r[-1, 5]), file=sys.__stdout__, flush=True)

    out = sys.stdout
    sys.stdout = open(os.devnull, "w")           # RandomStack.push prints per episode (utils.py:112-115)
    try:
        train.train_loop(cfg, sp, net, stack, trainer, steps, log=log)
    finally:
        sys.stdout = out
    stack.check()
    r = np.array(rows)
    z = np.load(os.path.join(GOLDEN, "tf_train_scalars.npz"))
    m = (z["step"] >= 6860) & (z["step"] <= 6960)
    tail = r[-min(300, len(r) // 2):]
    print("this run, last %d steps: x_entropy %.4f  value_loss %.4f  entropy %.4f  episode_len %.2f" % (
        len(tail), tail[:, 1].mean(), tail[:, 2].mean(), tail[:, 3].mean(), tail[:, 4].mean()))
    print("TF log, steps 6860..6960: x_entropy %.4f  value_loss %.4f  entropy %.4f  episode_len %.2f" % (
        z["x_entropy_loss"][m].mean(), z["value_loss"][m].mean(), z["entropy"][m].mean(), z["episode_len"][m].mean()))
    print("buffer: %d positions, %d episodes; wins black %d white %d; %.0f s" % (
        stack._size(), len(stack.data_len), stack.black_win, stack.white_win, time.time() - t0))
    sp.close()
    stack.close()


if __name__ == "__main__":
    main()
