"""The only numbers in the reference that came out of its TensorFlow graph are the scalars of its TensorBoard log
(main.py:41-45, one event per training step main.py:69; tests/golden/tf_train_scalars.npz, made by tests/golden/make_tf_scalars.py).
They are losses on TRAINING batches of a net that sees each of its 12,000 buffered positions ~80 times, so they cannot be
compared with one forward pass — but they can be compared with the same loop run here: main.py:57-76 from the shipped
alphaFive-6960 weights at the reference's settings (542 / 642 simulations, buffer 12,000, four minibatches of 512 per accepted
episode, lr 1e-3), self-play on the engine + the hand-written net, episodes device-to-device into DeviceRandomStack, the training
step of alphafive_amd/train.py.  A restored net that was not the function TF trained (wrong layout, activation, head) would start at
the log's step-2 level (x_entropy 4.8, value_loss 1.0) and play random-length games; the loop below must instead sit at TF's
step-6960 level within a few hundred steps: x_entropy 2.16, value_loss 0.32, entropy 2.15, accepted-episode length 27.
A statistical pin (tolerances below: several times the spread between seeds, a fraction of the distance to any wrong net), not a
bitwise one — see DESIGN.md §1."""
import os
import random
import sys

import numpy as np
import pytest

from conftest import GOLDEN, make_cfg

TF_WINDOW = (6860, 6960)            # the hundred logged steps that end at the shipped checkpoint
TOL = dict(x_entropy_loss=0.15, value_loss=0.06, entropy=0.17, episode_len=3.5)


def _tf_means():
    z = np.load(os.path.join(GOLDEN, "tf_train_scalars.npz"))
    m = (z["step"] >= TF_WINDOW[0]) & (z["step"] <= TF_WINDOW[1])
    return {k: float(z[k][m].mean()) for k in TOL}, z


def test_tf_scalar_fixture_is_the_log_of_the_shipped_checkpoint():
    means, z = _tf_means()
    assert z["step"][0] == 6400 and z["step"][-1] == 7000 and len(z["step"]) == 601
    assert abs(means["x_entropy_loss"] - 2.1560) < 1e-3 and abs(means["value_loss"] - 0.3171) < 1e-3
    assert abs(means["entropy"] - 2.1480) < 1e-3 and abs(means["episode_len"] - 27.089) < 1e-2
    # total = weighted terms + 4e-5 * L2 (network.py:50): below the unweighted sum because the weights of utils.py:286-296 average < 1
    assert (z["total_loss"] < z["x_entropy_loss"] + 2 * z["value_loss"]).all()


@pytest.mark.gpu
def test_training_loop_reproduces_the_scalars_tensorflow_logged():
    from alphafive_amd import train
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    from alphafive_amd.replay import DeviceRandomStack
    import tempfile
    steps = int(os.environ.get("AF_TF_SCALAR_STEPS", 400))
    cfg = make_cfg(get_lr=lambda step: 1e-3, ckpt_path=tempfile.mkdtemp(prefix="af_ckpt_"))     # config.py:10: 1e-3 below step 7000
    random.seed(1)
    np.random.seed(1)
    net = ResNet(11, device="cuda")
    net.load_npz(os.path.join(GOLDEN, "alphaFive-6960.weights.npz"))
    sp = SelfPlayEngine(cfg, 128, net.select_backend("hip"), device=0, seed=1)
    stack = DeviceRandomStack(11, cfg.buffer_size, device=0)
    trainer = train.Trainer(net.variables, 11, device="cuda")
    rows = []

    def log(line):          # train_loop's line is main.py:71's: "step: %d, xcross_loss: %0.3f, mse: %0.3f, entropy: %0.3f"
        parts = line.replace(",", "").split()
        rows.append((float(parts[3]), float(parts[5]), float(parts[7]), float(stack.data_len[-1])))

    out, sys.stdout = sys.stdout, open(os.devnull, "w")          # RandomStack.push prints per episode (utils.py:112-115)
    try:
        train.train_loop(cfg, sp, net, stack, trainer, steps, log=log)
    finally:
        sys.stdout = out
    stack.check()
    assert stack.is_full() and len(rows) == steps - 1
    tail = np.array(rows[-200:])
    ours = dict(zip(("x_entropy_loss", "value_loss", "entropy", "episode_len"), tail.mean(axis=0)))
    want, _ = _tf_means()
    print("training loop, last 200 of %d steps: %s\nTF log, steps %d..%d: %s" % (steps, ours, TF_WINDOW[0], TF_WINDOW[1], want))
    for k, tol in TOL.items():
        assert abs(ours[k] - want[k]) < tol, (k, ours[k], want[k])
    sp.close()
    stack.close()
