"""Training step (alphafive_amd.train) on torch-CPU: loss vs the fp64 restatement of network.py:40-50,
TF-style Adam vs a numpy restatement, and a few steps actually reduce the loss; checkpoints round-trip."""
import numpy as np
import torch

from alphafive_amd import train
from alphafive_amd.network import ResNet
from oracle import net_fp64


def _batch(S, B, seed=0):
    rng = np.random.RandomState(seed)
    boards = (rng.rand(B, 3, S, S) < 0.25).astype(np.float32)
    pol = rng.rand(B, S * S).astype(np.float32)
    pol /= pol.sum(1, keepdims=True)
    winner = rng.choice([-1.0, 1.0], size=B).astype(np.float32)
    weights = (0.5 + rng.rand(B)).astype(np.float32)
    return boards, weights, winner, pol


def test_loss_matches_fp64_restatement():
    net = ResNet(7, device="cpu", seed=2)
    tr = train.Trainer(net.variables, 7, device="cpu")
    boards, weights, winner, pol = _batch(7, 6)
    t = train.loss_terms(tr.params, *(torch.from_numpy(a) for a in (boards, pol, winner, weights)))
    ref = net_fp64.loss_terms(net.variables, boards, pol, winner, weights)
    for k in ("total", "cross_entropy", "value_loss", "entropy"):
        assert abs(float(t[k]) - ref[k]) < 2e-5 * max(1.0, abs(ref[k])), k


def test_adam_update_follows_tf_semantics_and_learns():
    net = ResNet(6, device="cpu", seed=4)
    tr = train.Trainer(net.variables, 6, device="cpu")
    boards, weights, winner, pol = _batch(6, 32, seed=1)
    name = "value/fc2/kernel"
    p0 = tr.params[name].detach().numpy().copy()
    terms = train.loss_terms(tr.params, *(torch.from_numpy(a) for a in (boards, pol, winner, weights)))
    g = torch.autograd.grad(terms["total"], tr.params[name])[0].numpy()
    m0 = tr.step(boards, weights, winner, pol, lr=1e-3)
    # tf.train.AdamOptimizer, first step: m = .1 g, v = .001 g^2, lr_t = lr*sqrt(1-.999)/(1-.9)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    expect = p0 - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    np.testing.assert_allclose(tr.params[name].detach().numpy(), expect, rtol=2e-5, atol=1e-8)
    last = m0
    for _ in range(15):
        last = tr.step(boards, weights, winner, pol, lr=1e-3)
    assert last["total"] < m0["total"] - 0.05


def test_trainer_checkpoint_is_loadable_by_the_resnet_restore_path(tmp_path):
    net = ResNet(6, device="cpu", seed=4)
    tr = train.Trainer(net.variables, 6, device="cpu")
    boards, weights, winner, pol = _batch(6, 8)
    tr.step(boards, weights, winner, pol, lr=1e-3)
    tr.save(str(tmp_path), 60)
    net2 = ResNet(6, device="cpu", seed=0)
    net2.restore(str(tmp_path))
    for k, v in tr.variables().items():
        assert (net2.variables[k] == v).all()
