"""DeviceRandomStack (libaf_replay.so through the C ABI) vs the host RandomStack, which tests/test_host_utils.py pins
against the reference's utils.RandomStack (utils.py:14-146): same seeds -> the same accept/duplicate/evict decisions
and bit-identical get_data batches (boards, weights, values, policies)."""
import random

import numpy as np
import pytest

from alphafive_amd import utils

pytestmark = pytest.mark.gpu


def _episodes(S, n_ep, seed):
    """Random legal-looking episodes in the replay record format (utils.py:127: state, p[S,S], la, v, w)."""
    rng = np.random.RandomState(seed)
    eps = []
    for _ in range(n_ep):
        T = int(rng.randint(9, min(40, S * S)))
        board = np.zeros((S, S), np.int8)
        rec, la = [], None
        w = utils.construct_weights(T, 0.94)
        for t in range(T):
            p = rng.rand(S, S).astype(np.float32)
            p /= p.sum()
            rec.append((utils.board_to_state(board), p, la, float((-1.0) ** (T - t)), w[t]))
            empt = np.argwhere(board == 0)
            a = tuple(int(v) for v in empt[rng.randint(len(empt))])
            board = utils.step(board, a)
            la = a
        result = utils.DRAW if rng.rand() < 0.1 else (utils.BLACK_WIN if T % 2 == 1 else utils.WHITE_WIN)
        eps.append((rec, result))
    return eps


@pytest.mark.parametrize("S, length, n_ep, batch", [(11, 300, 40, 64), (5, 60, 25, 200), (15, 500, 12, 33)])
def test_device_randomstack_matches_host_class(S, length, n_ep, batch, capsys):
    import torch
    from alphafive_amd.replay import DeviceRandomStack
    eps = _episodes(S, n_ep, seed=S)
    outs = []
    for cls in (utils.RandomStack, DeviceRandomStack):
        random.seed(7)
        np.random.seed(7)
        st = cls(S, length) if cls is utils.RandomStack else cls(S, length, device=0)
        assert st.isEmpty()
        log = []
        for k, (rec, res) in enumerate(eps):
            log.append(st.push(rec, res))
            if k % 5 == 4:
                log.append([np.asarray(a.cpu() if torch.is_tensor(a) else a) for a in st.get_data(batch)])
        log.append((st.black_win, st.white_win, list(st.data_len), list(st.result), st._size(), st.is_full()))
        outs.append(log)
        if cls is DeviceRandomStack:
            st.close()
    capsys.readouterr()
    host, dev = outs
    assert len(host) == len(dev)
    n_batches = 0
    for a, b in zip(host, dev):
        if isinstance(a, list):
            n_batches += 1
            for x, y in zip(a, b):
                assert x.shape == y.shape and x.dtype == y.dtype == np.float32
                assert np.array_equal(x, y)
        else:
            assert a == b
    assert n_batches >= 2 and host[-1][4] <= length          # the eviction path ran


def test_device_randomstack_errors_are_loud():
    from alphafive_amd import replay
    st = replay.DeviceRandomStack(5, 10, device=0, max_episode=3)
    rec = [(utils.board_to_state(np.zeros((5, 5), np.int8)), np.full((5, 5), 0.04, np.float32), None, 1.0, np.float32(1.0))] * 17
    with pytest.raises(replay.ReplayError):          # 17 positions into a ring of 10 + 2*3
        st._store(rec)
    with pytest.raises(replay.ReplayError):
        st._drop_front(1)
    with pytest.raises(NotImplementedError):
        st.save()
    st.close()


def test_trainer_consumes_device_batches():
    """train.Trainer.step takes DeviceRandomStack.get_data's tensors directly (no host round trip)."""
    import torch
    from alphafive_amd.network import ResNet
    from alphafive_amd.replay import DeviceRandomStack
    from alphafive_amd.train import Trainer
    random.seed(1)
    np.random.seed(1)
    st = DeviceRandomStack(11, 400, device=0)
    for rec, res in _episodes(11, 12, seed=3):
        st.push(rec, res)
    net = ResNet(11, device="cuda", seed=0)
    tr = Trainer(net.variables, 11, device="cuda")
    boards, weights, values, policies = st.get_data(32)
    assert boards.is_cuda and boards.shape == (32, 3, 11, 11)
    # (f2) on the device, against the oracle: the loss terms of network.py:40-50 on this very batch vs the fp64
    # restatement, and the first Adam update (main.py:38-39 tf.train.AdamOptimizer) vs its closed form
    from alphafive_amd import train
    from oracle import net_fp64
    hb, hw, hv, hp = (t.cpu().numpy() for t in (boards, weights, values, policies))
    ref = net_fp64.loss_terms(net.variables, hb, hp, hv, hw)
    terms = train.loss_terms(tr.params, boards, policies, values, weights)
    for k in ("total", "cross_entropy", "value_loss", "entropy"):
        assert abs(float(terms[k].detach()) - ref[k]) < 2e-5 * max(1.0, abs(ref[k])), k
    names = ["value/fc2/kernel", "policy/fc/bias", "bone/block2_conv2/kernel", "bone/conv1/kernel"]
    before = {n: tr.params[n].detach().clone() for n in names}
    grads = dict(zip(names, torch.autograd.grad(terms["total"], [tr.params[n] for n in names])))
    m = tr.step(boards, weights, values, policies, lr=1e-3)
    assert abs(m["total"] - ref["total"]) < 2e-5 * max(1.0, abs(ref["total"]))
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)           # first step: m = .1 g, v = .001 g^2
    for n in names:
        g = grads[n]
        expect = before[n] - lr_t * (0.1 * g) / ((0.001 * g * g).sqrt() + 1e-8)
        # (the step recomputes the gradient; on the GPU the two backward passes agree to a few ulp, and where |g| is of the
        # order of eps = 1e-8 the update m / (sqrt(v) + eps) amplifies that: the closed form is checked on the well-conditioned
        # elements, the step bound on all)
        got, well = tr.params[n].detach(), g.abs() > 1e-4          # eps / (sqrt(.001) |g|) < 0.4 %: the update no longer depends on g's last bits
        assert well.float().mean().item() > 0.2
        torch.testing.assert_close(got[well], expect[well], rtol=2e-5, atol=3e-7)
        assert (got - before[n]).abs().max().item() <= 1.0001e-3       # every first Adam step is bounded by lr (ill-conditioned elements too)
        assert (tr.params[n].detach() - before[n]).abs().max().item() > 5e-4      # a full-size first step (~1e-3) happened
    st.close()


def test_closed_loop_self_play_replay_train_on_device(capsys):
    """train.train_loop (main.py:57-76) with every stage on the GPU: SelfPlayEngine -> DeviceRandomStack -> Trainer, the
    engine's hand-written evaluator picking up each weight update."""
    import types
    import torch
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    from alphafive_amd.replay import DeviceRandomStack
    from alphafive_amd.train import Trainer, train_loop
    from conftest import make_cfg
    random.seed(3)
    np.random.seed(3)
    S = 6
    cfg = make_cfg(board_size=S, goal=4, simulation_per_step=16, upper_simulation_per_step=24, batch_size=64)
    cfg.get_lr = lambda step: 1e-3
    cfg.ckpt_path = "/tmp/af_closed_loop_ckpt"
    net = ResNet(S, device="cuda", seed=0)
    before = {k: v.copy() for k, v in net.variables.items()}
    sp = SelfPlayEngine(cfg, 64, net.select_backend("hip"), device=0, seed=1)
    stack = DeviceRandomStack(S, 120, device=0)
    tr = Trainer(net.variables, S, device="cuda")
    logs = []
    steps = train_loop(cfg, sp, net, stack, tr, steps=4, log=logs.append)
    capsys.readouterr()
    assert steps >= 4 and len(logs) >= 3 and all("xcross_loss" in s for s in logs)
    assert stack.is_full() and stack._size() <= 120
    assert any(np.abs(net.variables[k] - before[k]).max() > 0 for k in before)     # the weights moved ...
    x = torch.zeros((2, 3, S, S), device="cuda")
    p_hip, v_hip = sp.pv(x) if hasattr(sp, "pv") else net.select_backend("hip")(x)
    p_t, v_t = net.eval_torch(x)
    assert (p_hip - p_t).abs().max().item() < 1e-5                                 # ... and the HIP evaluator has them
    sp.close()
    stack.close()


@pytest.mark.parametrize("S, goal, sims, G, length", [(6, 4, 24, 48, 400), (11, 5, 12, 32, 900)])
def test_device_to_device_hand_off_equals_the_host_path(S, goal, sims, G, length, capsys):
    """SURVEY 8f-1 / main.py:60-61: finished episodes go from the engine's pack kernels straight into the device replay ring
    (af_replay_append_packed: key -> board, policy, last move, value signs, construct_weights row) — the host reads only the
    header.  Same seeds => the same accept / duplicate / evict decisions and bit-identical batches as pushing the 5-tuples of
    a twin engine into the host RandomStack."""
    import torch
    import pseudonet
    from conftest import make_cfg
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.replay import DeviceRandomStack
    cfg = make_cfg(board_size=S, goal=goal, simulation_per_step=sims, upper_simulation_per_step=sims + 8)
    mk = lambda: SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, 321, 4096), device=0, seed=13)
    a, b = mk(), mk()
    host, dev = utils.RandomStack(S, length), DeviceRandomStack(S, length, device=0)
    logs = []
    for which, sp, st in (("host", a, host), ("dev", b, dev)):
        random.seed(5)
        np.random.seed(5)
        log = []
        for rnd in range(14):
            sp.run_ticks(120)
            sp.check()
            if which == "host":
                log += [st.push(rec, res) for rec, res in sp.pop_episodes(64)]
            else:
                log += st.push_packed(sp.post_episodes_device(64), 64, cfg.gamma)
            if st._size() >= 32:
                log.append([np.asarray(t.cpu() if torch.is_tensor(t) else t) for t in st.get_data(48)])
        log.append((st.black_win, st.white_win, list(st.data_len), list(st.result), st._size()))
        logs.append(log)
    dev.check()
    capsys.readouterr()
    h, d = logs
    assert len(h) == len(d) and sum(1 for x in h if x is True) > 10
    nb = 0
    for x, y in zip(h, d):
        if isinstance(x, list):
            nb += 1
            for u, v in zip(x, y):
                assert u.shape == v.shape and u.dtype == v.dtype and np.array_equal(u.view(np.uint32), v.view(np.uint32))
        else:
            assert x == y
    assert nb >= 3 and h[-1][4] <= length
    if S == 6:                                         # short games: the rejection draw (utils.py:80) said no at least once
        assert any(isinstance(x, bool) and not x for x in h)
    a.close()
    b.close()
    dev.close()
