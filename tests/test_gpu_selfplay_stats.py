"""Forward-only statistical pin of SURVEY §8 row Q.  The reference repository holds the bookkeeping of the
replay buffer its TensorFlow net filled by self-play up to the shipped checkpoint (data_buffer/data_len6960.pkl, result6960.pkl ->
tests/golden/selfplay_stats_6960.npz, made by tests/golden/make_selfplay_stats.py): 470 accepted episodes, mean length 25.53
(9..64), black : white = 251 : 219, no draw.  With the 6960 weights FIXED, the engine's self-play at the reference's settings
(542 / 642 simulations, genData/player.py) through the hand-written forward, pushed through RandomStack.push's acceptance /
duplication / eviction rules (utils.py:65-115), must reproduce that distribution — no optimiser in the loop that could re-adapt a
slightly wrong function (what the TF-scalar test goes through).  Two deliberately broken forwards (input planes of
utils.py:256 swapped; the value head's dense layer fed in NHWC instead of NCHW flatten order, network.py:70-72) must fail the
same check.  Its power is limited to errors that change how games unfold: a tanh(x) value head, flipped kernels and even the
policy head's dense layer in NHWC order pass it — 542 simulations with a sound value head play the same game lengths
(calibration: tools/probe_selfplay_stats.py, profiles/r4_05_* and r4_06_*).  Statistical, not bitwise: TF outputs cannot be had in this
image."""
import os
import random
import sys

import numpy as np
import pytest

from conftest import GOLDEN, make_cfg

W = os.path.join(GOLDEN, "alphaFive-6960.weights.npz")
TOL_LEN, TOL_BLACK = 1.5, 0.04


def _fixture():
    z = np.load(os.path.join(GOLDEN, "selfplay_stats_6960.npz"))
    dl, rs = z["data_len"], z["result"]
    return dict(mean_len=float(dl.mean()), black_share=float((rs == 1).mean()), draws=int((rs == 0).sum()), n=len(dl),
                positions=int(dl.sum()), buffer=int(z["buffer_size"]), sims=int(z["simulation_per_step"]),
                upper=int(z["upper_simulation_per_step"]))


def test_selfplay_fixture_is_the_buffer_of_the_shipped_checkpoint():
    f = _fixture()
    assert f["n"] == 470 and f["positions"] == 12000 == f["buffer"] and f["draws"] == 0
    assert abs(f["mean_len"] - 25.532) < 1e-3 and abs(f["black_share"] - 251 / 470) < 1e-9
    assert (f["sims"], f["upper"]) == (542, 642)


def broken_variables(variables, how):
    """Weight-space forms of two forward bugs (the kernels stay untouched): 'tanh' = value head tanh(x) instead of tanh(x/2)
    (value/fc2 scaled by 2); 'nhwc' = policy/fc fed the 1x1-conv output flattened [pixel][channel] instead of [channel][pixel]."""
    v = {k: a.copy() for k, a in variables.items()}
    if how == "tanh":
        v["value/fc2/kernel"] *= 2.0
        v["value/fc2/bias"] *= 2.0
    elif how == "nhwc":
        k = v["policy/fc/kernel"]                      # [16*C, C], row index c*C + pixel (NCHW flatten, network.py:84)
        C = k.shape[1]
        v["policy/fc/kernel"] = np.ascontiguousarray(k.reshape(16, C, C).transpose(1, 0, 2).reshape(16 * C, C))   # row pixel*16 + c
    elif how == "vnhwc":                               # the same flatten-order bug in the value head's fc1
        k = v["value/fc1/kernel"]                      # [4*C, 64]
        C = k.shape[0] // 4
        v["value/fc1/kernel"] = np.ascontiguousarray(k.reshape(4, C, 64).transpose(1, 0, 2).reshape(4 * C, 64))
    elif how == "flip":                                # convolution instead of cross-correlation: every 3x3 / 5x5 kernel flipped
        for name, a in v.items():
            if a.ndim == 4 and a.shape[0] > 1:
                v[name] = np.ascontiguousarray(a[::-1, ::-1])
    elif how == "transpose":                           # HWIO kernels read as WHIO: x and y swapped
        for name, a in v.items():
            if a.ndim == 4 and a.shape[0] > 1:
                v[name] = np.ascontiguousarray(a.transpose(1, 0, 2, 3))
    elif how == "planes":                              # board_to_inputs' first two planes (own / opponent stones) swapped
        k = v["bone/conv1/kernel"].copy()
        k[:, :, [0, 1]] = k[:, :, [1, 0]]
        v["bone/conv1/kernel"] = k
    else:
        raise ValueError(how)
    return v


def selfplay_buffer_stats(variables, G, seed=3, max_steps=400):
    """First complete episode of each of G games (an unbiased sample: every game starts at the empty board with its own noise
    stream) at 542 / 642 simulations -> pushed in game order through DeviceRandomStack (length 12,000) -> statistics of the
    accepted episodes and of the final buffer."""
    from alphafive_amd.engine import SelfPlayEngine, assemble_episode
    from alphafive_amd.network import ResNet
    from alphafive_amd.replay import DeviceRandomStack
    f = _fixture()
    cfg = make_cfg(simulation_per_step=f["sims"], upper_simulation_per_step=f["upper"])
    net = ResNet(11, device="cuda")
    net.set_variables(variables)
    sp = SelfPlayEngine(cfg, G, net.select_backend("hip"), device=0, seed=seed)
    first = {}
    for _ in range(max_steps):
        for _ in range(16):
            sp.run_ticks_graph(16)
        sp.check()
        while True:
            raws = sp.pop_raw(512)
            for r in raws:
                if r["seq"] == 0:
                    first[r["game"]] = r
            if len(raws) < 256:
                break
        if len(first) == G:
            break
    sp.close()
    assert len(first) == G, "only %d of %d games finished" % (len(first), G)
    random.seed(seed)
    np.random.seed(seed)
    stack = DeviceRandomStack(11, f["buffer"], device=0)
    out, sys.stdout = sys.stdout, open(os.devnull, "w")          # RandomStack.push prints per episode (utils.py:112-115)
    draws, snaps = 0, []
    try:
        for g in range(G):
            rec, result = assemble_episode(first[g], 11, cfg.gamma)
            stack.push(rec, result)
            draws += result == 0
            if stack.is_full():                        # one snapshot of the buffer's bookkeeping per push, once it is full
                snaps.append((sum(stack.data_len) / len(stack.data_len), stack.result.count(1) / len(stack.result)))
    finally:
        sys.stdout = out
    stack.check()
    lens = np.array([first[g]["T"] for g in range(G)])
    dl, rs = np.array(stack.data_len), np.array(stack.result)
    snaps = np.array(snaps) if snaps else np.zeros((1, 2))
    # buffer_* = the final buffer (one sample of ~470 episodes, like the reference's file); avg_* = the same two numbers averaged
    # over every state the full buffer went through (G = 2048: ~4 turnovers), the lower-variance estimate the test asserts on
    st = dict(games=G, raw_mean_len=float(lens.mean()), raw_black_share=float((lens % 2 == 1).mean()), draws=int(draws),
              buffer_episodes=len(dl), buffer_mean_len=float(dl.mean()), buffer_black_share=float((rs == 1).mean()),
              buffer_positions=int(dl.sum()), len_min=int(lens.min()), len_max=int(lens.max()),
              avg_mean_len=float(snaps[:, 0].mean()), avg_black_share=float(snaps[:, 1].mean()), snapshots=len(snaps))
    stack.close()
    return st


@pytest.mark.gpu
def test_fixed_6960_weights_reproduce_the_reference_buffer_and_broken_forwards_do_not():
    f = _fixture()
    G = int(os.environ.get("AF_SELFPLAY_STATS_GAMES", 2048))
    with np.load(W) as z:
        variables = {k: z[k] for k in z.files}
    good = selfplay_buffer_stats(variables, G)
    print("reference buffer (TF net, steps ~6490..6960): mean length %.2f, black share %.3f, %d episodes" % (f["mean_len"], f["black_share"], f["n"]))
    print("6960 weights, hand-written forward:", good)
    assert good["buffer_positions"] == f["buffer"] and good["draws"] <= G // 100 and good["snapshots"] > G // 2
    assert abs(good["avg_mean_len"] - f["mean_len"]) < TOL_LEN, (good["avg_mean_len"], f["mean_len"])
    assert abs(good["avg_black_share"] - f["black_share"]) < TOL_BLACK, (good["avg_black_share"], f["black_share"])
    # the check has teeth for layout-class errors, not for everything: measured (profiles/r4_05_selfplay_stats_calibration.txt),
    # a value head with tanh(x) for tanh(x/2) or spatially flipped / transposed kernels (the net was trained on all 8 board
    # symmetries, utils.py:118-146) play games of the same length distribution and pass, and so does a policy head whose dense
    # layer reads NHWC order (25.3-25.5 plies: the search leans on the value head); swapped input planes (86 plies) or the VALUE
    # head's dense layer in the wrong flatten order (28.3 plies) do not
    # ("planes" plays 86-ply games: three times the test's run time for the second demonstration of the same thing — opt-in)
    for how in (("planes", "vnhwc") if os.environ.get("AF_SELFPLAY_STATS_ALL") else ("vnhwc",)):
        bad = selfplay_buffer_stats(broken_variables(variables, how), G // 2)
        print("broken forward (%s):" % how, bad)
        ok = (abs(bad["avg_mean_len"] - f["mean_len"]) < TOL_LEN and abs(bad["avg_black_share"] - f["black_share"]) < TOL_BLACK)
        assert not ok, "the broken forward (%s) passes the check: %s" % (how, bad)
