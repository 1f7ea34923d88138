"""The evaluation memo (include/af_engine.h ABI v5): games of one engine share the net's evaluations of positions with few
stones.  The reference gives every worker its own tree (genData/player.py:38) and asks the net for every unseen leaf
(player.py:186-197); the memo only answers such a request at once when ANY game has asked it before — the answer is the same
bits (the forward is batch-slot and batch-size independent, tests/test_gpu_net.py), so every tree, every visit count and every
recorded episode must be what the run without the memo produces: compared here bit for bit, and against the oracle."""
import os

import numpy as np
import pytest

import oracle
import pseudonet
from conftest import GOLDEN, make_cfg
from test_gpu_graph_loop import _drain, _same_episodes

pytestmark = pytest.mark.gpu
W = os.path.join(GOLDEN, "alphaFive-6960.weights.npz")


def _run_until(sp, plies, graph=False, chunk=64, episodes=0):
    got = []
    while sp.progress()[0] < plies or sp.progress()[1] < episodes:
        if graph:
            for _ in range(chunk // 16):
                sp.run_ticks_graph(16)
        else:
            sp.run_ticks(chunk)
        sp.check()
        got += _drain(sp)
    sp.check()
    got += _drain(sp)
    return got


def _common(a, b):
    by = {(e["game"], e["seq"]): e for e in b}
    ca = [e for e in a if (e["game"], e["seq"]) in by]
    return ca, [by[(e["game"], e["seq"])] for e in ca]


def test_memo_run_equals_plain_run_and_the_oracle_on_the_pseudo_net():
    from alphafive_amd.engine import SelfPlayEngine, EngineError
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=60, upper_simulation_per_step=80)
    salt, peak, seed, G = 11, 16384, 3, 64
    pv = lambda x: pseudonet.pseudonet_torch(x, salt, peak)                  # noqa: E731
    with pytest.raises(EngineError, match="weights_version"):                # no way to tell when to forget: refused
        SelfPlayEngine(cfg, 4, pv, device=0, seed=seed, eval_memo=True)
    a = SelfPlayEngine(cfg, G, pv, device=0, seed=seed, weights_version=0, eval_memo=dict(log2_buckets=8, max_stones=6))
    b = SelfPlayEngine(cfg, G, pv, device=0, seed=seed)
    got_a = _run_until(a, 40 * G)
    got_b = _run_until(b, 40 * G)
    st = a.engine.memo_stats()
    assert st["entries"] == 4 << 8 and st["hits"] > 1000 and st["replaced"] > 0 and st["probes"] >= st["hits"]
    assert a.ticks < b.ticks                                                  # the same plies in fewer launches
    ca, cb = _common(got_a, got_b)
    assert len(ca) >= G
    _same_episodes(ca, cb)
    ct_a, ct_b = a.counters(), b.counters()
    assert ct_a["expands"] > 0 and st["hits"] < ct_a["expands"]
    for e in [x for x in got_a if x["seq"] <= 1][:10]:
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=e["game"],
                                  pseudo_salt=salt, pseudo_peak=peak)
        for _ in range(e["seq"] + 1):
            orec, extra = orc.run()
        assert e["T"] == len(orec) and (e["visits"] == extra["visits"]).all() and (e["actions"] == extra["actions"]).all()
    a.close()
    b.close()


def test_memo_with_the_hand_written_net_graph_loop_and_15x15():
    """alphaFive-6960 through af_conv_f16s, eager and inside the HIP graph (the insert kernel is captured with the forward); a
    small table so that entries are replaced all the time; 15x15 keys (4 words per bitboard) on the random-init net."""
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    for S, sims, G, plies in ((11, 40, 128, 24), (15, 24, 64, 10)):
        cfg = make_cfg(board_size=S, simulation_per_step=sims, upper_simulation_per_step=sims + 20)
        nets = [ResNet(S, device="cuda", seed=0) for _ in range(3)]
        if S == 11:
            for nt in nets:
                nt.load_npz(W)
        memo = dict(log2_buckets=7, max_stones=7)
        a = SelfPlayEngine(cfg, G, nets[0].select_backend("hip"), device=0, seed=21, eval_memo=memo)
        g = SelfPlayEngine(cfg, G, nets[1].select_backend("hip"), device=0, seed=21, eval_memo=memo)
        b = SelfPlayEngine(cfg, G, nets[2].select_backend("hip"), device=0, seed=21)
        got_a, got_g, got_b = (_run_until(sp, plies * G, graph=sp is g, episodes=G) for sp in (a, g, b))
        assert g._graph is not None and g._graph[0][1][-1] == (7, 7)
        for sp in (a, g):
            st = sp.engine.memo_stats()
            assert st["hits"] > 200 and st["replaced"] > 0, st
        assert a.ticks <= b.ticks
        for got in (got_a, got_g):
            ca, cb = _common(got, got_b)
            assert len(ca) >= G // 4
            _same_episodes(ca, cb)
        for sp in (a, g, b):
            sp.close()


def test_new_weights_empty_the_memo():
    """Episodes that start after a weight change must be the new net's from their first simulation on: the memo is full of the
    old net's evaluations of exactly the opening positions such an episode asks for first."""
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet, random_variables
    cfg = make_cfg(simulation_per_step=24, upper_simulation_per_step=30)
    G = 48
    na, nb = ResNet(11, device="cuda"), ResNet(11, device="cuda")
    na.load_npz(W)
    nb.set_variables(random_variables(11, seed=5))
    a = SelfPlayEngine(cfg, G, na.select_backend("hip"), device=0, seed=2, eval_memo=dict(log2_buckets=12, max_stones=8))
    b = SelfPlayEngine(cfg, G, nb.select_backend("hip"), device=0, seed=2)
    old = _run_until(a, 30 * G)                                        # W1: games are somewhere in their 2nd .. 3rd episode
    assert a.engine.memo_stats()["hits"] > 500
    na.set_variables(random_variables(11, seed=5))
    started = {gm: 0 for gm in range(G)}                               # the episode every game was in at the switch:
    for e in old:                                                      # (everything finished has been drained)
        started[e["game"]] = max(started[e["game"]], e["seq"] + 1)
    new, after = [], []
    while len(after) < G // 2:
        new += _run_until(a, a.progress()[0] + 40 * G)
        after = [e for e in new if e["seq"] > started[e["game"]]]      # begun after the switch: all of it is the new net's
        assert a.ticks < 400000
    need = max(e["seq"] for e in after)
    got_b = []
    while True:
        got_b += _run_until(b, b.progress()[0] + 40 * G)
        have = {(e["game"], e["seq"]) for e in got_b}
        if all((e["game"], e["seq"]) in have for e in after):
            break
        assert b.ticks < 400000 and need < 50
    ca, cb = _common(after, got_b)
    assert len(ca) == len(after)
    _same_episodes(ca, cb)
    a.close()
    b.close()


def test_memo_at_the_metric_settings_complete_episodes_equal_the_oracle():
    """configs[1]'s search settings (11x11, 500 sims/move, cap 642) on 1024 games with the memo on: every game finishes an
    episode (terminal simulations, store collection, restarts, hit streaks at the start of every episode), sampled games'
    complete episodes bit for bit vs the oracle, and the bookkeeping invariants with the memo's share of the expansions."""
    from alphafive_amd.engine import SelfPlayEngine
    from test_gpu_fullsize import SALT, PEAK, SEED, _assert_episode_equals_oracle, _play_until_every_game_finished
    cfg = make_cfg(simulation_per_step=500, upper_simulation_per_step=642)
    G = 1024
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED, weights_version=0,
                        eval_memo=dict(log2_buckets=14, max_stones=5))
    got = _play_until_every_game_finished(sp, G, max_rounds=200)
    ct, st = sp.counters(), sp.engine.memo_stats()
    sp.close()
    assert len(got) == G
    assert ct["sims"] == ct["expands"] + ct["terminals"] and ct["stalls"] == 0
    assert 0.03 * ct["expands"] < st["hits"] < 0.5 * ct["expands"] and st["inserts"] > 0     # a real share, far from all
    for g in (0, 1, 333, 777, G - 1):
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g,
                                  pseudo_salt=SALT, pseudo_peak=PEAK)
        for raw in got[g][:2 if g == 0 else 1]:
            _assert_episode_equals_oracle(raw, orc, 11, cfg.gamma)


def test_bench_reports_the_memo_leg_keys():
    """`bench.py --eval-memo L:K` (what the default run starts as its config2_memo leg): the line says what the memo did in the timed
    region of rank 0's engine, and the headline keys keep their meaning."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--games", "512", "--steps", "2", "--warmup", "0",
                        "--eval-memo", "12:5", "--no-cpu-baseline", "--no-pmc", "--no-extra-configs"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    m = d["config"]["eval_memo"]
    assert m["log2_buckets"] == 12 and m["max_stones"] == 5 and m["entries"] == 4 << 12
    assert m["probes"] >= m["hits"] > 0 and m["inserts"] > 0 and 0 < m["hits_per_simulation"] < 1
    assert d["unit"] == "moves/s" and d["config"]["games_per_gpu"] == 512 and "power" in d["roofline"]


def test_memo_on_a_16x16_board_keeps_the_value_out_of_the_policy_row():
    """on 16x16 a policy row has no spare slot (C == 256 == the row's stride), the value used to be looked for
    at row[C] = the next entry's first prior.  The value lives in an array of its own now: memo run == plain run == oracle."""
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=16, goal=5, simulation_per_step=40, upper_simulation_per_step=50)
    salt, peak, seed, G = 5, 16384, 8, 32
    pv = lambda x: pseudonet.pseudonet_torch(x, salt, peak)                  # noqa: E731
    a = SelfPlayEngine(cfg, G, pv, device=0, seed=seed, weights_version=0, eval_memo=dict(log2_buckets=6, max_stones=6))
    b = SelfPlayEngine(cfg, G, pv, device=0, seed=seed)
    got_a = _run_until(a, 12 * G)
    got_b = _run_until(b, 12 * G)
    st = a.engine.memo_stats()
    assert st["hits"] > 200 and st["replaced"] > 0
    for g in (0, G - 1):                                                     # whole trees, mid-game
        ta, tb = a.engine.tree_dump(g), b.engine.tree_dump(g)
        na, nb = {k.tobytes(): i for i, k in enumerate(ta["keys"])}, {k.tobytes(): i for i, k in enumerate(tb["keys"])}
        common = [k for k in na if k in nb]
        assert len(common) > 50
        for k in common:
            i, j = na[k], nb[k]
            assert (ta["p"][i].view(np.uint32) == tb["p"][j].view(np.uint32)).all()
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=3, pseudo_salt=salt, pseudo_peak=peak)
    state, last = oracle.board_to_state(np.zeros((16, 16), np.int8)), None
    # the committed plies of game 3 so far are in its record buffer only once the episode ends; compare through move results of a
    # fresh pair of engines instead: first 3 moves of game 3, visit counts bit for bit
    a.close(), b.close()
    from alphafive_amd.engine import Engine, MODE_EXTERNAL, state_to_key
    import torch
    ext = Engine(cfg, 4, device=0, mode=MODE_EXTERNAL, training=True, seed=seed, first_game_id=0)   # game 3 = the oracle's; 0..2 fill the memo
    ext.memo_enable(6, 6)
    pol = torch.zeros((4, 256), device="cuda")
    val = torch.zeros((4,), device="cuda")
    planes = torch.zeros((4, 3, 16, 16), device="cuda")
    st_ = torch.cuda.current_stream().cuda_stream
    for ply in range(3):
        key, lc = state_to_key(state, 16), -1 if last is None else last[0] * 16 + last[1]
        ext.set_roots([0, 1, 2, 3], np.stack([key] * 4), [lc] * 4)
        for _ in range(200):
            ext.tick(pol.data_ptr(), val.data_ptr(), planes.data_ptr(), st_)
            p, v = pv(planes)
            pol.copy_(p), val.copy_(v)
            ext.memo_insert(pol.data_ptr(), val.data_ptr(), st_)
            if (ext.status(st_) == 2).all():
                break
        act, _, vis, _ = ext.move_result(3)
        opol, oact, ovis = orc.get_action(state, last)
        assert (vis == ovis).all() and (act // 16, act % 16) == oact, ply
        board = oracle.step(oracle.state_to_board(state, 16), oact)
        state, last = oracle.board_to_state(board), oact
    assert ext.memo_stats()["hits"] > 0                                       # transpositions of the first moves came from the memo
    ext.close()


def test_memo_clear_is_an_epoch_bump_and_the_tick_budget_keeps_the_memo_budget_below_it():
    """ADVICE r5 (low x2): af_engine_memo_clear is O(1) (entries carry the epoch they were written in; the epoch is part of
    params_key so a captured graph goes with it), and af_engine_set_tick_budget re-derives the memo's own yield budget."""
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=30, upper_simulation_per_step=40)
    salt, peak, seed, G = 3, 16384, 2, 32
    ver = [0]
    a = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, salt, peak), device=0, seed=seed,
                       weights_version=lambda: ver[0], eval_memo=dict(log2_buckets=6, max_stones=6))
    b = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, salt, peak), device=0, seed=seed)
    k0 = a.engine.params_key()
    got_a = _run_until(a, 20 * G, graph=True)
    h0 = a.engine.memo_stats()
    assert h0["hits"] > 100
    ver[0] = 1                                            # "new weights" (the same function here: the trees must not change)
    got_a += _run_until(a, 40 * G, graph=True)
    k1 = a.engine.params_key()
    assert k1 != k0 and a.engine._params["memo_epoch"] == 2
    h1 = a.engine.memo_stats()
    assert h1["inserts"] > h0["inserts"] and h1["hits"] > h0["hits"]          # refilled under the new epoch, and hit again
    a.engine.set_tick_budget(3)                           # below 6: the memo's yield budget follows it down
    got_a += _run_until(a, 60 * G, graph=True)
    got_b = _run_until(b, 60 * G)
    ca, cb = _common(got_a, got_b)
    assert len(ca) >= G
    _same_episodes(ca, cb)
    a.close(), b.close()


def test_eager_ticks_do_not_need_a_weight_version_but_graph_replay_does():
    """an evaluator with bind_outputs and no version source is fine for eager tick(); run_ticks_graph refuses it."""
    from alphafive_amd.engine import SelfPlayEngine, EngineError
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=20, upper_simulation_per_step=30)

    class Stub:
        def __call__(self, x):
            return pseudonet.pseudonet_torch(x, 1, 0)

        def bind_outputs(self, p, v):
            pass
    sp = SelfPlayEngine(cfg, 8, Stub(), device=0, seed=1)
    sp.run_ticks(30)
    sp.check()
    assert sp.counters()["plies"] >= 8
    with pytest.raises(EngineError, match="weights_version"):
        sp.run_ticks_graph(4)
    sp.close()
    from alphafive_amd.network import ResNet
    from alphafive_amd.net_hip import HipNet
    net = ResNet(6, device="cuda", seed=2)
    h = HipNet(net.variables, 6, 8, net.device)           # a raw handle: counts its own load() calls
    v0 = h.weights_version()
    sp = SelfPlayEngine(cfg, 8, h, device=0, seed=1)
    sp.run_ticks_graph(4)
    k0 = sp._graph[0]
    h.load(net.variables)
    assert h.weights_version() == v0 + 1
    sp.run_ticks_graph(4)
    assert sp._graph[0] != k0
    sp.close()
