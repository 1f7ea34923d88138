"""BASELINE.json full sizes through the C ABI, in the steady state the metric is defined on: configs[1]
(4096 concurrent 11x11 games, 500 sims/move) and configs[3] (15x15, 800 sims/move) are run until every game has
finished at least one episode; complete episodes of sampled games bit-exact vs the oracle, size-independent
invariants on all games, and shard invariance (games g..g+n of a big engine == the same games run in their own
engine with first_game_id = g — the single-GPU form of the 8-GPU sharding check)."""
import os

import numpy as np
import pytest

import oracle
import pseudonet
from conftest import make_cfg, run_in_threads

pytestmark = pytest.mark.gpu
SALT, PEAK, SEED = 777, 8192, 42


def _play_until_every_game_finished(sp, G, max_rounds, ticks_per_round=512, cap=512):
    """-> {game: [raw episodes in order]} once every game has finished at least one episode."""
    got = {}
    for _ in range(max_rounds):
        sp.run_ticks(ticks_per_round)
        sp.check()
        while True:
            raws = sp.pop_raw(cap=cap)
            for raw in raws:
                got.setdefault(raw["game"], []).append(raw)
            if len(raws) < cap // 2:
                break
        if len(got) == G:
            break
    return got


def _assert_episode_equals_oracle(raw, orc, S, gamma):
    from alphafive_amd.engine import assemble_episode
    orec, extra = orc.run()
    assert raw["T"] == len(orec), f"game {raw['game']} seq {raw['seq']}: length"
    assert (raw["actions"] == extra["actions"]).all() and (raw["visits"] == extra["visits"]).all()
    assert raw["final_value"] == extra["final_value"]
    rec, _ = assemble_episode(raw, S, gamma)
    for (s, p, la, v, w), (os_, op, ola, ov, ow) in zip(rec, orec):
        assert s == os_ and la == ola and v == ov and w == ow
        assert (p.view(np.uint32) == op.view(np.uint32)).all()


def test_config2_full_size_steady_state_parity_invariants_and_sharding():
    """BASELINE configs[1] at full size and in the steady state SURVEY §8d defines: 4096 games, 11x11, 500 sims/move
    (cap 642), run until EVERY game has finished at least one episode — terminal simulations, store collection,
    episode ends, restarts and the device hand-off all happen at scale (reference: player.py:53-82 run, :73 tree reset).
    Sampled games' complete episodes are compared with the oracle bit for bit; shard invariance: the last 256 games
    in their own engine (first_game_id = 3840) reproduce exactly the episodes they produce inside the big engine."""
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(simulation_per_step=500, upper_simulation_per_step=642)
    G = 4096
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED)
    got = _play_until_every_game_finished(sp, G, max_rounds=200)
    ct = sp.counters()
    hist = sp.engine.tick_histogram()
    sp.close()
    assert len(got) == G, "some games never finished an episode"
    # bookkeeping invariants at scale
    assert ct["sims"] == ct["expands"] + ct["terminals"] and ct["selects"] >= ct["sims"]
    assert ct["terminals"] > 0 and ct["collector_runs"] > G      # every game collected its store several times
    assert ct["stalls"] == 0                                     # the hand-off kept up
    n_eps = sum(len(v) for v in got.values())
    assert ct["episodes"] >= n_eps >= G
    lens = np.array([v[0]["T"] for v in got.values()])
    assert lens.min() >= 9 and lens.max() <= 121
    for g, eps in got.items():                                   # per game: sequence numbers 0,1,2,... in order
        assert [e["seq"] for e in eps] == list(range(len(eps)))
    assert hist["selects"][9:].sum() > 0                         # deep simulations happened; the per-launch budget bounds terminal chains
    # bit-exact complete episodes vs the oracle: 8 sampled games, every episode each of them finished
    def replay(g):                                               # (one host thread per sampled game)
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g,
                                  pseudo_salt=SALT, pseudo_peak=PEAK)
        for raw in got[g][:2 if g in (0, G - 1) else 1]:
            _assert_episode_equals_oracle(raw, orc, 11, cfg.gamma)

    run_in_threads(replay, (0, 1, 777, 1234, 2048, 3000, 4000, G - 1))
    # shard invariance (the single-GPU form of SURVEY §8e's check: an N-GPU run = N independent shards)
    sp2 = SelfPlayEngine(cfg, 256, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED, first_game_id=G - 256)
    got2 = _play_until_every_game_finished(sp2, 256, max_rounds=200)
    sp2.close()
    for g2, eps2 in got2.items():
        a, b = got[G - 256 + g2][0], eps2[0]
        assert a["T"] == b["T"] and (a["actions"] == b["actions"]).all() and (a["visits"] == b["visits"]).all()
        assert (a["policies"].view(np.uint32) == b["policies"].view(np.uint32)).all() and (a["keys"] == b["keys"]).all()


def test_config4_15x15_steady_state_episodes_match_oracle():
    """BASELINE configs[3] search settings (15x15, 800 sims/move, cap 942; 4-word bitboards) on a 256-game engine until
    every game has finished an episode; sampled games bit-exact against the oracle."""
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=15, simulation_per_step=800, upper_simulation_per_step=942)
    G = 256
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED)
    got = _play_until_every_game_finished(sp, G, max_rounds=400, cap=256)
    ct = sp.counters()
    sp.close()
    assert len(got) == G and ct["collector_runs"] > 0 and ct["stalls"] == 0
    def replay(g):
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g,
                                  pseudo_salt=SALT, pseudo_peak=PEAK)
        _assert_episode_equals_oracle(got[g][0], orc, 15, cfg.gamma)

    run_in_threads(replay, (0, 100, G - 1))


def test_config4_full_size_15x15_invariants_oracle_and_sharding():
    """BASELINE configs[3] shape: 4096 games on 15x15, 800 sims/move (cap 942), KW = 4 bitboard words per colour.
    Forced root visits alone take 2 * 225 = 450 of the 800 simulations at the empty board."""
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=15, simulation_per_step=800, upper_simulation_per_step=942)
    G, S = 4096, 15

    def run(G_, first, ticks=830):
        sp = SelfPlayEngine(cfg, G_, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED,
                            first_game_id=first)
        sp.run_ticks(ticks)
        sp.check()
        ct = sp.counters()
        dumps = {g: sp.engine.tree_dump(g) for g in (0, G_ - 1)}
        sp.close()
        return ct, dumps

    ct, dumps = run(G, 0)
    assert ct["plies"] == G                           # 830 ticks: every game committed exactly one move
    assert ct["sims"] == ct["expands"] + ct["terminals"] and ct["episodes"] == 0
    for g, d in dumps.items():
        ahead = d["sum_n"] - d["n"].sum(1)
        assert ((ahead == 0) | (ahead == 1)).all() and ahead.sum() <= 8
        assert np.abs(d["p"].sum(1) - 1.0).max() < 1e-4 and np.isfinite(d["w"]).all()
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g,
                                  pseudo_salt=SALT, pseudo_peak=PEAK)
        state = oracle.board_to_state(np.zeros((S, S), np.int8))
        orc.get_action(state, None)                   # the one committed move
        od = orc.tree_dump()
        omap = {od["keys"][i].tobytes(): i for i in range(len(od["sum_n"]))}
        hits = 0
        for i in range(len(d["sum_n"])):
            j = omap.get(np.ascontiguousarray(d["keys"][i]).tobytes())
            if j is None:
                continue                              # created by the second move's simulations already under way
            assert (d["p"][i].view(np.uint32) == od["p"][j].view(np.uint32)).all()
            assert (d["n"][i] >= od["n"][j]).all()
            hits += 1
        assert hits > 300
    # ... and EXACTLY at the one-move point (the comparison above can only say ">=" because the second move's
    # simulations are under way in SELFPLAY mode): the same 4096 games as an EXTERNAL-mode engine, which holds a game at MOVE_DONE,
    # so every tree is frozen the moment its move is committed — sampled games' whole trees (N, W bits, P bits, sum_n, dtype
    # flags), visit vectors and moves against the oracle's after one get_action (player.py:128-147)
    import torch
    from alphafive_amd.engine import Engine, MODE_EXTERNAL, STATUS_MOVE_DONE, state_to_key
    from test_gpu_parity import _compare_tree
    ext = Engine(cfg, G, device=0, mode=MODE_EXTERNAL, training=True, seed=SEED, node_cap=4 * 800 + 64)   # (SELFPLAY's default; EXTERNAL's 32768 is sized for one game)
    empty = oracle.board_to_state(np.zeros((S, S), np.int8))
    ext.set_roots(np.arange(G), np.stack([state_to_key(empty, S)] * G), [-1] * G)
    pol, val = torch.zeros((G, S * S), device="cuda"), torch.zeros((G,), device="cuda")
    planes = torch.zeros((G, 3, S, S), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for t in range(1200):
        ext.tick(pol.data_ptr(), val.data_ptr(), planes.data_ptr(), st)
        p_, v_ = pseudonet.pseudonet_torch(planes, SALT, PEAK)
        pol.copy_(p_), val.copy_(v_)
        if t >= 800 and t % 16 == 0 and (ext.status(st) == STATUS_MOVE_DONE).all():
            break
    assert (ext.status(st) == STATUS_MOVE_DONE).all()
    for g in (0, 1, 1234, 2047, G - 1):
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g,
                                  pseudo_salt=SALT, pseudo_peak=PEAK)
        opol, oact, ovis = orc.get_action(empty, None)
        act, pol_g, vis, _ = ext.move_result(g)
        assert (vis == ovis).all() and (act // S, act % S) == oact
        assert (pol_g.view(np.uint32) == opol.reshape(-1).view(np.uint32)).all()
        seen, total = _compare_tree(ext.tree_dump(g), orc, S)
        assert seen == total > 300
    ext.close()
    ct_s, dumps_s = run(1024, 3072)                   # games 3072..4095 in their own engine
    for k in ("keys", "sum_n", "n"):
        assert (dumps[G - 1][k] == dumps_s[1023][k]).all()
    assert (dumps[G - 1]["w"].view(np.uint32) == dumps_s[1023]["w"].view(np.uint32)).all()


def test_every_game_of_a_512_game_engine_equals_the_oracle():
    """not a sample — every game of a 512-game engine at configs[1]'s settings (11x11, 500 / 642), first_game_id
    3584 (the last 512 ids of the 4096-game bench engine; a game's tree depends on its id only), each first episode (and, for
    every 8th game, the second one too if it exists: an episode that starts from a restarted store) replayed by the C oracle on
    the host cores (tools/parity_sweep_fullsize.py; the 4096-game run of the same sweep, two episodes of every game, is
    profiles/r5_40)."""
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import parity_sweep_fullsize as sweep
    rep = sweep.sweep(G=512, board=11, sims=500, upper=642, memo=False, first_game_id=3584, second_every=8)
    print(rep)
    assert rep["mismatches"] == 0, rep["first_mismatches"]
    assert rep["episodes_compared_with_the_oracle"] >= 512 and rep["plies_compared"] > 512 * 9
