"""BASELINE.json full size (configs[1]: 4096 concurrent 11x11 games, 500 sims/move) through the C ABI:
bit-exact visit-count vectors vs the oracle on a sample of games, size-independent invariants on all of
them, run-to-run determinism, and shard invariance (games g..g+n of a big engine == the same games run in
their own engine with first_game_id = g — the single-GPU form of the 8-GPU sharding check)."""
import numpy as np
import pytest

import oracle
import pseudonet
from conftest import make_cfg

pytestmark = pytest.mark.gpu
SALT, PEAK, SEED = 777, 8192, 42


def test_config2_full_size_parity_invariants_and_sharding():
    import torch
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(simulation_per_step=500, upper_simulation_per_step=642)
    G = 4096

    def run(G_, first, ticks=1030):
        sp = SelfPlayEngine(cfg, G_, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED,
                            first_game_id=first)
        sp.run_ticks(ticks)
        sp.check()
        ct = sp.counters()
        dumps = {g: sp.engine.tree_dump(g) for g in (0, G_ - 1)}
        sp.close()
        return ct, dumps, None

    ct, dumps, _ = run(G, 0)
    # size-independent invariants (SURVEY §8a rule 5 and the counters' bookkeeping)
    assert ct["plies"] == 2 * G                       # 1030 ticks: every game committed exactly two moves
    assert ct["sims"] == ct["expands"] + ct["terminals"]
    assert ct["selects"] >= ct["sims"] and ct["episodes"] == 0
    for d in dumps.values():
        # rule 5: sum_n == sum of edge visits, except along the ONE simulation parked at its leaf (select has
        # counted the node, the backup has not happened yet): those nodes are ahead by exactly 1
        ahead = d["sum_n"] - d["n"].sum(1)
        assert ((ahead == 0) | (ahead == 1)).all() and ahead.sum() <= 8
        assert (d["p"] >= 0).all() and np.isfinite(d["w"]).all()
        legal_p = d["p"].sum(1)
        assert np.abs(legal_p - 1.0).max() < 1e-4     # priors renormalised over legal moves
    # bit-exact trees vs the oracle for the sampled games (same seed, game id, pseudo-net)
    for g, d in dumps.items():
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g,
                                  pseudo_salt=SALT, pseudo_peak=PEAK)
        board = np.zeros((11, 11), np.int8)
        state, last = oracle.board_to_state(board), None
        for _ in range(2):                            # the two committed moves
            _, act, _ = orc.get_action(state, last)
            board = oracle.step(oracle.state_to_board(state, 11), act)
            state, last = oracle.board_to_state(board), act
        # continue the third move's simulations the engine has already done: 1030 ticks = 2 moves + the rest
        od = orc.tree_dump()
        omap = {od["keys"][i].tobytes(): i for i in range(len(od["sum_n"]))}
        hits = 0
        for i in range(len(d["sum_n"])):
            k = np.zeros(8, np.uint64)
            k[:2], k[4:6] = d["keys"][i][:2], d["keys"][i][2:]
            j = omap.get(k.tobytes())
            if j is None:
                continue                              # created by the third move's sims the oracle has not run
            # nodes untouched by the in-flight third move must be identical; visited ones only grow
            assert (d["p"][i].view(np.uint32) == od["p"][j].view(np.uint32)).all()
            assert (d["n"][i] >= od["n"][j]).all()
            hits += 1
        assert hits > 500
    # determinism: the same engine configuration reproduces the same counters and trees
    ct2, dumps2, _ = run(G, 0)
    assert ct2 == ct
    for g in dumps:
        for k in ("keys", "sum_n", "n"):
            assert (dumps[g][k] == dumps2[g][k]).all()
        assert (dumps[g]["w"].view(np.uint32) == dumps2[g]["w"].view(np.uint32)).all()
    # shard invariance: games 2048..4095 in their own engine == the same games inside the 4096-game engine
    ct_s, dumps_s, _ = run(2048, 2048)
    big_last, shard_last = dumps[G - 1], dumps_s[2047]
    for k in ("keys", "sum_n", "n"):
        assert (big_last[k] == shard_last[k]).all()
    assert (big_last["w"].view(np.uint32) == shard_last["w"].view(np.uint32)).all()


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
    ct_s, dumps_s = run(1024, 3072)                   # games 3072..4095 in their own engine
    for k in ("keys", "sum_n", "n"):
        assert (dumps[G - 1][k] == dumps_s[1023][k]).all()
    assert (dumps[G - 1]["w"].view(np.uint32) == dumps_s[1023]["w"].view(np.uint32)).all()
