"""Tier B (GPU box): the HIP engine, called through the C ABI, against the C oracle in
AFO_RNG_PHILOX mode on the same seeded inputs.  Bit-exact: visit-count vectors, actions,
temperature policies, tau, tree contents (N, W, dtype flag, P, sum_n), episode records."""
import numpy as np
import pytest

import oracle
import pseudonet
from conftest import run_in_threads, make_cfg

pytestmark = pytest.mark.gpu


def _cell(S, a):
    return None if a is None else a[0] * S + a[1]


def _compare_tree(dev_dump, orc, S, root_state=None):
    """Every device node equals the oracle's node; every oracle node that is still reachable
    (stones superset of the root's) is present on the device."""
    from alphafive_amd import engine as eng
    od = orc.tree_dump()
    omap = {od["keys"][i].tobytes(): i for i in range(len(od["sum_n"]))}
    kw2 = dev_dump["keys"].shape[1]
    seen = set()
    for i in range(len(dev_dump["sum_n"])):
        k = np.zeros(8, np.uint64)
        kw = kw2 // 2
        k[:kw] = dev_dump["keys"][i][:kw]
        k[4:4 + kw] = dev_dump["keys"][i][kw:]
        j = omap.get(k.tobytes())
        assert j is not None, f"device node {i} unknown to the oracle: {eng.key_to_state(dev_dump['keys'][i], S)}"
        seen.add(j)
        assert dev_dump["sum_n"][i] == od["sum_n"][j]
        assert (dev_dump["n"][i] == od["n"][j]).all()
        if "w64" in od:                           # pipe variant: W is an fp64 running sum
            assert dev_dump["w"].dtype == np.float64 and (dev_dump["w"][i] == od["w64"][j]).all()
        else:
            assert (dev_dump["w"][i].view(np.uint32) == od["w"][j].view(np.uint32)).all()
        assert (dev_dump["p"][i].view(np.uint32) == od["p"][j].view(np.uint32)).all()
        assert (dev_dump["f32"][i] == od["f32"][j]).all()
    if root_state is None:
        assert len(seen) == len(omap)
    return len(seen), len(omap)


@pytest.mark.parametrize("S,goal,sims,upper,training,seed,salt,peak,random_a", [
    (6, 4, 120, 160, True, 3, 1237, 16384, False),
    (7, 4, 50, 70, True, 1, 1235, 4096, False),
    (11, 5, 60, 80, True, 0, 1234, 0, False),
    (11, 5, 300, 400, True, 5, 99, 8192, False),
    (11, 5, 80, 100, False, 2, 1236, 8192, False),
    (7, 4, 60, 80, False, 6, 78, 4096, True),
    (15, 5, 40, 60, True, 8, 5, 8192, False),
])
def test_player_get_action_matches_oracle(S, goal, sims, upper, training, seed, salt, peak, random_a):
    from alphafive_amd.player import Player
    from alphafive_amd import utils
    cfg = make_cfg(board_size=S, goal=goal, simulation_per_step=sims, upper_simulation_per_step=upper)
    pl = Player(cfg, training=training, pv_fn=lambda x: pseudonet.pseudonet_np(x, salt, peak), seed=seed, game_id=7)
    orc = oracle.OraclePlayer(cfg, training=training, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=7,
                              pseudo_salt=salt, pseudo_peak=peak)
    state, last, over, ply = pl.get_init_state(), None, False, 0
    max_plies = 12 if S >= 11 else 60
    while not over and ply < max_plies:
        pol, act = pl.get_action(state, last_action=last, random_a=random_a)
        opol, oact, ovis = orc.get_action(state, last, random_a)
        assert (pl.last_visits == ovis).all(), f"ply {ply}: visit counts differ"
        assert act == oact, f"ply {ply}"
        if opol is None:
            assert pol is None
        else:
            assert (pol.view(np.uint32) == opol.view(np.uint32)).all(), f"ply {ply}: policy bits differ"
        assert pl.tau == orc.tau
        board = utils.step(utils.state_to_board(state, S), act)
        state = utils.board_to_state(board)
        over, _ = utils.is_game_over(board, goal)
        last, ply = act, ply + 1
    _compare_tree(pl._engine.tree_dump(0), orc, S)
    # the dict-like tree view mirrors player.py's State/Action objects
    tv = pl.tree
    assert len(tv) == orc.tree_size()
    pl.close()


class _PseudoAgent:
    """agent_model of NetworkAPI (networkAPI.py:64-65) answering with the 24-bit pseudo-net."""

    class _G:
        def as_default(self):
            import contextlib
            return contextlib.nullcontext()

    def __init__(self, salt, peak):
        self.graph, self.salt, self.peak = self._G(), salt, peak

    def eval(self, data):
        return pseudonet.pseudonet_np(data, self.salt, self.peak, 24)


@pytest.mark.parametrize("S,goal,sims,upper,training,seed,salt,peak", [
    (6, 4, 120, 160, True, 3, 1237, 16384),
    (11, 5, 300, 400, True, 9, 4321, 8192),
    (6, 4, 100, 120, False, 4, 77, 16384),
])
def test_pipe_path_keeps_w_and_q_in_fp64(S, goal, sims, upper, training, seed, salt, peak):
    """The path main.py's workers run: Player(pipe=...) behind NetworkAPI, values as python floats (networkAPI.py:72)
    => W, Q fp64 (SURVEY 8a rule 2, pipe variant).  Drop-in Player + NetworkAPI mirror vs the oracle's pipe variant,
    which tests/test_oracle_golden.py pins on the reference's own Player + NetworkAPI (mcts_*_pipe.npz, same settings)."""
    from alphafive_amd.networkAPI import NetworkAPI
    from alphafive_amd.player import Player
    from alphafive_amd import utils
    cfg = make_cfg(board_size=S, goal=goal, simulation_per_step=sims, upper_simulation_per_step=upper)
    api = NetworkAPI(cfg, _PseudoAgent(salt, peak))
    api.start(reload=False)
    pl = Player(cfg, training=training, pipe=api.get_pipe(reload=False), seed=seed, game_id=3)
    orc = oracle.OraclePlayer(cfg, training=training, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=3, value_f64=True,
                              pv_fn=lambda x: pseudonet.pseudonet_np(x, salt, peak, 24))
    state, last, over, ply = pl.get_init_state(), None, False, 0
    while not over and ply < (3 if S >= 11 else 60):
        pol, act = pl.get_action(state, last_action=last)
        opol, oact, ovis = orc.get_action(state, last, False)
        assert (pl.last_visits == ovis).all() and act == oact, f"ply {ply}"
        if opol is not None:
            assert (pol.view(np.uint32) == opol.view(np.uint32)).all()
        board = utils.step(utils.state_to_board(state, S), act)
        state = utils.board_to_state(board)
        over, _ = utils.is_game_over(board, goal)
        last, ply = act, ply + 1
    dd = pl._engine.tree_dump(0)
    _compare_tree(dd, orc, S)
    assert (dd["w"] != dd["w"].astype(np.float32)).any()      # sums that an fp32 store would have rounded
    tv = pl.tree                                             # the State/Action view hands out python floats, as the reference
    e = max(tv[pl.get_init_state()].a.values(), key=lambda x: x.n)
    assert isinstance(e.w, float) and isinstance(e.q, float)
    # reset(search_tree) keeps the fp64 rows
    pl.reset(search_tree=tv)
    pl.get_action(state, last_action=last) if not over else None
    pl.close()
    api.close()


def test_selfplay_engine_value_f64_matches_pipe_oracle_run():
    """SelfPlayEngine(value_f64=True): whole training episodes (Player.run, player.py:53-82) with the pipe path's fp64 W / Q."""
    import torch
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=100, upper_simulation_per_step=130)
    G = 6
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, 31, 8192, 24), device=0, seed=12, value_f64=True)
    eps = {}
    for _ in range(400):
        sp.run_ticks(64)
        sp.check()
        for raw in sp.pop_raw(cap=64):
            eps.setdefault(raw["game"], raw)
        if len(eps) == G:
            break
    assert len(eps) == G
    for g in range(G):
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=12, game_id=g, value_f64=True,
                                  pv_fn=lambda x: pseudonet.pseudonet_np(x, 31, 8192, 24))
        orec, extra = orc.run()
        raw = eps[g]
        assert raw["seq"] == 0 and raw["T"] == len(orec)
        assert (raw["actions"] == extra["actions"]).all() and (raw["visits"] == extra["visits"]).all()
        assert raw["final_value"] == extra["final_value"]
    sp.close()


def test_player_run_and_reset_match_oracle():
    from alphafive_amd.player import Player
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=60, upper_simulation_per_step=80)
    salt, peak = 4242, 16384
    pl = Player(cfg, training=True, pv_fn=lambda x: pseudonet.pseudonet_np(x, salt, peak), seed=11, game_id=0)
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=11, game_id=0,
                              pseudo_salt=salt, pseudo_peak=peak)
    for _ in range(2):
        rec = pl.run()
        orec, _ = orc.run()
        assert len(rec) == len(orec)
        for (s, p, la, v, w), (os_, op, ola, ov, ow) in zip(rec, orec):
            assert s == os_ and la == ola and v == ov and isinstance(v, float)
            assert w == ow and isinstance(w, np.float32)
            assert (p.view(np.uint32) == op.view(np.uint32)).all()
        assert len(pl.tree) == 0
    pl.close()


@pytest.mark.parametrize("S,goal,sims,upper,G,node_cap", [
    (6, 4, 60, 80, 64, 0),
    (7, 4, 40, 60, 48, 0),
    (6, 4, 60, 80, 16, 100),      # tight store: exercises the superset compaction every move
    (11, 5, 500, 642, 16, 0),     # BASELINE configs[1] search settings: complete 11x11 episodes, 500 sims/move
    (15, 5, 120, 160, 8, 0),      # 4-word bitboards (BASELINE configs[3] board), complete episodes
])
def test_selfplay_episodes_match_oracle(S, goal, sims, upper, G, node_cap):
    import torch
    from alphafive_amd.engine import SelfPlayEngine, assemble_episode
    cfg = make_cfg(board_size=S, goal=goal, simulation_per_step=sims, upper_simulation_per_step=upper)
    salt, peak, seed, first = 31337, 16384, 2024, 100
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, salt, peak), device=0, seed=seed,
                        first_game_id=first, node_cap=node_cap)
    want = 2 * G
    got = {}
    for _ in range(400 if S < 11 else 1500):
        sp.run_ticks(64)
        sp.check()
        for raw in sp.pop_raw(cap=64):
            got.setdefault(raw["game"], []).append(raw)
        if sum(len(v) for v in got.values()) >= want and all(len(got.get(g, [])) >= 1 for g in range(G)):
            break
    assert all(len(got.get(g, [])) >= 1 for g in range(G)), "some games never finished"
    def replay(g):                                   # one game's episodes, in order, on its own oracle player (a host thread each)
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=first + g,
                                  pseudo_salt=salt, pseudo_peak=peak)
        n = 0
        for raw in got[g]:
            orec, extra = orc.run()
            assert raw["T"] == len(orec), f"game {g} seq {raw['seq']}"
            assert (raw["actions"] == extra["actions"]).all()
            assert (raw["visits"] == extra["visits"]).all()
            assert raw["final_value"] == extra["final_value"]
            rec, result = assemble_episode(raw, S, cfg.gamma)
            for (s, p, la, v, w), (os_, op, ola, ov, ow) in zip(rec, orec):
                assert s == os_ and la == ola and v == ov and w == ow
                assert (p.view(np.uint32) == op.view(np.uint32)).all()
            n += 1
        return n

    checked = sum(run_in_threads(replay, range(0, G, max(1, G // 12))))
    assert checked >= 6
    ct = sp.counters()
    assert ct["episodes"] >= len(got)
    sp.close()


@pytest.mark.parametrize("owner", ["side_to_move", "opponent"])
def test_get_action_on_a_decided_root_fails_like_the_reference(owner):
    """an EXTERNAL-mode caller may hand over ANY position.  A root in which a line is already complete — for the
    opponent (value -1) or for the side to move (utils.py:199-235 returns (True, 1.0) there too) — is terminal in every simulation
    (player.py:213-217), so nothing is ever expanded and the reference's get_action dies in calc_policy on an empty node; the
    engine reports AF_ERR_NO_ROOT ("get_action on a finished position") and must not search such a root."""
    from alphafive_amd import engine as eng
    from alphafive_amd.player import Player
    S = 6
    board = np.zeros((S, S), np.int8)
    board[2, 0:4] = 1 if owner == "side_to_move" else -1      # goal = 4 in a row
    board[4, 0:3] = -1 if owner == "side_to_move" else 1
    state = oracle.board_to_state(board)
    assert oracle.is_game_over(board, 4) == (True, 1.0 if owner == "side_to_move" else -1.0)
    cfg = make_cfg(board_size=S, goal=4, simulation_per_step=20, upper_simulation_per_step=30)
    pl = Player(cfg, training=False, pv_fn=lambda x: pseudonet.pseudonet_np(x, 3, 0), seed=1, game_id=0)
    with pytest.raises(eng.EngineError, match="finished position"):
        pl.get_action(state, last_action=None)
    pl.close()
    orc = oracle.OraclePlayer(cfg, training=False, rng_mode=oracle.RNG_PHILOX, seed=1, game_id=0, pseudo_salt=3, pseudo_peak=0)
    with pytest.raises(Exception):
        orc.get_action(state, None)


def test_engine_fails_loudly_on_full_store():
    from alphafive_amd import engine as eng
    import torch
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=50, upper_simulation_per_step=60)
    e = eng.Engine(cfg, 1, mode=eng.MODE_EXTERNAL, training=True, seed=1, node_cap=58)
    planes = torch.zeros((1, 3, 6, 6), device="cuda")
    policy = torch.full((1, 36), 1 / 36, device="cuda")
    value = torch.zeros((1,), device="cuda")
    e.set_root(0, eng.state_to_key("g/g/g/g/g/g/", 6))
    with pytest.raises(eng.EngineError):
        for _ in range(3):
            # second move would need > 58 nodes even after compaction is impossible at the same root
            for _ in range(200):
                e.tick(policy.data_ptr(), value.data_ptr(), planes.data_ptr())
            e.status()
            e.set_root(0, eng.state_to_key("g/g/g/g/g/g/", 6))
    e.close()


def test_batched_arena_matches_per_game_oracle_players():
    """alphafive_amd.arena.play_matches (choose_best_player.py:38-60, many games at once) vs the same
    protocol played game by game with two oracle players on their own trees."""
    import torch
    from alphafive_amd import arena, utils
    S, goal, G = 6, 4, 10
    cfg = make_cfg(board_size=S, goal=goal, simulation_per_step=40, upper_simulation_per_step=60)
    nets = [(101, 16384), (202, 4096)]
    pvs = [(lambda x, sp=sp: pseudonet.pseudonet_torch(x, sp[0], sp[1])) for sp in nets]
    out = arena.play_matches(cfg, pvs[0], pvs[1], G, seed0=5, seed1=6)
    wins, draws = [0, 0], 0
    for i in range(G):
        players = [oracle.OraclePlayer(cfg, training=False, rng_mode=oracle.RNG_PHILOX, seed=s, game_id=i,
                                       pseudo_salt=sp[0], pseudo_peak=sp[1]) for s, sp in zip((5, 6), nets)]
        for pl in players:
            pl.reset()
        board = np.zeros((S, S), np.int8)
        state, action, cur, over, seq = utils.board_to_state(board), None, i % 2, False, []
        while not over:
            _, action, _ = players[cur].get_action(state, action, random_a=True)
            seq.append(action[0] * S + action[1])
            board = utils.step(utils.state_to_board(state, S), action)
            state = utils.board_to_state(board)
            over, v = utils.is_game_over(board, goal)
            cur = (cur + 1) % 2
        assert seq == out["moves"][i], f"game {i}"
        if v == 0.0:
            draws += 1
        else:
            wins[(cur + 1) % 2] += 1
    assert wins == out["wins"] and draws == out["draws"]


@pytest.mark.parametrize("kw,G", [
    (dict(board_size=3, goal=3, simulation_per_step=30, upper_simulation_per_step=40), 32),       # full-board draws
    (dict(board_size=4, goal=3, simulation_per_step=24, upper_simulation_per_step=30, init_temp=0.0105), 24),  # tau <= 0.01
    (dict(board_size=5, goal=4, simulation_per_step=30, upper_simulation_per_step=31), 24),       # cap: few/no sims left
    (dict(board_size=6, goal=4, simulation_per_step=50, upper_simulation_per_step=70, c_puct=1.5,
          dirichlet_alpha=0.15, tau_decay_rate=0.8, init_temp=2.0), 24),
    (dict(board_size=9, goal=5, simulation_per_step=40, upper_simulation_per_step=60), 16),
    (dict(board_size=4, goal=3, simulation_per_step=24, upper_simulation_per_step=30, dirichlet_alpha=0.002), 24),   # all-zero gamma draws
])
def test_selfplay_edge_cases_match_oracle(kw, G):
    """Same edge cases as the tier-A goldens (run_s3_draws, run_s4_lowtau, mcts_s5_cap, mcts_s6_params):
    drawn games, the low-temperature branch of calc_policy, a nearly exhausted simulation cap, other
    hyper-parameters — HIP engine vs oracle, bit for bit."""
    from alphafive_amd.engine import SelfPlayEngine, assemble_episode
    cfg = make_cfg(**kw)
    S = cfg.board_size
    salt, peak, seed = 900, 4096 if S > 3 else 0, 77
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, salt, peak), device=0, seed=seed)
    got = {}
    for _ in range(400):
        sp.run_ticks(50)
        sp.check()
        for raw in sp.pop_raw(cap=128):
            got.setdefault(raw["game"], []).append(raw)
        if all(len(got.get(g, [])) >= 2 for g in range(G)):
            break
    assert all(len(got.get(g, [])) >= 2 for g in range(G))
    draws = 0
    for g in range(0, G, 3):
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=g,
                                  pseudo_salt=salt, pseudo_peak=peak)
        for raw in got[g][:3]:
            orec, extra = orc.run()
            assert raw["T"] == len(orec) and (raw["actions"] == extra["actions"]).all(), f"game {g} seq {raw['seq']}"
            assert (raw["visits"] == extra["visits"]).all() and raw["final_value"] == extra["final_value"]
            rec, result = assemble_episode(raw, S, cfg.gamma)
            for (s, p, la, v, w), (os_, op, ola, ov, ow) in zip(rec, orec):
                assert s == os_ and la == ola and v == ov and w == ow
                assert (p.view(np.uint32) == op.view(np.uint32)).all()
            draws += result == 0
    if S == 3:
        assert draws > 0, "the 3x3 case is there to exercise drawn games"
    sp.close()


def test_unpopped_episodes_apply_back_pressure_and_small_caps_keep_order():
    """a game that has two finished, unpopped episodes must not start writing a third into the oldest
    record buffer.  The engine applies back-pressure instead (the reference's Queue(50), main.py:51,94): the game waits
    at the start of its next episode until the host pops.  Nothing is popped for a long time here, then the episodes
    are popped with caps far smaller than what is pending (the rest must wait for the next call, in order), mid-way
    through the games' resumed play; every episode must still equal the oracle's."""
    from alphafive_amd.engine import SelfPlayEngine
    S, G = 6, 24
    cfg = make_cfg(board_size=S, goal=4, simulation_per_step=20, upper_simulation_per_step=30)
    salt, peak, seed = 4321, 8192, 99
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, salt, peak), device=0, seed=seed)
    sp.run_ticks(6000)                           # far more than two episodes' worth (an episode is <= 36 plies x 20 sims)
    sp.check()
    ct = sp.counters()
    assert ct["episodes"] == 2 * G and ct["stalls"] > 0, ct       # every game finished exactly two episodes, then waited
    got = {}
    first = sp.pop_raw(cap=5)                    # caps smaller than the 48 pending episodes: a prefix in (game, seq) order
    assert [(r["game"], r["seq"]) for r in first] == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]
    for r in first:
        got.setdefault(r["game"], []).append(r)
    sp.run_ticks(300)                            # games 0 and 1 resume (mid-episode now), game 2 has one buffer free
    for rnd in range(40):
        raws = sp.pop_raw(cap=7 if rnd < 3 else 64)   # (small caps serve low-numbered games first: do not starve the rest)
        for r in raws:
            got.setdefault(r["game"], []).append(r)
        sp.run_ticks(200)
        sp.check()
        if all(len(got.get(g, [])) >= 4 for g in range(G)):
            break
    assert all(len(got.get(g, [])) >= 4 for g in range(G))
    for g in range(G):
        assert [e["seq"] for e in got[g]] == list(range(len(got[g])))
    for g in range(0, G, 4):
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=g,
                                  pseudo_salt=salt, pseudo_peak=peak)
        for raw in got[g][:4]:
            orec, extra = orc.run()
            assert raw["T"] == len(orec) and (raw["actions"] == extra["actions"]).all(), f"game {g} seq {raw['seq']}"
            assert (raw["visits"] == extra["visits"]).all() and raw["final_value"] == extra["final_value"]
            assert (raw["lasts"][1:] == raw["actions"][:-1]).all() and raw["lasts"][0] == -1
    sp.close()


def test_engine_rejects_bad_configuration_and_roots():
    from alphafive_amd import engine as eng
    with pytest.raises(eng.EngineError):         # numpy's shape >= 1 gamma branch is not implemented: refuse, do not mis-sample
        eng.Engine(make_cfg(dirichlet_alpha=1.5), 1)
    with pytest.raises(eng.EngineError):
        eng.Engine(make_cfg(dirichlet_alpha=0.0), 1)
    e = eng.Engine(make_cfg(board_size=6, goal=4), 1, mode=eng.MODE_EXTERNAL)
    key = eng.state_to_key("g/g/g/g/g/g/", 6)
    bad = key.copy()
    bad[0] |= np.uint64(1) << np.uint64(40)      # bit 40 >= 36 cells
    with pytest.raises(eng.EngineError):
        e.set_root(0, bad)
    both = key.copy()
    both[0] |= np.uint64(2)
    both[2] |= np.uint64(2)                      # a cell owned by both colours
    with pytest.raises(eng.EngineError):
        e.set_root(0, both)
    e.set_root(0, key)
    e.close()


def test_batched_roots_and_results_equal_the_per_game_calls():
    """af_engine_set_roots / af_engine_move_results (one launch for n games) vs af_engine_set_root / af_engine_move_result."""
    import torch
    from alphafive_amd import engine as eng
    S, G = 6, 6
    cfg = make_cfg(board_size=S, goal=4, simulation_per_step=30, upper_simulation_per_step=40)
    from alphafive_amd import utils
    boards = [np.zeros((S, S), np.int8) for _ in range(3)]
    for (r, c_) in ((1, 1), (1, 2)):
        boards[1] = utils.step(boards[1], (r, c_))
    for (r, c_) in ((2, 2), (3, 3), (3, 1)):
        boards[2] = utils.step(boards[2], (r, c_))
    games = [0, 2, 5]
    keys = np.stack([eng.state_to_key(utils.board_to_state(b_), S) for b_ in boards])
    lcs = [-1, 1 * S + 2, 3 * S + 1]
    outs = []
    for batched in (False, True):
        e = eng.Engine(cfg, G, mode=eng.MODE_EXTERNAL, training=True, seed=9)
        planes = torch.zeros((G, 3, S, S), device="cuda")
        pol, val = torch.zeros((G, S * S), device="cuda"), torch.zeros((G,), device="cuda")
        for rnd in range(2):                                 # the second round searches on the trees of the first
            if batched:
                e.set_roots(games, keys, lcs, random_a=[0, 1, 0], reset_tree=(rnd == 0))
            else:
                for g, k, lc, ra in zip(games, keys, lcs, (0, 1, 0)):
                    e.set_root(g, k, lc, random_a=bool(ra), reset_tree=(rnd == 0))
            while True:
                e.tick(pol.data_ptr(), val.data_ptr(), planes.data_ptr())
                st = e.status()
                if all(st[g] == eng.STATUS_MOVE_DONE for g in games):
                    break
                pr, va = pseudonet.pseudonet_torch(planes, 7, 2048)
                pol.copy_(pr.reshape(G, -1))
                val.copy_(va.reshape(G))
            if batched:
                act, hp, po, vi, tau = e.move_results(games)
                outs.append([(int(act[i]), int(hp[i]), po[i].copy(), vi[i].copy(), float(tau[i])) for i in range(3)])
            else:
                r = []
                for g in games:
                    cell, po, vi, tau = e.move_result(g)
                    r.append((cell, 0 if po is None else 1, po, vi, tau))
                outs.append(r)
        with pytest.raises(eng.EngineError):
            e.move_results([1])                              # game 1 never got a root: not MOVE_DONE
        bad = keys.copy()
        bad[1, 0] |= np.uint64(1) << np.uint64(50)
        with pytest.raises(eng.EngineError):
            e.set_roots(games, bad, lcs)
        e.close()
    for a_, b_ in zip(outs[:2], outs[2:]):
        for x, y in zip(a_, b_):
            assert x[0] == y[0] and x[1] == y[1] and x[4] == y[4]
            if x[2] is not None:
                np.testing.assert_array_equal(x[2], y[2])
            np.testing.assert_array_equal(x[3], y[3])


def test_player_reset_adopts_a_search_tree():
    """Player.reset(search_tree) (player.py:48-51): the handed-over tree becomes the store — round trip through the
    engine, and get_action then honours the adopted visit counts (num = min(sims, upper - sum_n), player.py:140-143)."""
    from alphafive_amd.player import Player
    from alphafive_amd import utils
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=60, upper_simulation_per_step=100)
    pv = lambda x: pseudonet.pseudonet_np(x, 321, 8192)
    a = Player(cfg, training=False, pv_fn=pv, seed=1, game_id=0)
    state, last = a.get_init_state(), None
    for _ in range(2):
        _, act = a.get_action(state, last_action=last)
        board = utils.step(utils.state_to_board(state, 6), act)
        state, last = utils.board_to_state(board), act
    tree = a.tree                                    # snapshot (TreeView: state string -> State-like)
    d0 = a._engine.tree_dump(0)
    b = Player(cfg, training=False, pv_fn=pv, seed=2, game_id=5)
    b.reset(tree)
    assert len(b.tree) == len(tree)
    root_n = tree[state].sum_n if state in tree else 0
    _, act_b = b.get_action(state, last_action=last)
    d1 = b._engine.tree_dump(0)
    k0 = {d0["keys"][i].tobytes(): i for i in range(len(d0["sum_n"]))}
    hits = 0
    for i in range(len(d1["sum_n"])):                # every adopted node is there, counts only grew, priors untouched
        j = k0.get(d1["keys"][i].tobytes())
        if j is None:
            continue
        hits += 1
        assert (d1["n"][i] >= d0["n"][j]).all() and (d1["p"][i].view(np.uint32) == d0["p"][j].view(np.uint32)).all()
        assert (d1["f32"][i] >= d0["f32"][j]).all()
    assert hits == len(d0["sum_n"])
    kroot = {d1["keys"][i].tobytes(): i for i in range(len(d1["sum_n"]))}
    from alphafive_amd import engine as eng
    ri = kroot[eng.state_to_key(state, 6).tobytes()]
    assert d1["sum_n"][ri] == root_n + min(60, 100 - root_n)
    _, act_a = a.get_action(state, last_action=last)  # the original player, same position, eval mode: same move
    assert act_a == act_b
    a.close()
    b.close()


def test_pack_kernels_across_scan_chunks_with_caps():
    """af_engine_pack_episodes over more games than one 1024-thread scan chunk, with caps that cut inside the second and
    third chunk: successive packs must return every finished episode exactly once, in (game, sequence) order, with the
    ply cap respected, and identical to what one uncapped pop returns from a twin engine."""
    from alphafive_amd.engine import SelfPlayEngine
    S, G = 5, 2500
    cfg = make_cfg(board_size=S, goal=4, simulation_per_step=8, upper_simulation_per_step=12)
    mk = lambda: SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, 99, 4096), device=0, seed=17)
    a, b = mk(), mk()
    for sp in (a, b):
        sp.run_ticks(260)                          # long enough for most games to finish one or two episodes (then they wait)
        sp.check()
    ref = b.pop_raw(cap=2 * G)                     # everything at once
    assert len(ref) > G and [(r["game"], r["seq"]) for r in ref] == sorted((r["game"], r["seq"]) for r in ref)
    got = []
    for cap in (700, 1, 1300, 64, 2 * G, 5):       # odd caps: cuts land in different scan chunks
        got += a.pop_raw(cap=cap)
    assert len(got) == len(ref)
    for x, y in zip(got, ref):
        assert (x["game"], x["seq"], x["T"], x["final_value"]) == (y["game"], y["seq"], y["T"], y["final_value"])
        assert (x["keys"] == y["keys"]).all() and (x["visits"] == y["visits"]).all() and (x["actions"] == y["actions"]).all()
        assert (x["policies"].view(np.uint32) == y["policies"].view(np.uint32)).all() and (x["lasts"] == y["lasts"]).all()
    # ply cap: a buffer that holds fewer plies than are pending returns a prefix and leaves the rest
    a.run_ticks(200)
    b.run_ticks(200)
    box = a._outbox(4000)
    box["max_plies"] = 50                          # (the buffer is larger than that: only the cap changes)
    part = a.pop_raw(cap=4000)
    assert 0 < sum(r["T"] for r in part) <= 50
    rest = a.pop_raw(cap=2 * G + 1)
    ref2 = b.pop_raw(cap=2 * G)
    assert [(r["game"], r["seq"]) for r in part + rest] == [(r["game"], r["seq"]) for r in ref2]
    a.close()
    b.close()


def test_pop_with_cap_one_never_stalls_on_a_long_episode():
    """the pack buffer of pop_raw(cap) held cap*40 plies, and af_pack_scan only takes a prefix in (game, seq)
    order — on 11x11 (episodes of up to 121 plies) an episode longer than 40 plies with cap=1 could never be packed and
    blocked every later game: self-play stalled behind the back-pressure without an error.  The buffer now always holds
    one maximum-length episode."""
    from alphafive_amd.engine import SelfPlayEngine
    S, G = 11, 16
    cfg = make_cfg(board_size=S, goal=5, simulation_per_step=6, upper_simulation_per_step=8)     # near-random play: long games
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, 5, 0), device=0, seed=3)
    assert sp._pack_plies(1) >= S * S and sp._pack_plies(2) >= S * S and sp._pack_plies(64) == 64 * 40
    got = []
    for rnd in range(400):
        sp.run_ticks(60)
        sp.check()
        got += sp.pop_raw(cap=1)
        if len(got) >= 3 * G and max(r["T"] for r in got) > 80:
            break
    assert len(got) >= 3 * G, (len(got), sp.counters())
    assert max(r["T"] for r in got) > 40                                   # episodes the old 40-ply buffer could not hold
    seqs = {}
    for r in got:
        seqs.setdefault(r["game"], []).append(r["seq"])
    assert all(v == list(range(len(v))) for v in seqs.values())
    sp.close()


def test_simulation_budget_follows_the_live_config():
    """player.py:140-143 reads config.simulation_per_step / upper_simulation_per_step at every get_action: changing them
    between two moves changes the next move's budget (af_engine_set_simulations; a move in progress keeps its own)."""
    from alphafive_amd.player import Player
    from alphafive_amd import engine as eng, utils
    S = 7
    cfg = make_cfg(board_size=S, goal=4, simulation_per_step=90, upper_simulation_per_step=120)
    salt, peak = 31, 4096
    pl = Player(cfg, training=True, pv_fn=lambda x: pseudonet.pseudonet_np(x, salt, peak), seed=2, game_id=5)
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=2, game_id=5, pseudo_salt=salt, pseudo_peak=peak)
    state, last = pl.get_init_state(), None
    for sims, upper in ((90, 120), (20, 30), (150, 400), (40, 41), (90, 120)):
        cfg.simulation_per_step, cfg.upper_simulation_per_step = sims, upper
        orc.set_simulations(sims, upper)
        pol, act = pl.get_action(state, last_action=last)
        opol, oact, ovis = orc.get_action(state, last)
        assert (pl.last_visits == ovis).all() and act == oact and (pol.view(np.uint32) == opol.view(np.uint32)).all()
        board = utils.step(utils.state_to_board(state, S), act)
        state, last = utils.board_to_state(board), act
    _compare_tree(pl._engine.tree_dump(0), orc, S)
    with pytest.raises(eng.EngineError):
        pl._engine.set_simulations(0, 10)
    with pytest.raises(eng.EngineError):
        pl._engine.set_simulations(10 ** 6, 10 ** 6)          # beyond the node capacity chosen at create
    pl.close()
