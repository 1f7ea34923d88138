"""Tier A: the C oracle (MT19937 mode) against golden vectors recorded from the
unmodified reference (tests/golden/make_golden.py).  Bit-exact for visit counts,
actions, tree contents (n, w, dtype class, p, sum_n), records; the temperature
policy (fp32 np.power, not bit-portable — SURVEY §8a row M) to 3e-7 relative."""
import glob
import os

import numpy as np
import pytest

import oracle
import pseudonet
from conftest import GOLDEN, cfg_from_golden


def _cell(S, c):
    return None if c < 0 else (int(c) // S, int(c) % S)


def test_rules_known_answers():
    z = np.load(os.path.join(GOLDEN, "rules.npz"))
    for k in range(len(z["S"])):
        S, goal = int(z["S"][k]), int(z["goal"][k])
        b = z["board"][k][:S, :S].copy()
        s = str(z["state"][k])
        assert oracle.board_to_state(b) == s
        assert (oracle.state_to_board(s, S) == b).all()
        over, v = oracle.is_game_over(b, goal)
        assert over == bool(z["over"][k]) and v == float(z["value"][k])
        la = oracle.legal_actions(b)
        L = int(z["legal_count"][k])
        assert [i * S + j for i, j in la] == list(z["legal_cells"][k][:L])
        last = la[L // 2] if la else None
        assert (oracle.board_to_inputs(b, last) == z["inputs"][k][:, :S, :S]).all()
        if la:
            assert (oracle.step(b, la[0]) == z["stepped"][k][:S, :S]).all()
    for name in z.files:
        if name.startswith("cw_"):
            _, T, g = name.split("_")
            w = oracle.construct_weights(int(T), float(g))
            assert w.dtype == np.float32 and (w == z[name]).all(), name


def test_survey_known_answers():
    # SURVEY.md §8c known-answer seeds
    init = oracle.board_to_state(np.zeros((11, 11), np.int8))
    assert init == "l/l/l/l/l/l/l/l/l/l/l/"
    b = oracle.step(oracle.state_to_board(init, 11), (6, 3))
    assert oracle.board_to_state(b) == "l/l/l/l/l/l/d1h/l/l/l/l/"
    assert oracle.legal_actions(np.array([[0, 1], [0, 0]], np.int8)) == [(0, 0), (1, 0), (1, 1)]
    np.testing.assert_array_equal(oracle.construct_weights(5, 0.94),
                                  np.array([0.8802264, 0.936411, 0.996182, 1.059768, 1.1274128], np.float32))


def _node_digest(nd, state, S):
    """The digest tests/golden/make_golden.py:node_digest takes of a reference tree node, from an oracle node."""
    import hashlib
    legal = (oracle.state_to_board(state, S).reshape(-1) == 0)
    h = hashlib.blake2b(digest_size=8)
    h.update(np.int32(nd["sum_n"]).tobytes())
    h.update(np.ascontiguousarray(nd["n"], np.int32).tobytes())
    h.update(np.ascontiguousarray(nd["w"], np.float32).astype(np.float64).tobytes())
    h.update(np.ascontiguousarray(np.where(legal, nd["f32"], 0), np.uint8).tobytes())
    h.update(np.ascontiguousarray(nd["p"], np.float32).tobytes())
    return np.frombuffer(h.digest(), np.uint64)[0]


MCTS = sorted(glob.glob(os.path.join(GOLDEN, "mcts_*.npz")))


@pytest.mark.parametrize("path", MCTS, ids=[os.path.basename(p) for p in MCTS])
def test_mcts_trace_bit_exact(path):
    z = dict(np.load(path))
    cfg = cfg_from_golden(z)
    S = cfg.board_size
    pipe = bool(z.get("pipe", False))            # *_pipe.npz: the reference Player behind its NetworkAPI pipe (fp64 w / q)
    vbits = int(z.get("vbits", 16))
    sharp = int(z.get("sharp", 0))               # *_sharp.npz: the tie-free eval-mode traces (tests/pseudonet.py)
    pv = None if (vbits == 16 and not sharp) else (lambda x: pseudonet.pseudonet_np(x, int(z["salt"]), int(z["peak"]), vbits, sharp))
    pl = oracle.OraclePlayer(cfg, training=bool(z["training"]), rng_mode=oracle.RNG_MT, seed=int(z["seed"]), pv_fn=pv,
                             pseudo_salt=int(z["salt"]), pseudo_peak=int(z["peak"]), value_f64=pipe)
    for t in range(len(z["states"])):
        pol, act, vis = pl.get_action(str(z["states"][t]), _cell(S, z["lasts"][t]), bool(z["random_a"]))
        assert (vis == z["visits"][t]).all(), f"ply {t}: visit counts differ"
        assert act == _cell(S, z["actions"][t]), f"ply {t}: action differs"
        if z["has_policy"][t]:
            np.testing.assert_allclose(pol.reshape(-1), z["policies"][t], rtol=3e-7, atol=0)
        else:
            assert pol is None
        assert pl.tau == float(z["taus"][t])
    if sharp:
        # the property tests/test_gpu_reference_fixtures.py rests on: no score tie at any select of these eval-mode games, so the
        # searches are a function of the net alone and the HIP engine (another generator) can be held to these arrays directly
        ties = pl.tie_stats()
        assert ties["select"] == 0 and ties["forced"] == 0 and ties["best"] <= 0.2 * len(z["states"]), ties
    # both MT streams consumed exactly as many words as the reference did
    assert pl.np_u32() == int(z["np_next"])
    assert pl.py_u32() == int(z["py_next"])
    # whole tree
    assert pl.tree_size() == len(z["tree_keys"])
    if "tree_digest" in z:                       # full-size traces: one digest per node (make_golden.py:node_digest)
        for k, key in enumerate(z["tree_keys"]):
            nd = pl.node(str(key))
            assert nd is not None and nd["sum_n"] == z["tree_sum_n"][k] and nd["sum_n"] == nd["n"].sum()
            assert _node_digest(nd, str(key), S) == z["tree_digest"][k], f"node {k} ({key}) differs"
        return
    for k, key in enumerate(z["tree_keys"]):
        nd = pl.node(str(key))
        assert nd is not None
        assert nd["sum_n"] == z["tree_sum_n"][k]
        legal = z["tree_legal"][k].astype(bool)
        assert (nd["n"] == z["tree_n"][k]).all()
        if pipe:
            assert (nd["w64"] == z["tree_w"][k]).all() and not z["tree_wf32"][k].any()
            with np.errstate(divide="ignore", invalid="ignore"):
                q = np.where(nd["n"] == 0, 0.0, nd["w64"] / nd["n"].astype(np.float64))
            assert (q == z["tree_q"][k]).all()
            assert (nd["p"] == z["tree_p"][k]).all()
            continue
        assert (nd["w"].astype(np.float64) == z["tree_w"][k]).all()
        assert (nd["f32"][legal] == z["tree_wf32"][k][legal]).all()
        assert (nd["p"] == z["tree_p"][k]).all()
        n = nd["n"].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            q32 = (nd["w"] / nd["n"].astype(np.float32)).astype(np.float64)
            q64 = nd["w"].astype(np.float64) / n
        q = np.where(nd["n"] == 0, 0.0, np.where(nd["f32"] == 1, q32, q64))
        assert (q == z["tree_q"][k]).all()
        assert nd["sum_n"] == nd["n"].sum()       # SURVEY §8a rule 5


RUNS = sorted(glob.glob(os.path.join(GOLDEN, "run_*.npz")))


@pytest.mark.parametrize("path", RUNS, ids=[os.path.basename(p) for p in RUNS])
def test_run_episode_records(path):
    z = dict(np.load(path))
    cfg = cfg_from_golden(z)
    S = cfg.board_size
    pl = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_MT, seed=int(z["seed"]),
                             pseudo_salt=int(z["salt"]), pseudo_peak=int(z["peak"]))
    for e in range(int(z["episodes"])):
        recs, extra = pl.run()
        assert len(recs) == len(z[f"ep{e}_states"])
        for t, (s, pol, la, v, w) in enumerate(recs):
            assert s == str(z[f"ep{e}_states"][t])
            np.testing.assert_allclose(pol.reshape(-1), z[f"ep{e}_policies"][t], rtol=3e-7, atol=0)
            assert la == _cell(S, z[f"ep{e}_lasts"][t])
            assert v == float(z[f"ep{e}_values"][t]) and isinstance(v, float)
            assert w == z[f"ep{e}_weights"][t] and isinstance(w, np.float32)
        value = recs[-1][-2]
        result = 0 if value == 0.0 else (1 if len(recs) % 2 == 1 else -1)     # main.py:85-93
        assert result == int(z[f"ep{e}_result"])
        assert pl.tree_size() == 0
    assert pl.np_u32() == int(z["np_next"])
    assert pl.py_u32() == int(z["py_next"])
