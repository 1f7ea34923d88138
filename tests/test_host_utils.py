"""alphafive_amd.utils (host mirror of utils.py:149-296) against the reference-recorded goldens."""
import os

import numpy as np

from alphafive_amd import utils
from conftest import GOLDEN


def test_rules_match_reference_goldens():
    z = np.load(os.path.join(GOLDEN, "rules.npz"))
    for k in range(len(z["S"])):
        S, goal = int(z["S"][k]), int(z["goal"][k])
        b = z["board"][k][:S, :S].copy()
        s = str(z["state"][k])
        assert utils.board_to_state(b) == s
        assert (utils.state_to_board(s, S) == b).all()
        over, v = utils.is_game_over(b, goal)
        assert (over, v) == (bool(z["over"][k]), float(z["value"][k])), (k, s)
        assert isinstance(v, float)
        la = utils.get_legal_actions(b)
        L = int(z["legal_count"][k])
        assert [i * S + j for i, j in la] == list(z["legal_cells"][k][:L])
        last = la[L // 2] if la else None
        x = utils.board_to_inputs(b, last_action=last)
        assert x.dtype == np.float32 and (x == z["inputs"][k][:, :S, :S]).all()
        if la:
            assert (utils.step(b.copy(), la[0]) == z["stepped"][k][:S, :S]).all()
    for name in z.files:
        if name.startswith("cw_"):
            _, T, g = name.split("_")
            w = utils.construct_weights(int(T), gamma=float(g))
            assert w.dtype == np.float32 and (w == z[name]).all(), name


def test_episode_assembly_matches_run_goldens():
    """engine.assemble_episode (Player.run tail + gen_data result) on records rebuilt from the goldens."""
    from alphafive_amd import engine as eng
    z = np.load(os.path.join(GOLDEN, "run_s6.npz"))
    S = int(z["cfg_board_size"])
    for e in range(int(z["episodes"])):
        T = len(z[f"ep{e}_states"])
        final = z[f"ep{e}_values"][T - 1]          # value of the last mover = -final is_game_over value
        raw = dict(T=T, final_value=-float(final), lasts=z[f"ep{e}_lasts"], policies=z[f"ep{e}_policies"],
                   keys=np.stack([_key(str(s), S) for s in z[f"ep{e}_states"]]))
        try:
            rec, result = eng.assemble_episode(raw, S, float(z["cfg_gamma"]))
        except eng.EngineError:
            import pytest
            pytest.skip("libaf_hip.so not built")
        assert result == int(z[f"ep{e}_result"])
        for t, (s, p, la, v, w) in enumerate(rec):
            assert s == str(z[f"ep{e}_states"][t])
            assert v == float(z[f"ep{e}_values"][t]) and w == z[f"ep{e}_weights"][t]


def _key(state, S):
    from alphafive_amd import engine as eng
    return eng.state_to_key(state, S)


def test_randomstack_matches_reference_under_seeded_streams():
    """alphafive_amd.utils.RandomStack vs goldens recorded from utils.RandomStack (utils.py:14-146):
    same accept/duplicate/evict decisions and the same augmented batches, consuming np.random and
    random in the same order."""
    import contextlib
    import io
    import random
    z = dict(np.load(os.path.join(GOLDEN, "randomstack.npz")))
    S = int(z["S"])
    np.random.seed(21)
    random.seed(21)
    st = utils.RandomStack(board_size=S, length=int(z["length"]))
    accepted = []
    with contextlib.redirect_stdout(io.StringIO()):
        for e in range(int(z["n_episodes"])):
            recs = []
            for t in range(len(z[f"ep{e}_states"])):
                la = int(z[f"ep{e}_lasts"][t])
                recs.append((str(z[f"ep{e}_states"][t]), z[f"ep{e}_policies"][t], None if la < 0 else (la // S, la % S),
                             float(z[f"ep{e}_values"][t]), np.float32(z[f"ep{e}_weights"][t])))
            accepted.append(st.push(recs, int(z[f"ep{e}_result"])))
    assert accepted == list(z["accepted"])
    assert st.data_len == list(z["data_len"]) and st.result == list(z["result"])
    assert len(st.data) == int(z["n_data"]) and st.is_full() == (len(st.data) >= st.length)
    assert (st.black_win, st.white_win) == (int(z["black_win"]), int(z["white_win"]))
    assert st.data[0][0] == str(z["first_state"]) and st.data[-1][0] == str(z["last_state"])
    for b in range(3):
        boards, weights, values, policies = st.get_data(batch_size=48)
        assert (boards == z[f"batch{b}_boards"]).all() and (weights == z[f"batch{b}_weights"]).all()
        assert (values == z[f"batch{b}_values"]).all() and (policies == z[f"batch{b}_policies"]).all()
        assert boards.dtype == np.float32 and policies.shape == (48, S * S)
    assert int(np.random.randint(0, 2 ** 32, dtype=np.uint64)) == int(z["np_next"])
    assert random.getrandbits(32) == int(z["py_next"])
