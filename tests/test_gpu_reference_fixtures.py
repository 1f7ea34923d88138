"""HIP engine against arrays the UNMODIFIED reference generated — no oracle in between (VERDICT r4 "missing" #2 / next #5).

Every other `-m gpu` parity test compares the engine with the C oracle in Philox mode, so the reference's own outputs reach the GPU
only through "oracle == reference (MT, CPU)" + "HIP == oracle (Philox, GPU)".  In eval mode (`training=False`: self_play.py:79-106,
choose_best_player.py:52) the reference applies no noise (player.py:247) and forces no root visits (:264), so the only random
decisions of a search are uniform picks among tied candidates (:277-279 score ties, :101-102 max-visit ties).  The
`tests/golden/mcts_*_eval_sharp.npz` traces were recorded from the reference `Player` (tests/golden/make_golden.py:gen_sharp_cases)
with a pseudo-net whose priors never coincide; tests/test_oracle_golden.py checks on the CPU that no score tie occurred in them
(oracle tie_stats), i.e. their searches are a function of the net alone.  Here the HIP `Player` is handed the recorded states and
must reproduce the reference-generated visit-count vectors on EVERY ply and the chosen move on every ply without a max-visit tie
(on a tie ply the reference's pick came from Python's MT stream; both picks must lie in the tie set — the search is unaffected
because the recorded states are replayed), and every node left in the device store must equal the reference's node."""
import hashlib
import os

import numpy as np
import pytest

import pseudonet
from conftest import GOLDEN, cfg_from_golden

pytestmark = pytest.mark.gpu


def _cell(S, c):
    return None if c < 0 else (int(c) // S, int(c) % S)


def _digest(sum_n, n, w, f32, p, legal):
    """tests/golden/make_golden.py:node_digest on a device node."""
    h = hashlib.blake2b(digest_size=8)
    h.update(np.int32(sum_n).tobytes())
    h.update(np.ascontiguousarray(n, np.int32).tobytes())
    h.update(np.ascontiguousarray(w, np.float32).astype(np.float64).tobytes())
    h.update(np.ascontiguousarray(np.where(legal, f32, 0), np.uint8).tobytes())
    h.update(np.ascontiguousarray(p, np.float32).tobytes())
    return np.frombuffer(h.digest(), np.uint64)[0]


@pytest.mark.parametrize("name,min_nodes", [("mcts_s6_eval_sharp.npz", 200), ("mcts_s11_eval_sharp.npz", 500),
                                            ("mcts_s15_eval_sharp.npz", 200)])
def test_hip_player_reproduces_reference_generated_eval_traces(name, min_nodes):
    from alphafive_amd import engine as eng, utils
    from alphafive_amd.player import Player
    z = dict(np.load(os.path.join(GOLDEN, name)))
    cfg = cfg_from_golden(z)
    S = cfg.board_size
    assert not bool(z["training"]) and int(z["sharp"]) == 1
    salt, peak, vbits = int(z["salt"]), int(z["peak"]), int(z["vbits"])
    # the reference's own evaluator seam: a numpy callable (player.py:190-192); every leaf crosses the host boundary
    pl = Player(cfg, training=False, pv_fn=lambda x: pseudonet.pseudonet_np(x, salt, peak, vbits, 1), seed=123, game_id=5)
    T = len(z["states"])
    tie_plies = 0
    for t in range(T):
        pol, act = pl.get_action(str(z["states"][t]), last_action=_cell(S, z["lasts"][t]))
        assert pol is None                                            # player.py:106-107
        vis = np.asarray(pl.last_visits)
        assert (vis == z["visits"][t]).all(), f"{name} ply {t}: visit counts differ from the reference's"
        best = set(np.flatnonzero(vis == vis.max()).tolist())
        got, want = act[0] * S + act[1], int(z["actions"][t])
        if len(best) == 1:
            assert got == want, f"{name} ply {t}: move {act} != reference {_cell(S, want)}"
        else:                                                         # random.choice(best_actions), player.py:101-102
            tie_plies += 1
            assert got in best and want in best
    assert T - tie_plies >= 0.8 * T                                   # moves compared on >= 80 % of the plies, visits on all
    # the store: whatever the collector kept (nodes whose stones contain the root's) must be the reference's nodes
    dump = pl._engine.tree_dump(0)
    ref = {str(k): i for i, k in enumerate(z["tree_keys"])}
    assert len(dump["sum_n"]) >= min_nodes
    for i in range(len(dump["sum_n"])):
        state = eng.key_to_state(dump["keys"][i], S)
        j = ref.get(state)
        assert j is not None, f"device node {state} unknown to the reference"
        assert dump["sum_n"][i] == z["tree_sum_n"][j] == dump["n"][i].sum()
        legal = utils.state_to_board(state, S).reshape(-1) == 0
        if "tree_digest" in z:
            assert _digest(dump["sum_n"][i], dump["n"][i], dump["w"][i], dump["f32"][i], dump["p"][i], legal) == z["tree_digest"][j], state
        else:
            assert (dump["n"][i] == z["tree_n"][j]).all() and (dump["p"][i] == z["tree_p"][j]).all()
            assert (dump["w"][i].astype(np.float64) == z["tree_w"][j]).all()
            assert (dump["f32"][i][legal] == z["tree_wf32"][j][legal]).all()
    pl.close()
