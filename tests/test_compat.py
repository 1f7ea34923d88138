"""alphafive_amd.compat.install(): the reference's scripts keep their import lines (main.py:2-15, self_play.py:2-8,
choose_best_player.py:2-9) and get the engine's classes; the replay pickles stay interchangeable with the reference's
own utils.RandomStack (utils.py:29-57)."""
import importlib.util
import os
import random
import sys

import numpy as np
import pytest

REF = "/root/reference"


@pytest.fixture
def installed():
    from alphafive_amd import compat
    done = compat.install()
    yield done
    compat.uninstall()


# a caller written with the reference's own import lines (main.py:3,4,7,11,15 and the body of gen_data, main.py:82-94)
_CALLER = '''
import utils
from genData.network import ResNet as model
import config
from genData.player import Player
from genData.networkAPI import NetworkAPI
from utils import RandomStack


def gen_data_once(player):
    game_record = player.run()
    value = game_record[-1][-2]
    game_length = len(game_record)
    if value == 0.0:
        result = utils.DRAW
    elif game_length % 2 == 1:
        result = utils.BLACK_WIN
    else:
        result = utils.WHITE_WIN
    return game_record, result
'''


def _load_caller():
    ns = {}
    exec(compile(_CALLER, "<reference-shaped caller>", "exec"), ns)
    return ns


def test_reference_import_lines_resolve_to_the_engine(installed):
    import alphafive_amd.config
    import alphafive_amd.network
    import alphafive_amd.networkAPI
    import alphafive_amd.player
    import alphafive_amd.utils
    ns = _load_caller()
    assert ns["Player"] is alphafive_amd.player.Player
    assert ns["model"] is alphafive_amd.network.ResNet
    assert ns["NetworkAPI"] is alphafive_amd.networkAPI.NetworkAPI
    assert ns["RandomStack"] is alphafive_amd.utils.RandomStack
    assert ns["utils"] is alphafive_amd.utils and ns["config"] is alphafive_amd.config
    assert (ns["utils"].BLACK_WIN, ns["utils"].WHITE_WIN, ns["utils"].DRAW) == (1, -1, 0)
    # the attributes main.py / self_play.py / choose_best_player.py read from config
    for a in ("board_size", "buffer_size", "simulation_per_step", "upper_simulation_per_step", "goal", "batch_size",
              "ckpt_path", "total_step", "max_processes", "get_lr"):
        assert hasattr(ns["config"], a), a
    import genData
    assert genData.player is alphafive_amd.player


def test_install_keeps_a_caller_owned_config_and_uninstall_restores():
    import types
    from alphafive_amd import compat
    before = {k: sys.modules.get(k) for k in ("utils", "config", "genData", "genData.player")}
    mine = types.ModuleType("config")
    mine.board_size = 7
    compat.install(config=mine)
    try:
        import config
        assert config is mine
    finally:
        compat.uninstall()
    assert {k: sys.modules.get(k) for k in before} == before


def _episode(S, T, rng):
    from alphafive_amd import utils
    board = np.zeros((S, S), np.int8)
    rec, last = [], None
    w = utils.construct_weights(T, gamma=0.94)
    v = 1.0
    for t in range(T):
        p = rng.rand(S, S).astype(np.float32)
        rec.append((utils.board_to_state(board), p / p.sum(), last, v, w[t]))
        legal = utils.get_legal_actions(board)
        last = legal[rng.randint(len(legal))]
        board = utils.step(board, last)
        v = -v
    return rec


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "utils.py")), reason="build container only: needs /root/reference")
def test_replay_pickles_load_in_the_reference_randomstack(tmp_path, monkeypatch):
    """RandomStack.save here -> utils.RandomStack.load of the unmodified reference (and back): same records, same
    bookkeeping, and the same augmented batch under the same seeds."""
    from alphafive_amd import utils as mine
    spec = importlib.util.spec_from_file_location("af_reference_utils", os.path.join(REF, "utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    monkeypatch.chdir(tmp_path)
    os.mkdir("data_buffer")
    rng = np.random.RandomState(3)
    S = 11
    a = mine.RandomStack(S, length=300)
    random.seed(1)
    for i in range(14):
        T = int(rng.randint(22, 40))
        a.push(_episode(S, T, rng), mine.BLACK_WIN if T % 2 else mine.WHITE_WIN)
    a.save(60)
    b = ref.RandomStack(S, length=300)
    b.load(60)
    assert len(b.data) == len(a.data) and b.data_len == a.data_len and b.result == a.result
    assert (b.black_win, b.white_win) == (a.black_win, a.white_win)
    for x, y in zip(a.data, b.data):
        assert x[0] == y[0] and (x[1] == y[1]).all() and x[2:] == y[2:]
    for st in (a, b):
        np.random.seed(5)
        random.seed(5)
        st.batch = st.get_data(batch_size=64)
    for x, y in zip(a.batch, b.batch):
        assert x.dtype == y.dtype and (x == y).all()
    # and the other way round: what the reference writes, this class loads
    b.save(120)
    c = mine.RandomStack(S, length=300)
    c.load(120)
    assert c.data_len == a.data_len and c.result == a.result and len(c.data) == len(a.data)


@pytest.mark.gpu
def test_gen_data_shaped_caller_runs_on_the_engine(installed):
    """main.py:82-94's gen_data body, written against the reference's module names, plays an episode on the HIP engine and
    yields the queue item format (list of 5-tuples, result code)."""
    import pseudonet
    from conftest import make_cfg
    ns = _load_caller()
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=40, upper_simulation_per_step=60)
    player = ns["Player"](cfg, training=True, pv_fn=lambda x: pseudonet.pseudonet_np(x, 7, 16384), seed=11, game_id=3)
    rec, result = ns["gen_data_once"](player)
    player.close()
    assert result in (1, -1, 0) and 7 <= len(rec) <= 36
    s, p, la, v, w = rec[-1]
    assert isinstance(s, str) and p.shape == (6, 6) and abs(float(p.sum()) - 1) < 1e-5
    st = ns["RandomStack"](6, length=100)
    st.push(rec, result)
