"""SelfPlayEngine.run_ticks_graph(): the steady-state loop of main.py:57-94 (five workers + the NetworkAPI thread) as ONE
HIP graph per n ticks, progress read one replay late (ABI v4 af_engine_progress_async).  The graph must be the same
computation as the eager launches — bit for bit — and must follow everything a launch has baked in."""
import os

import numpy as np
import pytest

import oracle
import pseudonet
from conftest import GOLDEN, make_cfg

pytestmark = pytest.mark.gpu
W = os.path.join(GOLDEN, "alphaFive-6960.weights.npz")


def _drain(sp, cap=256):
    out = []
    while True:
        raws = sp.pop_raw(cap)
        out += raws
        if len(raws) < cap:
            return out


def _same_episodes(a, b):
    assert len(a) == len(b) and len(a) > 0
    ka = sorted(a, key=lambda e: (e["game"], e["seq"]))
    kb = sorted(b, key=lambda e: (e["game"], e["seq"]))
    for x, y in zip(ka, kb):
        assert (x["game"], x["seq"], x["T"], x["final_value"]) == (y["game"], y["seq"], y["T"], y["final_value"])
        for k in ("keys", "visits", "actions", "lasts"):
            assert (x[k] == y[k]).all(), k
        assert (x["policies"].view(np.uint32) == y["policies"].view(np.uint32)).all()


def test_graph_loop_equals_eager_loop_and_the_oracle_on_the_pseudo_net():
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=60, upper_simulation_per_step=80)
    salt, peak, seed, G, n, reps = 7, 16384, 5, 48, 16, 120
    a = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, salt, peak), device=0, seed=seed)
    b = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, salt, peak), device=0, seed=seed)
    got_a, got_b, last = [], [], (0, 0)
    for r in range(reps):
        a.run_ticks_graph(n)
        lag = a.progress_lagged()
        assert lag[0] >= last[0] and lag[1] >= last[1]            # monotone, never ahead of the truth
        last = lag
        if r % 8 == 7:
            a.check()
            now = a.progress()
            assert lag[0] <= now[0] and lag[1] <= now[1]
            got_a += _drain(a)
    assert a._graph is not None and a.ticks == 1 + reps * n       # one eager warm-up tick, then replays only
    b.run_ticks(a.ticks)
    b.check()
    a.check()
    got_a += _drain(a)
    got_b += _drain(b)
    # un-popped episodes stall a game (back-pressure), so the two engines are compared where both were drained alike: per game,
    # the common prefix of finished episodes — and the first episode of every game against the oracle
    by = {}
    for e in got_b:
        by[(e["game"], e["seq"])] = e
    common = [e for e in got_a if (e["game"], e["seq"]) in by]
    assert len(common) >= G // 2
    _same_episodes(common, [by[(e["game"], e["seq"])] for e in common])
    for e in [x for x in got_a if x["seq"] == 0][:8]:
        orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=seed, game_id=e["game"],
                                  pseudo_salt=salt, pseudo_peak=peak)
        orec, extra = orc.run()
        assert e["T"] == len(orec) and (e["visits"] == extra["visits"]).all() and (e["actions"] == extra["actions"]).all()
    a.close()
    b.close()


def test_graph_loop_with_the_hand_written_net_follows_budget_and_weight_changes():
    """11x11, alphaFive-6960 through af_conv_f16s inside the graph (value branch fork / join captured): same counters and same
    trees as the eager loop after the same number of ticks; a changed simulation budget or weight set drops the graph."""
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet, random_variables
    cfg = make_cfg(simulation_per_step=40, upper_simulation_per_step=60)
    G = 96
    nets = [ResNet(11, device="cuda") for _ in range(2)]
    for nt in nets:
        nt.load_npz(W)
    a = SelfPlayEngine(cfg, G, nets[0].select_backend("hip"), device=0, seed=9)
    b = SelfPlayEngine(cfg, G, nets[1].select_backend("hip"), device=0, seed=9)
    for _ in range(6):
        a.run_ticks_graph(8)
    k0 = a._graph[0]
    b.run_ticks(a.ticks)
    a.check(), b.check()
    assert a.counters() == b.counters()
    a.engine.set_simulations(24, 30)
    b.engine.set_simulations(24, 30)
    for _ in range(6):
        a.run_ticks_graph(8)
    k1 = a._graph[0]
    assert k1 != k0 and k1[1][1:3] == (24, 30)
    b.run_ticks(a.ticks - b.ticks)
    a.check(), b.check()
    assert a.counters() == b.counters()
    for nt in nets:
        nt.set_variables(random_variables(11, seed=3))
    for _ in range(4):
        a.run_ticks_graph(8)
    assert a._graph[0] != k1
    b.run_ticks(a.ticks - b.ticks)
    a.check(), b.check()
    assert a.counters() == b.counters()
    for g in (0, 17, G - 1):
        ta, tb = a.engine.tree_dump(g), b.engine.tree_dump(g)
        for k in ("keys", "sum_n", "n"):
            assert (ta[k] == tb[k]).all(), k
        assert (ta["w"].view(np.uint32) == tb["w"].view(np.uint32)).all() and (ta["p"].view(np.uint32) == tb["p"].view(np.uint32)).all()
    a.close()
    b.close()


@pytest.mark.parametrize("form", ["bound_method", "wrapped_with_version"])
def test_graph_loop_follows_a_weight_update_behind_a_bound_method_or_wrapper(form):
    """pv_device = net.eval_device (a bound method: no .weights_version of its own) or a lambda around it.
    The graph key's weight component must still change with net.set_variables() — a replay skips the Python wrapper that reloads
    the weights, so a key without it would keep playing on the stale set, silently."""
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet, random_variables
    cfg = make_cfg(simulation_per_step=30, upper_simulation_per_step=40)
    G = 64
    nets = [ResNet(11, device="cuda") for _ in range(2)]
    for nt in nets:
        nt.load_npz(W)
    if form == "bound_method":
        a = SelfPlayEngine(cfg, G, nets[0].eval_device, device=0, seed=3)
    else:
        a = SelfPlayEngine(cfg, G, lambda x: nets[0].eval_device(x), device=0, seed=3, weights_version=lambda: nets[0].version)
    b = SelfPlayEngine(cfg, G, nets[1].select_backend("hip"), device=0, seed=3)
    for _ in range(4):
        a.run_ticks_graph(8)
    k0 = a._graph[0]
    assert k0[2] == nets[0].version
    b.run_ticks(a.ticks)                           # the same ticks on the old weights ...
    for nt in nets:
        nt.set_variables(random_variables(11, seed=5))
    for _ in range(4):
        a.run_ticks_graph(8)
    assert a._graph[0] != k0 and a._graph[0][2] == nets[0].version
    b.run_ticks(a.ticks - b.ticks)                 # ... and on the new ones
    a.check(), b.check()
    assert a.counters() == b.counters()
    for g in (0, G - 1):
        ta, tb = a.engine.tree_dump(g), b.engine.tree_dump(g)
        assert (ta["keys"] == tb["keys"]).all() and (ta["n"] == tb["n"]).all()
        assert (ta["p"].view(np.uint32) == tb["p"].view(np.uint32)).all()          # priors of the NEW weights in both
    a.close()
    b.close()
    for nt in nets:
        nt.close()


def test_graph_loop_refuses_an_evaluator_of_ours_without_a_weight_version():
    from alphafive_amd.engine import SelfPlayEngine, EngineError

    class Anonymous(object):                       # takes bind_outputs (so it owns weights and output tensors) but names no version
        def __call__(self, x):
            raise AssertionError("never called")

        def bind_outputs(self, p, v):
            pass
    sp = SelfPlayEngine(make_cfg(), 4, Anonymous(), device=0)      # constructing is fine (eager tick() users: ADVICE r5) ...
    with pytest.raises(EngineError, match="weights_version"):      # ... replaying launches over weights nobody versions is not
        sp.run_ticks_graph(4)
    sp.close()
    with pytest.raises(EngineError, match="weights_version"):      # ... and neither is keeping their evaluations
        SelfPlayEngine(make_cfg(), 4, Anonymous(), device=0, eval_memo=True)


@pytest.mark.parametrize("board,goal,value_f64", [(15, 5, False), (7, 4, True)])
def test_graph_loop_on_the_other_instantiations(board, goal, value_f64):
    """af_tick_kernel<4, false> (15x15: four-word bitboards) and <2, true> (fp64 W / Q: the pipe path's arithmetic) inside the graph:
    same counters, progress and trees as the eager loop after the same number of ticks."""
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=board, goal=goal, simulation_per_step=50, upper_simulation_per_step=70)
    G = 40
    mk = lambda: SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, 11, 8192, vbits=24), device=0, seed=4, value_f64=value_f64)
    a, b = mk(), mk()
    for _ in range(30):
        a.run_ticks_graph(12)
    b.run_ticks(a.ticks)
    a.check(), b.check()
    assert a.counters() == b.counters() and a.progress() == b.progress() and a.counters()["plies"] > G
    for g in (0, G - 1):
        ta, tb = a.engine.tree_dump(g), b.engine.tree_dump(g)
        for k in ("keys", "sum_n", "n"):
            assert (ta[k] == tb[k]).all(), k
        assert ta["w"].tobytes() == tb["w"].tobytes() and ta["p"].tobytes() == tb["p"].tobytes()
    assert [(e["game"], e["seq"], e["T"]) for e in _drain(a)] == [(e["game"], e["seq"], e["T"]) for e in _drain(b)]
    a.close()
    b.close()


def test_stamped_replay_times_the_kernels_inside_the_graph_and_changes_nothing():
    """ABI v6 af_engine_stamp: the stamped capture of the loop (three device-clock stamps per tick) is the same computation as the plain
    one — counters and trees equal an engine that never stamps — and its stamps are ordered, plausible durations."""
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    cfg = make_cfg(simulation_per_step=40, upper_simulation_per_step=60)
    G = 128
    nets = [ResNet(11, device="cuda") for _ in range(2)]
    for nt in nets:
        nt.load_npz(W)
    a = SelfPlayEngine(cfg, G, nets[0].select_backend("hip"), device=0, seed=4)
    b = SelfPlayEngine(cfg, G, nets[1].select_backend("hip"), device=0, seed=4)
    for r in range(12):
        a.run_ticks_graph(8, stamped=(r % 3 == 1))
    st = a.read_stamps()
    assert st.shape == (4 * 8, 2)
    assert (st > 0).all() and (st[:, 0] < 5.0).all() and (st[:, 1] < 20.0).all()          # ms: ordered stamps, sane magnitudes
    assert st[:, 1].mean() > st[:, 0].mean() * 0.5                                       # the forward is not shorter than half a tick kernel
    b.run_ticks(a.ticks)
    a.check(), b.check()
    assert a.counters() == b.counters() and a.progress() == b.progress()
    ta, tb = a.engine.tree_dump(5), b.engine.tree_dump(5)
    for k in ("keys", "sum_n", "n"):
        assert (ta[k] == tb[k]).all()
    assert ta["w"].tobytes() == tb["w"].tobytes()
    a.close(), b.close()
