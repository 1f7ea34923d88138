#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by importing the UNMODIFIED
reference from /root/reference (build container only; the reference never travels).

    python tests/golden/make_golden.py

What is recorded (data only — inputs and expected outputs):
  rules.npz      utils.py known answers: boards -> state strings, is_game_over,
                 get_legal_actions, board_to_inputs, step, construct_weights
  mcts_*.npz     genData.player.Player driven under np.random.seed(s); random.seed(s)
                 with the integer pseudo-net: per-ply states, actions, root
                 visit-count vectors, policies, and the complete final tree
                 (sum_n, n, w, dtype class of w, p per node)
  run_*.npz      Player.run() episodes: the 5-tuples + main.gen_data's result code
The pseudo-net (tests/pseudonet.py) is the build's own deterministic stand-in for
ResNet.eval; it is handed to the reference through its pv_fn seam (player.py:24).
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, "/root/reference")

import config as refcfg            # noqa: E402  (reference)
import utils as refutils           # noqa: E402  (reference)
from genData.player import Player  # noqa: E402  (reference)

import pseudonet                   # noqa: E402  (build's own)


def mkcfg(**kw):
    c = types.SimpleNamespace(**{k: getattr(refcfg, k) for k in dir(refcfg)
                                 if not k.startswith("_") and k != "get_lr"})
    for k, v in kw.items():
        setattr(c, k, v)
    return c


CFG_KEYS = ["board_size", "goal", "simulation_per_step", "upper_simulation_per_step", "init_temp", "gamma",
            "tau_decay_rate", "tau_decay_rate_r", "dirichlet_alpha", "c_puct"]


def cfg_arrays(cfg):
    return {"cfg_" + k: np.asarray(getattr(cfg, k)) for k in CFG_KEYS}


def dump_tree(player, S):
    C = S * S
    keys = list(player.tree.keys())
    n_nodes = len(keys)
    sum_n = np.zeros(n_nodes, np.int32)
    n = np.zeros((n_nodes, C), np.int32)
    w = np.zeros((n_nodes, C), np.float64)
    wf32 = np.zeros((n_nodes, C), np.uint8)
    q = np.zeros((n_nodes, C), np.float64)
    p = np.zeros((n_nodes, C), np.float32)
    legal = np.zeros((n_nodes, C), np.uint8)
    for k, key in enumerate(keys):
        st = player.tree[key]
        sum_n[k] = st.sum_n
        for (i, j), a in st.a.items():
            c = i * S + j
            legal[k, c] = 1
            n[k, c] = a.n
            w[k, c] = float(a.w)
            wf32[k, c] = isinstance(a.w, np.float32)
            q[k, c] = float(a.q)
            p[k, c] = a.p
    return dict(tree_keys=np.array(keys), tree_sum_n=sum_n, tree_n=n, tree_w=w, tree_wf32=wf32, tree_q=q,
                tree_p=p, tree_legal=legal)


def node_digest(sum_n, n, w, wf32, p, legal):
    """8-byte digest of one tree node (edge arrays dense by cell, illegal cells zero): lets a 20,000-node tree of an
    11x11 / 500-simulation trace be pinned in a few hundred KB instead of 60 MB of (mostly zero) edge arrays."""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    h.update(np.int32(sum_n).tobytes())
    h.update(np.ascontiguousarray(n, np.int32).tobytes())
    h.update(np.ascontiguousarray(w, np.float64).tobytes())
    h.update(np.ascontiguousarray(np.where(legal, wf32, 0), np.uint8).tobytes())
    h.update(np.ascontiguousarray(p, np.float32).tobytes())
    return np.frombuffer(h.digest(), np.uint64)[0]


def digest_tree(d):
    """dump_tree() output -> keys, sum_n and one digest per node."""
    dg = np.array([node_digest(d["tree_sum_n"][k], d["tree_n"][k], d["tree_w"][k], d["tree_wf32"][k], d["tree_p"][k],
                               d["tree_legal"][k].astype(bool)) for k in range(len(d["tree_sum_n"]))], np.uint64)
    return dict(tree_keys=d["tree_keys"], tree_sum_n=d["tree_sum_n"], tree_digest=dg)


def gen_rules(path):
    rng = np.random.RandomState(7)
    boards, states, over, value, goals = [], [], [], [], []
    cases = []
    for S, goal in [(11, 5), (15, 5), (7, 4), (6, 4), (3, 3)]:
        for t in range(60):
            fill = rng.rand()
            b = rng.choice([-1, 0, 1], size=(S, S), p=[fill / 2, 1 - fill, fill / 2]).astype(np.int8)
            cases.append((b, goal))
        # crafted lines in every direction, both colours, at edges, overlines, truncated windows
        for colour in (1, -1):
            for length in (goal - 1, goal, goal + 1):
                for (di, dj) in [(1, 0), (0, 1), (1, 1), (-1, 1)]:
                    for _ in range(3):
                        b = np.zeros((S, S), np.int8)
                        i0 = rng.randint(0, S)
                        j0 = rng.randint(0, S)
                        ok = True
                        for k in range(length):
                            i, j = i0 + di * k, j0 + dj * k
                            if not (0 <= i < S and 0 <= j < S):
                                ok = False
                                break
                            b[i, j] = colour
                        if ok:
                            cases.append((b, goal))
        full = rng.choice([-1, 1], size=(S, S)).astype(np.int8)
        cases.append((full, goal))
        # full board without a line (stripe pattern) — draw
        stripe = np.fromfunction(lambda i, j: ((i // 2 + j) % 2) * 2 - 1, (S, S)).astype(np.int8)
        cases.append((stripe, goal))
        both = np.zeros((S, S), np.int8)
        if S >= goal + 1:
            both[0, :goal] = -1
            both[1, :goal] = 1
            cases.append((both, goal))
            cases.append((-both, goal))
    rec = dict(S=[], goal=[], board=[], state=[], over=[], value=[], legal_count=[])
    legal_cells, inputs, stepped = [], [], []
    for b, goal in cases:
        S = b.shape[0]
        rec["S"].append(S)
        rec["goal"].append(goal)
        pad = np.zeros((15, 15), np.int8)
        pad[:S, :S] = b
        rec["board"].append(pad)
        s = refutils.board_to_state(b)
        assert (refutils.state_to_board(s, S) == b).all()
        rec["state"].append(s)
        o, v = refutils.is_game_over(b, goal)
        rec["over"].append(o)
        rec["value"].append(v)
        la = refutils.get_legal_actions(b)
        rec["legal_count"].append(len(la))
        lc = np.full(225, -1, np.int32)
        lc[:len(la)] = [i * S + j for i, j in la]
        legal_cells.append(lc)
        last = la[len(la) // 2] if la else None
        x = refutils.board_to_inputs(b, last_action=last)
        xp = np.zeros((3, 15, 15), np.float32)
        xp[:, :S, :S] = x
        inputs.append(xp)
        sp = np.zeros((15, 15), np.int8)
        if la:
            sp[:S, :S] = refutils.step(b.copy(), la[0])
        stepped.append(sp)
    cw = {f"cw_{T}_{g}": refutils.construct_weights(T, gamma=g) for T in (1, 2, 5, 9, 26, 64, 121, 200)
          for g in (0.94, 0.95)}
    np.savez_compressed(path, S=np.array(rec["S"]), goal=np.array(rec["goal"]), board=np.array(rec["board"]),
                        state=np.array(rec["state"]), over=np.array(rec["over"]), value=np.array(rec["value"]),
                        legal_count=np.array(rec["legal_count"]), legal_cells=np.array(legal_cells),
                        inputs=np.array(inputs), stepped=np.array(stepped), **cw)
    print("rules:", len(cases), "cases")


class _FakeAgent:
    """Stands where ResNet stands in networkAPI.py:64-65 (`with agent_model.graph.as_default(): agent_model.eval(data)`)."""

    class _Graph:
        def as_default(self):
            import contextlib
            return contextlib.nullcontext()

    def __init__(self, salt, peak, vbits):
        self.graph, self.salt, self.peak, self.vbits = self._Graph(), salt, peak, vbits

    def eval(self, data):
        return pseudonet.pseudonet_np(data, self.salt, self.peak, self.vbits)


def gen_mcts(path, S, goal, sims, upper, training, seed, salt, peak, max_plies, random_a=False, reset_every=None, pipe=False,
             vbits=16, digest=False, sharp=0, **cfg_kw):
    cfg = mkcfg(board_size=S, goal=goal, simulation_per_step=sims, upper_simulation_per_step=upper, **cfg_kw)
    np.random.seed(seed)
    random.seed(seed)
    api = None
    if pipe:
        # the path main.py's workers run (main.py:82-94 -> networkAPI.py:43-78): the UNMODIFIED NetworkAPI thread answers
        # over a multiprocessing Pipe with (policy row, float(v)): w and q become python floats (SURVEY 8a rule 2)
        from genData.networkAPI import NetworkAPI
        api = NetworkAPI(cfg, _FakeAgent(salt, peak, vbits))
        api.start(reload=False)
        pl = Player(cfg, training=training, pipe=api.get_pipe(reload=False))
    else:
        pl = Player(cfg, training=training, pv_fn=lambda x: pseudonet.pseudonet_np(x, salt, peak, vbits, sharp))
    state = pl.get_init_state()
    last, over, ply = None, False, 0
    C = S * S
    states, actions, lasts, visits, policies, has_pol, taus = [], [], [], [], [], [], []
    while not over and ply < max_plies:
        pol, act = pl.get_action(state, last_action=last, random_a=random_a)
        node = pl.tree[state]
        v = np.zeros(C, np.int32)
        for (i, j), a in node.a.items():
            v[i * S + j] = a.n
        states.append(state)
        actions.append(act[0] * S + act[1])
        lasts.append(-1 if last is None else last[0] * S + last[1])
        visits.append(v)
        has_pol.append(pol is not None)
        policies.append(np.zeros(C, np.float32) if pol is None else pol.reshape(-1))
        taus.append(pl.tau)
        board = refutils.step(refutils.state_to_board(state, S), act)
        state = refutils.board_to_state(board)
        over, _ = refutils.is_game_over(board, goal)
        last = act
        ply += 1
    # position of both MT streams after the run (pins the number of draws consumed)
    np_next = int(np.random.randint(0, 2 ** 32, dtype=np.uint64))
    py_next = random.getrandbits(32)
    out = dict(training=np.asarray(training), seed=np.asarray(seed), salt=np.asarray(salt), peak=np.asarray(peak),
               random_a=np.asarray(random_a), states=np.array(states), actions=np.array(actions, np.int32),
               lasts=np.array(lasts, np.int32), visits=np.array(visits), policies=np.array(policies),
               has_policy=np.array(has_pol), taus=np.array(taus), np_next=np.asarray(np_next),
               py_next=np.asarray(py_next), finished=np.asarray(over), pipe=np.asarray(bool(pipe)), vbits=np.asarray(vbits),
               sharp=np.asarray(sharp))
    out.update(cfg_arrays(cfg))
    out.update(digest_tree(dump_tree(pl, S)) if digest else dump_tree(pl, S))
    if api is not None:
        api.done = True
    np.savez_compressed(path, **out)
    print(os.path.basename(path), "plies", ply, "nodes", len(pl.tree), "over", over)


def gen_run(path, S, goal, sims, upper, seed, salt, peak, episodes, **cfg_kw):
    cfg = mkcfg(board_size=S, goal=goal, simulation_per_step=sims, upper_simulation_per_step=upper, **cfg_kw)
    np.random.seed(seed)
    random.seed(seed)
    pl = Player(cfg, training=True, pv_fn=lambda x: pseudonet.pseudonet_np(x, salt, peak))
    C = S * S
    out = dict(seed=np.asarray(seed), salt=np.asarray(salt), peak=np.asarray(peak), episodes=np.asarray(episodes))
    out.update(cfg_arrays(cfg))
    for e in range(episodes):
        rec = pl.run()
        # main.py:85-93 gen_data result code
        value = rec[-1][-2]
        if value == 0.0:
            result = refutils.DRAW
        elif len(rec) % 2 == 1:
            result = refutils.BLACK_WIN
        else:
            result = refutils.WHITE_WIN
        out[f"ep{e}_states"] = np.array([r[0] for r in rec])
        out[f"ep{e}_policies"] = np.array([r[1].reshape(-1) for r in rec], np.float32)
        out[f"ep{e}_lasts"] = np.array([-1 if r[2] is None else r[2][0] * S + r[2][1] for r in rec], np.int32)
        out[f"ep{e}_values"] = np.array([r[3] for r in rec], np.float64)
        out[f"ep{e}_weights"] = np.array([r[4] for r in rec], np.float32)
        out[f"ep{e}_result"] = np.asarray(result)
        assert all(isinstance(r[4], np.float32) for r in rec) and all(isinstance(r[3], float) for r in rec)
        assert len(pl.tree) == 0
        print(os.path.basename(path), "episode", e, "T", len(rec), "result", result)
    out["np_next"] = np.asarray(int(np.random.randint(0, 2 ** 32, dtype=np.uint64)))
    out["py_next"] = np.asarray(random.getrandbits(32))
    np.savez_compressed(path, **out)


def gen_randomstack(path):
    """utils.RandomStack (utils.py:14-146): seeded pushes of synthetic episodes, then seeded get_data."""
    import io
    import contextlib
    S = 7
    rng = np.random.RandomState(5)
    episodes = []
    for e in range(60):
        T = int(rng.randint(9, 40))
        recs = []
        board = np.zeros((S, S), np.int8)
        last = None
        for t in range(T):
            empt = refutils.get_legal_actions(board)
            a = empt[rng.randint(len(empt))]
            pol = rng.rand(S, S).astype(np.float32)
            pol /= pol.sum()
            recs.append((refutils.board_to_state(board), pol, last, float((-1.0) ** (T - t)), np.float32(rng.rand() + 0.5)))
            board = refutils.step(board, a)
            last = a
        episodes.append((recs, int(rng.choice([1, -1, 0], p=[0.55, 0.4, 0.05]))))
    np.random.seed(21)
    random.seed(21)
    st = refutils.RandomStack(board_size=S, length=400)
    accepted = []
    with contextlib.redirect_stdout(io.StringIO()):
        for recs, res in episodes:
            accepted.append(st.push(recs, res))
    out = dict(S=np.asarray(S), length=np.asarray(400), accepted=np.array(accepted),
               data_len=np.array(st.data_len), result=np.array(st.result), n_data=np.asarray(len(st.data)),
               black_win=np.asarray(st.black_win), white_win=np.asarray(st.white_win),
               first_state=np.array(st.data[0][0]), last_state=np.array(st.data[-1][0]))
    for e, (recs, res) in enumerate(episodes):
        out[f"ep{e}_states"] = np.array([r[0] for r in recs])
        out[f"ep{e}_policies"] = np.array([r[1] for r in recs])
        out[f"ep{e}_lasts"] = np.array([-1 if r[2] is None else r[2][0] * S + r[2][1] for r in recs], np.int32)
        out[f"ep{e}_values"] = np.array([r[3] for r in recs])
        out[f"ep{e}_weights"] = np.array([r[4] for r in recs], np.float32)
        out[f"ep{e}_result"] = np.asarray(res)
    out["n_episodes"] = np.asarray(len(episodes))
    for b in range(3):
        boards, weights, values, policies = st.get_data(batch_size=48)
        out[f"batch{b}_boards"], out[f"batch{b}_weights"] = boards, weights
        out[f"batch{b}_values"], out[f"batch{b}_policies"] = values, policies
    out["np_next"] = np.asarray(int(np.random.randint(0, 2 ** 32, dtype=np.uint64)))
    out["py_next"] = np.asarray(random.getrandbits(32))
    np.savez_compressed(path, **out)
    print("randomstack:", sum(accepted), "accepted of", len(episodes), "buffer", len(st.data))


def gen_edge_cases(G):
    """Edge cases of the path: drawn games on a full board (3x3, goal 3: value 0.0 stays a python float),
    the tau <= 0.01 branch of calc_policy (player.py:112-115), a simulation cap that leaves few or no
    simulations (upper - sum_n <= 0, player.py:143), other c_puct / alpha / decay values."""
    gen_run(G("run_s3_draws.npz"), 3, 3, 30, 40, 4, 900, 0, 6)       # seed 4: two of the six games are draws
    gen_run(G("run_s4_lowtau.npz"), 4, 3, 24, 30, 22, 901, 4096, 3, init_temp=0.0105)
    gen_mcts(G("mcts_s5_cap.npz"), 5, 4, 30, 31, True, 23, 902, 16384, 40)
    gen_mcts(G("mcts_s6_params.npz"), 6, 4, 50, 70, True, 24, 903, 8192, 40, c_puct=1.5, dirichlet_alpha=0.15,
             tau_decay_rate=0.8, init_temp=2.0)


def gen_pipe_cases(G):
    """The pipe path (what main.py runs): values are python floats, so every W is an fp64 running sum and Q = W/N in fp64."""
    # (sims > 2L so that simulations beyond the forced root visits exist and the values steer the search)
    gen_mcts(G("mcts_s6_train_pipe.npz"), 6, 4, 120, 160, True, 3, 1237, 16384, 40, pipe=True, vbits=24)
    gen_mcts(G("mcts_s11_train_pipe.npz"), 11, 5, 300, 400, True, 9, 4321, 8192, 3, pipe=True, vbits=24)
    gen_mcts(G("mcts_s6_eval_pipe.npz"), 6, 4, 100, 120, False, 4, 77, 16384, 40, pipe=True, vbits=24)
    # the same 24-bit values through the pv_fn seam: fp32 running sums that DO round (the 16-bit pseudo-net never does)
    gen_mcts(G("mcts_s6_train_v24.npz"), 6, 4, 120, 160, True, 3, 1237, 16384, 40, vbits=24)


def gen_fullsize_cases(G):
    """BASELINE.json's own search settings (11x11, 500 simulations per move, cap 642 — the `metric`), training mode, 24
    plies: ~12,000 simulations through the unmodified reference Player; the tree is stored as one digest per node."""
    gen_mcts(G("mcts_s11_train_500.npz"), 11, 5, 500, 642, True, 17, 2024, 8192, 24, digest=True)


def gen_sharp_cases(G):
    """Eval-mode games (self_play.py:79-106 / choose_best_player.py:52: training=False) with the sharp pseudo-net
    (tests/pseudonet.py, sharp=1, 24-bit values): no noise is applied in eval mode (player.py:247), so the only random
    decisions are uniform picks among tied candidates (:277-279, :101-102) — and with priors that never coincide these
    traces have (almost) none: tests/test_gpu_reference_fixtures.py compares the HIP Player's visit counts and moves with
    THESE reference-generated arrays directly, no oracle in between.  11x11 at BASELINE's own 500 / 642."""
    # salts picked (with the C oracle in MT mode, which replays these games bit for bit) among those whose game has no score tie at
    # any select: 11x11 one max-visit tie (ply 12: changes the move picked, not the search), the other two none at all
    gen_mcts(G("mcts_s11_eval_sharp.npz"), 11, 5, 500, 642, False, 31, 20312, 0, 121, vbits=24, sharp=1, digest=True)
    gen_mcts(G("mcts_s6_eval_sharp.npz"), 6, 4, 100, 120, False, 32, 20251, 0, 36, vbits=24, sharp=1)
    gen_mcts(G("mcts_s15_eval_sharp.npz"), 15, 5, 200, 260, False, 33, 20276, 0, 12, vbits=24, sharp=1, digest=True)


def main():
    if "--only-sharp" in sys.argv:
        gen_sharp_cases(lambda name: os.path.join(HERE, name))
        return
    if "--only-fullsize" in sys.argv:
        gen_fullsize_cases(lambda name: os.path.join(HERE, name))
        return
    if "--only-edges" in sys.argv:
        gen_edge_cases(lambda name: os.path.join(HERE, name))
        return
    if "--only-pipe" in sys.argv:
        gen_pipe_cases(lambda name: os.path.join(HERE, name))
        return
    if "--only-randomstack" in sys.argv:
        gen_randomstack(os.path.join(HERE, "randomstack.npz"))
        return
    gen_rules(os.path.join(HERE, "rules.npz"))
    gen_randomstack(os.path.join(HERE, "randomstack.npz"))
    G = lambda name: os.path.join(HERE, name)  # noqa: E731
    # training mode, forced root visits not exhausted (sims < 2L) and exhausted (sims > 2L)
    gen_mcts(G("mcts_s11_train_a.npz"), 11, 5, 60, 80, True, 0, 1234, 0, 12)
    gen_mcts(G("mcts_s11_train_b.npz"), 11, 5, 300, 400, True, 5, 99, 8192, 3)
    gen_mcts(G("mcts_s6_train.npz"), 6, 4, 120, 160, True, 3, 1237, 16384, 40)
    gen_mcts(G("mcts_s7_train.npz"), 7, 4, 50, 70, True, 1, 1235, 4096, 40)
    # eval mode (self_play.py loop: training=False, tree shared, never reset)
    gen_mcts(G("mcts_s11_eval.npz"), 11, 5, 80, 100, False, 2, 1236, 8192, 14)
    gen_mcts(G("mcts_s6_eval.npz"), 6, 4, 100, 120, False, 4, 77, 16384, 40)
    # arena mode (choose_best_player.py: training=False, random_a=True)
    gen_mcts(G("mcts_s7_arena.npz"), 7, 4, 60, 80, False, 6, 78, 4096, 40, random_a=True)
    # 15x15
    gen_mcts(G("mcts_s15_train.npz"), 15, 5, 40, 60, True, 8, 5, 8192, 4)
    # whole episodes through Player.run (tree reset, value signs, weights, result code)
    gen_run(G("run_s6.npz"), 6, 4, 60, 80, 11, 4242, 16384, 3)
    gen_run(G("run_s7.npz"), 7, 4, 40, 60, 12, 4243, 4096, 2)
    gen_edge_cases(G)
    gen_pipe_cases(G)
    gen_fullsize_cases(G)
    gen_sharp_cases(G)


if __name__ == "__main__":
    main()
