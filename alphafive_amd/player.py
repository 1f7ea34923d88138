"""Drop-in for the reference's genData/player.py:Player on the HIP engine.

Same constructor, methods, argument meaning, return types and error behaviour:

    Player(cfg=None, training=True, pipe=None, pv_fn=None)      player.py:23-35
    .get_init_state() -> str                                    player.py:37
    .reset(search_tree=None)                                    player.py:48
    .run(e=0.25) -> [(state, policy[S,S], last_action, value, weight)]   player.py:53
    .get_action(state, e=0.25, last_action=None, random_a=False) -> (policy|None, (i,j))  player.py:128
    .pruning_tree(board, state=None)                            player.py:149
    .close()                                                    player.py:281
    attributes: tree, root_state, tau, training, config

The search itself (MCTS_search / select / expand / backup / calc_policy) runs in the
device engine (csrc/af_engine.hip) in EXTERNAL mode with one resident game; this class only
moves leaf planes / evaluations between the engine and the caller's pv_fn or pipe.
Noise comes from the engine's counter-based generator (include/af_noise.h), not from
np.random — seed it with the `seed` keyword.
"""
import gc
import os
from collections.abc import Mapping

import numpy as np

from . import engine as _eng
from . import utils


class _EdgeView(object):
    __slots__ = ("n", "w", "q", "p")

    def __init__(self, n, w, q, p):
        self.n, self.w, self.q, self.p = n, w, q, p


class _StateView(object):
    """Read-only stand-in for player.py:16 State: .a maps (i,j) -> edge stats, .sum_n."""

    def __init__(self, S, legal_cells, sum_n, n, w, p, f32):
        self.sum_n = int(sum_n)
        self.a = {}
        for c in legal_cells:
            nn = int(n[c])
            if nn == 0:
                q = 0
            elif f32[c]:
                q = np.float32(w[c]) / np.float32(nn)
            else:
                q = float(w[c]) / nn
            self.a[(c // S, c % S)] = _EdgeView(nn, w[c] if f32[c] else float(w[c]), q, p[c])


class TreeView(Mapping):
    """Snapshot of the device transposition store keyed by state string (player.py:29)."""

    def __init__(self, dump, S):
        self._S = S
        self._d = dump
        self._index = {_eng.key_to_state(dump["keys"][i], S): i for i in range(len(dump["sum_n"]))}

    def __getitem__(self, state):
        i = self._index[state]
        S, C = self._S, self._S * self._S
        key = self._d["keys"][i]
        kw = len(key) // 2
        occ = [(int(key[c >> 6]) | int(key[kw + (c >> 6)])) >> (c & 63) & 1 for c in range(C)]
        legal = [c for c in range(C) if not occ[c]]
        return _StateView(S, legal, self._d["sum_n"][i], self._d["n"][i], self._d["w"][i], self._d["p"][i],
                          self._d["f32"][i])

    def __iter__(self):
        return iter(self._index)

    def __len__(self):
        return len(self._index)


class Player(object):
    _next_game_id = [0]          # process-wide: every Player built without an explicit game_id gets its own noise stream

    def __init__(self, cfg=None, training=True, pipe=None, pv_fn=None, device=0, seed=None, game_id=None, node_cap=0,
                 graph=True):
        assert pipe is not None or pv_fn is not None
        import torch
        # The reference's workers draw from process-global MT19937 streams seeded from OS entropy (main.py:82: N workers each
        # `Player(config, training=True, pipe=pipe)`), so they never repeat each other.  Same here by default: the 64-bit
        # seed comes from os.urandom and game_id from a process-wide counter; pass both for reproducible runs.
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        if game_id is None:
            game_id = Player._next_game_id[0]
            Player._next_game_id[0] += 1
        self.seed, self.game_id = seed, game_id
        self.config = cfg
        self.training = training
        self.root_state = None
        self.goal = self.config.goal
        self.tau = self.config.init_temp
        self.pipe = pipe
        self.job_done = False
        self.pv_fn = pv_fn
        self._torch = torch
        self._S = cfg.board_size
        self._C = self._S * self._S
        self._dev = torch.device("cuda", device)
        # behind a pipe the values arrive as python floats (networkAPI.py:72 `float(v)`), which makes every W / Q of the
        # reference's tree fp64; through pv_fn they are np.float32 and W is an fp32 running sum (SURVEY 8a rule 2)
        self._engine = _eng.Engine(cfg, 1, device=device, mode=_eng.MODE_EXTERNAL, training=training, seed=seed,
                                   first_game_id=game_id, node_cap=node_cap, value_f64=(pv_fn is None))
        self._planes = torch.zeros((1, 3, self._S, self._S), dtype=torch.float32, device=self._dev)
        self._policy = torch.zeros((1, self._C), dtype=torch.float32, device=self._dev)
        self._value = torch.zeros((1,), dtype=torch.float32, device=self._dev)
        self._reset_pending = False
        self._adopt = None
        # pv_fn = ResNet.eval on a cuda device (self_play.py:88, choose_best_player.py:38-40): the leaves stay on the device and
        # are evaluated by the hand-written kernels straight into the tensors the tick kernel reads — this Player gets its own
        # evaluator handle (its output buffers are bound to this game).  select_backend("hip") raises when libaf_net.so is
        # missing or its ABI does not match; nothing falls back to vendor ops.  Any other owner with an eval_device() method
        # (e.g. DeepResNet) is used as it is.
        owner = getattr(pv_fn, "__self__", None)
        self._pv_owner = None
        self._pv_device = None
        self.backend = "host"                            # evaluations cross the host boundary (numpy pv_fn or pipe)
        self._use_graph = bool(graph)
        self._graph = None
        if pv_fn is not None and owner is not None and getattr(pv_fn, "__name__", "") == "eval" \
                and hasattr(owner, "eval_device") and getattr(getattr(owner, "device", None), "type", None) == "cuda":
            from .network import ResNet
            if isinstance(owner, ResNet):
                fn = owner.select_backend("hip")
                fn.bind_outputs(self._policy, self._value)
                self._pv_device, self._pv_owner, self.backend = fn, owner, "hip"
            else:
                self._pv_device, self.backend = owner.eval_device, "device"
        self.last_visits = None

    # -- player.py:37-46
    def get_init_state(self):
        return (chr(ord("a") + self.config.board_size) + "/") * self.config.board_size

    # -- player.py:48-51
    def reset(self, search_tree=None):
        self._reset_pending = True
        self._adopt = None
        if search_tree is not None and len(search_tree):
            self._adopt = self._tree_to_dump(search_tree)       # uploaded by the next get_action, after the reset
        self.root_state = None
        self.tau = self.config.init_temp

    def _tree_to_dump(self, tree):
        """A mapping state string -> State-like (.a: {(i,j): edge with n, w, p}, .sum_n) — the reference's
        defaultdict(State) or a TreeView of this class — in the engine's tree_dump format."""
        S, Cc = self._S, self._C
        K = self._engine.KW2
        states = list(tree)
        keys = np.zeros((len(states), K), np.uint64)
        sum_n = np.zeros(len(states), np.int32)
        n = np.zeros((len(states), Cc), np.int32)
        w = np.zeros((len(states), Cc), np.float64 if self._engine.value_f64 else np.float32)
        p = np.zeros((len(states), Cc), np.float32)
        f32 = np.zeros((len(states), Cc), np.uint8)
        for i, st in enumerate(states):
            node = tree[st]
            keys[i] = _eng.state_to_key(st, S)
            sum_n[i] = int(node.sum_n)
            for (a, b), e in node.a.items():
                c = a * S + b
                n[i, c], w[i, c], p[i, c] = int(e.n), float(e.w), float(e.p)
                f32[i, c] = isinstance(e.w, np.floating) and not isinstance(e.w, np.float64)   # np.float32 running sum (SURVEY rule 2)
        return dict(keys=keys, sum_n=sum_n, n=n, w=w, p=p, f32=f32)

    @property
    def tree(self):
        if self._reset_pending and self._adopt is not None:
            return TreeView(self._adopt, self._S)
        if self._reset_pending:
            return TreeView(dict(keys=np.zeros((0, self._engine.KW2), np.uint64), sum_n=np.zeros(0, np.int32),
                                 n=None, w=None, p=None, f32=None), self._S)
        return TreeView(self._engine.tree_dump(0), self._S)

    # -- player.py:186-197: the evaluation plug-in (pv_fn or pipe)
    def _evaluate_leaf(self):
        if self._pv_device is not None:
            p, v = self._pv_device(self._planes)
            if p.data_ptr() != self._policy.data_ptr():
                self._policy.copy_(p.reshape(1, self._C))
                self._value.copy_(v.reshape(1))
            return
        x = self._planes.cpu().numpy()
        if self.pv_fn is not None:
            policy, value = self.pv_fn(x)
            policy, value = policy[0], value[0]
        else:
            self.pipe.send([x[0]])
            while not self.pipe.poll(0.001):       # player.py:195-196 busy-waits on poll(); blocking in select() instead leaves
                pass                               # the core (and, with the NetworkAPI thread in this process, the GIL) to the server
            policy, value = self.pipe.recv()[0]
        self._policy.copy_(self._torch.from_numpy(np.ascontiguousarray(policy, np.float32).reshape(1, self._C)))
        self._value.copy_(self._torch.from_numpy(np.asarray([value], np.float32)))

    def _weights_version(self):
        return getattr(self._pv_owner, "version", None)

    def _search_batch(self, n=16):
        """n x (tick kernel -> leaf evaluation) without touching the host.  One game is launch-latency bound (14 dependent
        launches per simulation), so the batch is captured once into a HIP graph and replayed.  The graph is keyed on the
        training flag AND on the owner's weight version: a replay skips the Python evaluator wrapper — the only place that
        notices net.restore() / load_npz() / set_variables() and re-packs the weights for the kernels — so after a weight
        update (choose_best_player.py:78-82 reloads both nets between matches) the graph is dropped, one eager evaluation
        reloads the weights and the batch is captured again.  A failing capture raises (Player(graph=False) is the explicit
        way to run the same launches eagerly)."""
        torch = self._torch
        if not self._use_graph:
            stream = torch.cuda.current_stream(self._dev).cuda_stream
            for _ in range(n):
                self._engine.tick(self._policy.data_ptr(), self._value.data_ptr(), self._planes.data_ptr(), stream)
                self._evaluate_leaf()
            return
        # ... and on every host-mutable launch parameter (EngineParams travel by value: training flag, the simulation budget the
        # reference re-reads from the live config at every get_action (player.py:140-143), the per-launch select budget)
        key = (self._engine.params_key(), self._weights_version())
        if self._graph is None or self._graph[0] != key:
            self._graph = None
            cur = torch.cuda.current_stream(self._dev)
            self._engine.tick(self._policy.data_ptr(), self._value.data_ptr(), self._planes.data_ptr(), cur.cuda_stream)
            self._evaluate_leaf()                    # warm-up outside the capture (weight reload, lazy allocations)
            torch.cuda.synchronize(self._dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):     # (other threads — an RCCL watchdog — may touch the runtime)
                st = torch.cuda.current_stream(self._dev).cuda_stream
                for _ in range(n):
                    self._engine.tick(self._policy.data_ptr(), self._value.data_ptr(), self._planes.data_ptr(), st)
                    self._evaluate_leaf()
            self._graph = (key, g)
            return                                   # (the warm-up pair already advanced the search by one tick)
        self._graph[1].replay()

    # -- player.py:128-147
    def get_action(self, state, e=0.25, last_action=None, random_a=False):
        self.root_state = state
        S = self._S
        key = _eng.state_to_key(state, S)
        last_cell = -1 if last_action is None else last_action[0] * S + last_action[1]
        self._engine.set_training(self.training)
        # player.py:140-143 reads the budget from the live config object at every call
        self._engine.set_simulations(self.config.simulation_per_step, self.config.upper_simulation_per_step)
        self._engine.set_root(0, key, last_cell, random_a, reset_tree=self._reset_pending)
        self._reset_pending = False
        if getattr(self, "_adopt", None) is not None:          # reset(search_tree): the adopted tree replaces the fresh store
            self._engine.load_tree(0, self._adopt)
            self._adopt = None
        stream = self._torch.cuda.current_stream(self._dev).cuda_stream
        # device evaluator (pv_fn is a ResNet.eval bound method): nothing has to cross the host boundary per leaf, so
        # ticks and evaluations are queued 16 at a time and the status word is read once per batch (a game whose move
        # is decided ignores further ticks)
        while self._pv_device is not None:
            self._search_batch()
            if int(self._engine.status(stream)[0]) == _eng.STATUS_MOVE_DONE:
                break
        while self._pv_device is None:
            self._engine.tick(self._policy.data_ptr(), self._value.data_ptr(), self._planes.data_ptr(), stream)
            st = int(self._engine.status(stream)[0])
            if st == _eng.STATUS_NEED_EVAL:
                self._evaluate_leaf()
            elif st == _eng.STATUS_MOVE_DONE:
                break
            elif st == _eng.STATUS_YIELD:          # per-launch work budget used up: nothing to evaluate
                continue
            else:
                raise _eng.EngineError(f"unexpected engine status {st}")
        action, policy, visits, tau = self._engine.move_result(0)
        self.tau = tau
        self.last_visits = visits
        if policy is not None:
            policy = policy.reshape(S, S)
        return policy, (action // S, action % S)

    # -- player.py:53-82
    def run(self, e=0.25):
        S = self._S
        state = self.get_init_state()
        game_over = False
        data = []
        value = 0
        last_action = None
        while not game_over:
            policy, action = self.get_action(state, e, last_action)
            data.append((state, policy, last_action))
            board = utils.step(utils.state_to_board(state, S), action)
            state = utils.board_to_state(board)
            game_over, value = utils.is_game_over(board, self.goal)
            last_action = action
        self.reset()
        turns = len(data)
        if turns % 2 == 1:
            value = -value
        weights = utils.construct_weights(turns, gamma=self.config.gamma)
        final_data = []
        for i in range(turns):
            final_data.append((*data[i], value, weights[i]))
            value = -value
        return final_data

    # -- player.py:149-164: the reference's pruning only frees memory; the device store compacts
    # itself under capacity pressure, so this is an API-compatible no-op.
    def pruning_tree(self, board, state=None):
        return None

    # -- player.py:281-284
    def close(self):
        self.job_done = True
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        gc.collect()
