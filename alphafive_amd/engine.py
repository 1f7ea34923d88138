"""ctypes binding of the C-ABI HIP engine (include/af_engine.h) and the batched
self-play driver that replaces main.py:82-94 gen_data + genData/networkAPI.py.

No CPU fallback: if libaf_hip.so is missing or fails to load this module raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("AF_HIP_LIB") or os.path.join(_PKG, "_lib", "libaf_hip.so")

MODE_SELFPLAY, MODE_EXTERNAL = 0, 1
MODE_VALUE_F64 = 0x100       # OR into mode: the reference's pipe path (networkAPI.py:72 float(v)): W, Q fp64
STATUS_IDLE, STATUS_NEED_EVAL, STATUS_MOVE_DONE, STATUS_YIELD = 0, 1, 2, 3
STAMP_SLOTS = 64             # af_engine.h AF_STAMP_SLOTS
BLACK_WIN, WHITE_WIN, DRAW = 1, -1, 0          # utils.py:9-11

CFG_ATTRS = ("board_size", "goal", "simulation_per_step", "upper_simulation_per_step", "init_temp", "gamma",
             "tau_decay_rate", "tau_decay_rate_r", "dirichlet_alpha", "c_puct")


class AfConfig(C.Structure):
    _fields_ = [("board_size", C.c_int32), ("goal", C.c_int32), ("simulation_per_step", C.c_int32),
                ("upper_simulation_per_step", C.c_int32), ("init_temp", C.c_double), ("gamma", C.c_double),
                ("tau_decay_rate", C.c_double), ("tau_decay_rate_r", C.c_double), ("dirichlet_alpha", C.c_double),
                ("c_puct", C.c_double)]

    @classmethod
    def from_cfg(cls, cfg):
        """cfg: the reference's config module or any attribute bag with its names (player.py reads
        board_size, goal, init_temp, gamma, tau_decay_rate(_r), simulation_per_step,
        upper_simulation_per_step, dirichlet_alpha, c_puct)."""
        return cls(*[getattr(cfg, a) for a in CFG_ATTRS])


class EngineError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libaf_hip.so (built by alphafive_amd.build / __graft_entry__.build). Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise EngineError(f"{_LIBPATH} not found: build the HIP engine first "
                              f"(python -m alphafive_amd.build). There is no CPU fallback.")
        # torch first: its wheel bundles its own HIP runtime; a process in which /opt/rocm's copy was pulled in earlier (by this
        # library's DT_NEEDED) ends up with two runtimes, and the second one finds no device
        import torch  # noqa: F401
        L = C.CDLL(_LIBPATH)
        vp = C.c_void_p
        i32p, f32p, u64p, u8p = (C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint8))
        L.af_abi_version.restype = C.c_int
        L.af_strerror.restype = C.c_char_p
        L.af_strerror.argtypes = [C.c_int]
        L.af_engine_create.argtypes = [C.POINTER(AfConfig), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                       C.c_uint32, C.c_int32, C.POINTER(vp)]
        L.af_engine_destroy.argtypes = [vp]
        L.af_engine_destroy.restype = None
        for fn in ("af_engine_num_games", "af_engine_cells", "af_engine_key_words", "af_engine_max_plies"):
            getattr(L, fn).argtypes = [vp]
            getattr(L, fn).restype = C.c_int32
        L.af_engine_tick.argtypes = [vp, vp, vp, vp, vp]
        L.af_engine_status.argtypes = [vp, vp, i32p]
        L.af_engine_set_root.argtypes = [vp, C.c_int32, u64p, C.c_int32, C.c_int32, C.c_int32]
        L.af_engine_move_result.argtypes = [vp, C.c_int32, i32p, i32p, f32p, i32p, C.POINTER(C.c_double)]
        L.af_engine_set_training.argtypes = [vp, C.c_int32]
        L.af_engine_set_simulations.argtypes = [vp, C.c_int32, C.c_int32]
        L.af_engine_set_roots.argtypes = [vp, vp, C.c_int32, i32p, u64p, i32p, i32p, i32p]
        L.af_engine_move_results.argtypes = [vp, vp, C.c_int32, i32p, i32p, i32p, f32p, i32p, C.POINTER(C.c_double)]
        L.af_engine_pop_episodes.argtypes = [vp, vp, C.c_int32, i32p, f32p, u64p, f32p, i32p, i32p, i32p]
        L.af_engine_pack_ints.argtypes = [vp, C.c_int32, C.c_int32]
        L.af_engine_pack_ints.restype = C.c_int64
        L.af_engine_pack_episodes.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
        L.af_engine_counters.argtypes = [vp, vp, u64p]
        L.af_engine_tree_w64.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        L.af_engine_set_tree_w64.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        L.af_engine_progress.argtypes = [vp, vp, u64p]
        L.af_engine_progress_async.argtypes = [vp, vp, vp]
        L.af_engine_stamp.argtypes = [vp, vp, C.c_int32]
        L.af_engine_stamps_async.argtypes = [vp, vp, vp]
        L.af_engine_tick_histogram.argtypes = [vp, vp, u64p, C.c_int32]
        L.af_engine_set_tick_budget.argtypes = [vp, C.c_int32]
        L.af_engine_memo_enable.argtypes = [vp, C.c_int32, C.c_int32]
        L.af_engine_memo_insert.argtypes = [vp, vp, vp, vp]
        L.af_engine_memo_clear.argtypes = [vp, vp]
        L.af_engine_memo_stats.argtypes = [vp, vp, u64p]
        L.af_engine_memo_epoch.argtypes = [vp]
        L.af_engine_memo_epoch.restype = C.c_int64
        L.af_engine_tree_dump.argtypes = [vp, C.c_int32, C.c_int32, u64p, i32p, i32p, f32p, f32p, u8p]
        L.af_engine_load_tree.argtypes = [vp, C.c_int32, C.c_int32, u64p, i32p, i32p, f32p, f32p, u8p]
        L.af_state_to_key.argtypes = [C.c_char_p, C.c_int32, u64p]
        L.af_key_to_state.argtypes = [u64p, C.c_int32, C.c_char_p, C.c_int32]
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise EngineError(f"{what}: {lib().af_strerror(rc).decode()} (code {rc})")
    return rc


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def key_words(S):
    return 4 if S * S <= 128 else 8


def state_to_key(state, S):
    key = np.zeros(key_words(S), np.uint64)
    _check(lib().af_state_to_key(state.encode(), S, _p(key, C.c_uint64)), "af_state_to_key")
    return key


def key_to_state(key, S):
    key = np.ascontiguousarray(key, np.uint64)
    out = C.create_string_buffer(288)
    _check(lib().af_key_to_state(_p(key, C.c_uint64), S, out, 288), "af_key_to_state")
    return out.value.decode()


COUNTER_NAMES = ("sims", "selects", "expands", "terminals", "plies", "episodes", "legal_sum", "legal_sum_expand",
                 "nodes", "collector_runs", "collector_scanned", "yields", "stalls")


class Engine:
    """Thin RAII wrapper over af_engine_* (one handle per GPU)."""

    def __init__(self, cfg, num_games, device=0, mode=MODE_SELFPLAY, training=True, seed=0, first_game_id=0,
                 node_cap=0, value_f64=False):
        self._h = C.c_void_p()
        self.cfg = AfConfig.from_cfg(cfg) if not isinstance(cfg, AfConfig) else cfg
        self.value_f64 = bool(value_f64) or bool(mode & MODE_VALUE_F64)
        mode = (mode & ~MODE_VALUE_F64) | (MODE_VALUE_F64 if self.value_f64 else 0)
        _check(lib().af_engine_create(C.byref(self.cfg), num_games, device, mode, int(training), seed, first_game_id,
                                      node_cap, C.byref(self._h)), "af_engine_create")
        self.G = num_games
        self.S = self.cfg.board_size
        self.C = self.S * self.S
        self.KW2 = lib().af_engine_key_words(self._h)
        self.max_plies = lib().af_engine_max_plies(self._h)
        self.mode = mode & ~MODE_VALUE_F64
        self.device = device
        self._status = np.zeros(num_games, np.int32)
        # host-mutable fields of the by-value EngineParams a launch is issued with (training flag, simulation budget, per-launch
        # select budget): a captured HIP graph has them baked in, so whoever replays one keys it on params_key()
        self._params = {"training": int(bool(training)), "sims": int(self.cfg.simulation_per_step),
                        "upper": int(self.cfg.upper_simulation_per_step), "tick_budget": None}

    def close(self):
        if getattr(self, "_h", None):
            lib().af_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def tick(self, policy_ptr, value_ptr, planes_ptr, stream=None):
        """Device pointers (ints): policy float32[G,C], value float32[G] (consumed by parked games),
        planes float32[G,3,S,S] (written for every game that parks on a new leaf)."""
        _check(lib().af_engine_tick(self._h, stream, policy_ptr, value_ptr, planes_ptr), "af_engine_tick")

    def status(self, stream=None):
        rc = lib().af_engine_status(self._h, stream, _p(self._status, C.c_int32))
        _check(rc, "engine status")
        return self._status

    def params_key(self):
        """Everything set_training / set_simulations / set_tick_budget can change: part of the key of any captured graph."""
        p = self._params
        return (p["training"], p["sims"], p["upper"], p["tick_budget"], p.get("memo_epoch"), p.get("memo"))

    def set_training(self, training):
        _check(lib().af_engine_set_training(self._h, int(training)), "af_engine_set_training")
        self._params["training"] = int(bool(training))

    def set_simulations(self, sims, upper):
        """Budget of the moves that start from now on (the reference reads config.simulation_per_step at every get_action)."""
        _check(lib().af_engine_set_simulations(self._h, int(sims), int(upper)), "af_engine_set_simulations")
        self._params["sims"], self._params["upper"] = int(sims), int(upper)

    def set_root(self, game, key, last_cell=-1, random_a=False, reset_tree=False):
        key = np.ascontiguousarray(key, np.uint64)
        _check(lib().af_engine_set_root(self._h, game, _p(key, C.c_uint64), last_cell, int(random_a), int(reset_tree)),
               "af_engine_set_root")

    def move_result(self, game):
        a, hp, tau = C.c_int32(-1), C.c_int32(0), C.c_double(0)
        pol = np.zeros(self.C, np.float32)
        vis = np.zeros(self.C, np.int32)
        _check(lib().af_engine_move_result(self._h, game, C.byref(a), C.byref(hp), _p(pol, C.c_float),
                                           _p(vis, C.c_int32), C.byref(tau)), "af_engine_move_result")
        return a.value, (pol if hp.value else None), vis, tau.value

    def set_roots(self, games, keys, last_cells, random_a=False, reset_tree=False, stream=None):
        """Batched set_root: games int[n], keys uint64[n][2KW], last_cells int[n] (-1 = None); random_a / reset_tree scalar or [n]."""
        n = len(games)
        g = np.ascontiguousarray(games, np.int32)
        k = np.ascontiguousarray(keys, np.uint64).reshape(n, self.KW2)
        lc = np.ascontiguousarray(last_cells, np.int32)
        ra = np.ascontiguousarray(np.broadcast_to(np.asarray(random_a, np.int32), (n,)))
        rt = np.ascontiguousarray(np.broadcast_to(np.asarray(reset_tree, np.int32), (n,)))
        _check(lib().af_engine_set_roots(self._h, stream, n, _p(g, C.c_int32), _p(k, C.c_uint64), _p(lc, C.c_int32),
                                         _p(ra, C.c_int32), _p(rt, C.c_int32)), "af_engine_set_roots")

    def move_results(self, games, stream=None):
        """Batched move_result -> (actions int32[n], has_policy int32[n], policies float32[n][C], visits int32[n][C], taus float64[n])."""
        n = len(games)
        g = np.ascontiguousarray(games, np.int32)
        act, hp = np.zeros(n, np.int32), np.zeros(n, np.int32)
        pol, vis, tau = np.zeros((n, self.C), np.float32), np.zeros((n, self.C), np.int32), np.zeros(n, np.float64)
        _check(lib().af_engine_move_results(self._h, stream, n, _p(g, C.c_int32), _p(act, C.c_int32), _p(hp, C.c_int32),
                                            _p(pol, C.c_float), _p(vis, C.c_int32), _p(tau, C.c_double)), "af_engine_move_results")
        return act, hp, pol, vis, tau

    def counters(self, stream=None):
        out = np.zeros(len(COUNTER_NAMES), np.uint64)
        _check(lib().af_engine_counters(self._h, stream, _p(out, C.c_uint64)), "af_engine_counters")
        return {k: int(v) for k, v in zip(COUNTER_NAMES, out)}

    def tick_histogram(self, stream=None, reset=False):
        """-> dict(selects=int64[64] games per selects-in-one-launch, wave_us=int64[32] games per 8-us bin of
        wave lifetime, max_wave_us=float): the launch-shape evidence of include/af_engine.h."""
        out = np.zeros(97, np.uint64)
        _check(lib().af_engine_tick_histogram(self._h, stream, _p(out, C.c_uint64), int(reset)), "tick_histogram")
        return dict(selects=out[:64].astype(np.int64), wave_us=out[64:96].astype(np.int64),
                    max_wave_us=float(out[96]) * 0.01)

    # ---- evaluation memo (include/af_engine.h, ABI v5): off unless memo_enable() was called ----
    def memo_enable(self, log2_buckets=18, max_stones=5):
        """Share evaluations between the games of this engine: positions with <= max_stones stones, 4 << log2_buckets entries
        (588 B each at 11x11, 1164 B at 15x15: the default table is 0.6 / 1.2 GB — memo_bytes says what was allocated; bench.py's
        22:5 is 9.9 GB).  Trees stay bit-identical; a launch's parameters change (params_key)."""
        _check(lib().af_engine_memo_enable(self._h, int(log2_buckets), int(max_stones)), "af_engine_memo_enable")
        kw = self.KW2 // 2
        self.memo_bytes = (4 << int(log2_buckets)) * (32 * kw + 256 * kw + 12)
        self._params["memo"] = (int(log2_buckets), int(max_stones))
        self._params["memo_epoch"] = int(lib().af_engine_memo_epoch(self._h))

    @property
    def memo(self):
        return self._params.get("memo")

    def memo_insert(self, policy_ptr, value_ptr, stream=None):
        """After the forward of a tick, same stream: remember the evaluations of the leaves the last tick() parked."""
        _check(lib().af_engine_memo_insert(self._h, stream, policy_ptr, value_ptr), "af_engine_memo_insert")

    def memo_clear(self, stream=None):
        """The evaluator's weights changed: forget everything (stream-ordered, O(1): the launches that follow carry a new epoch
        under which no stored entry matches — the epoch is part of params_key(), so a captured graph is dropped with it)."""
        _check(lib().af_engine_memo_clear(self._h, stream), "af_engine_memo_clear")
        self._params["memo_epoch"] = int(lib().af_engine_memo_epoch(self._h))

    def memo_stats(self, stream=None):
        out = np.zeros(6, np.uint64)
        _check(lib().af_engine_memo_stats(self._h, stream, _p(out, C.c_uint64)), "af_engine_memo_stats")
        return {k: int(v) for k, v in zip(("launches", "probes", "hits", "inserts", "replaced", "entries"), out)}

    def set_tick_budget(self, selects_per_launch):
        _check(lib().af_engine_set_tick_budget(self._h, int(selects_per_launch)), "af_engine_set_tick_budget")
        self._params["tick_budget"] = int(selects_per_launch)

    def progress(self, stream=None):
        out = np.zeros(2, np.uint64)
        _check(lib().af_engine_progress(self._h, stream, _p(out, C.c_uint64)), "af_engine_progress")
        return int(out[0]), int(out[1])

    def progress_async(self, out_ptr, stream=None):
        """The 16-byte progress copy without the wait (ABI v4): out_ptr = pinned host memory; stream-ordered, capturable."""
        _check(lib().af_engine_progress_async(self._h, stream, out_ptr), "af_engine_progress_async")

    def stamp(self, slot, stream=None):
        """Device-clock stamp (ABI v6): a one-wave kernel writes the 100-MHz device clock into slot `slot`; capturable."""
        _check(lib().af_engine_stamp(self._h, stream, int(slot)), "af_engine_stamp")

    def stamps_async(self, out_ptr, stream=None):
        """The STAMP_SLOTS stamps -> pinned host memory (uint64[64]), stream-ordered, capturable."""
        _check(lib().af_engine_stamps_async(self._h, stream, out_ptr), "af_engine_stamps_async")

    def tree_dump(self, game):
        cnt = _check(lib().af_engine_tree_dump(self._h, game, 0, None, None, None, None, None, None), "tree_dump")
        keys = np.zeros((cnt, self.KW2), np.uint64)
        sum_n = np.zeros(cnt, np.int32)
        n = np.zeros((cnt, self.C), np.int32)
        w = np.zeros((cnt, self.C), np.float32)
        p = np.zeros((cnt, self.C), np.float32)
        f = np.zeros((cnt, self.C), np.uint8)
        if cnt:
            _check(lib().af_engine_tree_dump(self._h, game, cnt, _p(keys, C.c_uint64), _p(sum_n, C.c_int32),
                                             _p(n, C.c_int32), _p(w, C.c_float), _p(p, C.c_float), _p(f, C.c_uint8)),
                   "tree_dump")
            if self.value_f64:                   # the exact rows: W is a python float on the pipe path
                w = np.zeros((cnt, self.C), np.float64)
                _check(lib().af_engine_tree_w64(self._h, game, cnt, _p(w, C.c_double)), "af_engine_tree_w64")
        return dict(keys=keys, sum_n=sum_n, n=n, w=w, p=p, f32=f)

    def load_tree(self, game, dump):
        """Adopt a tree (tree_dump's dict format) as game `game`'s store: Player.reset(search_tree)."""
        cnt = len(dump["sum_n"])
        arr = {k: np.ascontiguousarray(dump[k], t) for k, t in (("keys", np.uint64), ("sum_n", np.int32), ("n", np.int32),
                                                                 ("w", np.float32), ("p", np.float32), ("f32", np.uint8))}
        _check(lib().af_engine_load_tree(self._h, game, cnt, _p(arr["keys"], C.c_uint64), _p(arr["sum_n"], C.c_int32),
                                         _p(arr["n"], C.c_int32), _p(arr["w"], C.c_float), _p(arr["p"], C.c_float),
                                         _p(arr["f32"], C.c_uint8)), "af_engine_load_tree")
        if self.value_f64:
            w64 = np.ascontiguousarray(dump["w"], np.float64)
            _check(lib().af_engine_set_tree_w64(self._h, game, cnt, _p(w64, C.c_double)), "af_engine_set_tree_w64")

    def pack_ints(self, max_eps, max_plies):
        return int(_check(lib().af_engine_pack_ints(self._h, max_eps, max_plies), "af_engine_pack_ints"))

    def pack_episodes(self, out_ptr, max_eps, max_plies, stream=None):
        """Device-side hand-off (include/af_engine.h af_engine_pack_episodes): compacts the finished episodes into the
        int32 buffer at out_ptr (device memory or device-writable pinned host memory) on `stream`.  No sync, no copy."""
        _check(lib().af_engine_pack_episodes(self._h, stream, max_eps, max_plies, out_ptr), "af_engine_pack_episodes")


def unpack_episodes(buf, max_eps):
    """Packed hand-off buffer (int32 numpy, layout of af_engine_pack_episodes) -> list of raw episode dicts."""
    buf = np.ascontiguousarray(buf, np.int32)
    n, _, K, Cc = (int(x) for x in buf[:4])
    R = 2 * K + 2 * Cc + 2
    head = 4 + 5 * max_eps
    out = []
    for i in range(n):
        g, seq, T, p0 = (int(x) for x in buf[4 + 4 * i:8 + 4 * i])
        rec = buf[head + p0 * R:head + (p0 + T) * R].reshape(T, R)
        out.append(dict(game=g, seq=seq, T=T, final_value=float(buf[4 + 4 * max_eps + i:5 + 4 * max_eps + i].view(np.float32)[0]),
                        keys=np.ascontiguousarray(rec[:, :2 * K]).view(np.uint64).reshape(T, K).copy(),
                        policies=np.ascontiguousarray(rec[:, 2 * K:2 * K + Cc]).view(np.float32).copy(),
                        visits=rec[:, 2 * K + Cc:2 * K + 2 * Cc].copy(),
                        lasts=rec[:, 2 * K + 2 * Cc].copy(), actions=rec[:, 2 * K + 2 * Cc + 1].copy()))
    return out


def packed_used_ints(hdr, max_eps):
    """ints of a packed buffer that carry data, from its 4-int header."""
    n_plies, K, Cc = int(hdr[1]), int(hdr[2]), int(hdr[3])
    return 4 + 5 * max_eps + n_plies * (2 * K + 2 * Cc + 2)


def assemble_episode(raw, S, gamma):
    """Player.run's tail (player.py:73-82) + main.gen_data's result code (main.py:85-93):
    raw device records -> ([(state, policy[S,S], last_action, value, weight)], result)."""
    from . import utils
    T = raw["T"]
    value = float(raw["final_value"])
    if T % 2 == 1:
        value = -value
    weights = utils.construct_weights(T, gamma=gamma)
    rec = []
    for t in range(T):
        la = None if raw["lasts"][t] < 0 else (int(raw["lasts"][t]) // S, int(raw["lasts"][t]) % S)
        rec.append((key_to_state(raw["keys"][t], S), raw["policies"][t].reshape(S, S).copy(), la, value, weights[t]))
        value = -value
    last_value = rec[-1][-2]
    if last_value == 0.0:
        result = DRAW
    elif T % 2 == 1:
        result = BLACK_WIN
    else:
        result = WHITE_WIN
    return rec, result


class SelfPlayEngine:
    """G concurrent self-play games on one GPU: the device-resident replacement of
    main.py:50-55 (5 worker processes running Player.run) + NetworkAPI batching.

    pv_device: callable planes float32[G,3,S,S] (torch, on device) -> (policy[G,C], value[G])
    torch tensors on the same device — alphafive_amd.network.ResNet.eval_device, or a test stub.
    weights_version: callable (or constant) naming the weight set pv_device evaluates with; only needed when pv_device
    hides its net behind a wrapper (a lambda around net.eval_device): run_ticks_graph re-captures when it changes.
    value_f64: the arithmetic of the workers main.py actually runs (values through NetworkAPI pipes as python floats,
    networkAPI.py:72): W / Q in fp64.  Default False = the pv_fn path (self_play.py), SURVEY's parity target.
    """

    def __init__(self, cfg, num_games, pv_device, device=0, seed=0, first_game_id=0, training=True, node_cap=0,
                 value_f64=False, weights_version=None, eval_memo=None):
        import torch
        if not torch.cuda.is_available():
            raise EngineError("SelfPlayEngine needs a HIP device (torch.cuda.is_available() is False)")
        self.torch = torch
        self.cfg = cfg
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.engine = Engine(cfg, num_games, device=device, mode=MODE_SELFPLAY, training=training, seed=seed,
                             first_game_id=first_game_id, node_cap=node_cap, value_f64=value_f64)
        S = cfg.board_size
        self.G, self.S, self.C = num_games, S, S * S
        self.planes = torch.zeros((num_games, 3, S, S), dtype=torch.float32, device=self.dev)
        self.policy = torch.zeros((num_games, S * S), dtype=torch.float32, device=self.dev)
        self.value = torch.zeros((num_games,), dtype=torch.float32, device=self.dev)
        self.pv_device = pv_device
        # run_ticks_graph keys its captured graph on this (a replay skips the Python wrapper that reloads weights): for a
        # pv_device wrapped in a lambda pass weights_version=lambda: net.version
        self._weights_version = self._resolve_weights_version(weights_version)
        if hasattr(pv_device, "bind_outputs"):       # the HIP net writes into our tensors: no copy per tick
            pv_device.bind_outputs(self.policy, self.value)
        # eval_memo=dict(log2_buckets=.., max_stones=..) (or True for the defaults): evaluations are shared between the games
        # (Engine.memo_enable).  The stored bits belong to one set of weights, so the memo needs the same version source as
        # the graph loop and is cleared, stream-ordered, in front of the first tick that sees a new version.
        self._memo_version = None
        if eval_memo:
            if self._weights_version is None:
                raise EngineError("eval_memo: the evaluator exposes no weights_version (the memo could not tell when to forget); "
                                  "pass SelfPlayEngine(..., weights_version=lambda: net.version)")
            self.engine.memo_enable(**({} if eval_memo is True else dict(eval_memo)))
            self._memo_version = self._weights_version()
        self.ticks = 0
        self._boxes = {}
        self._graph = None
        self._prog_host, self._prog_events, self._prog_turn, self._prog_replays = None, None, 0, 0
        self.replay_events = []                      # (ticks, start event, end event) of the replays run with timed=True
        self._graph_stamped, self._stamp_host, self._stamp_event, self._stamp_pending, self.stamp_samples = None, None, None, False, []

    def tick(self, stamp_base=None):
        """One simulation step for every game: tree kernel -> leaf batch -> net.  stamp_base = s: device-clock stamps in slots s (before
        the tree kernel), s + 1 (between the two) and s + 2 (after the forward and the memo insert) — Engine.stamp."""
        torch = self.torch
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        if stamp_base is not None:
            self.engine.stamp(stamp_base, stream)
        memo = self.engine.memo is not None
        if memo:
            ver = self._weights_version()
            if ver != self._memo_version:            # new weights: every leaf parked from this tick on is evaluated by them
                self.engine.memo_clear(stream)
                self._memo_version = ver
        self.engine.tick(self.policy.data_ptr(), self.value.data_ptr(), self.planes.data_ptr(), stream)
        if stamp_base is not None:
            self.engine.stamp(stamp_base + 1, stream)
        p, v = self.pv_device(self.planes)
        if p.data_ptr() != self.policy.data_ptr():
            self.policy.copy_(p.reshape(self.G, self.C))
        if v.data_ptr() != self.value.data_ptr():
            self.value.copy_(v.reshape(self.G))
        if memo:
            self.engine.memo_insert(self.policy.data_ptr(), self.value.data_ptr(), stream)
        if stamp_base is not None:
            self.engine.stamp(stamp_base + 2, stream)
        self.ticks += 1

    def run_ticks(self, n, check_every=0):
        for i in range(n):
            self.tick()
            if check_every and (i + 1) % check_every == 0:
                self.check()

    # ---- the steady-state loop as a HIP graph: n x (tree kernel -> leaf batch -> net) + the progress words, one launch ----
    def _resolve_weights_version(self, explicit):
        """-> callable giving the evaluator's weight version (what a captured graph is keyed on), or None when the evaluator
        has no weights the engine could know about (a stub).  Sources, in order: the caller's weights_version=, the
        evaluator's own .weights_version (net_hip.make_eval's closure, tower_hip), the `version` of the object a bound method
        belongs to (pv_device=net.eval_device).  An evaluator that takes bind_outputs (i.e. one of ours, whose Python wrapper
        is the only place that reloads weights) without any version source would replay stale weights silently: refused — by
        run_ticks_graph() and by eval_memo=, not by the constructor (eager tick() users are not affected)."""
        if explicit is not None:
            return explicit if callable(explicit) else (lambda: explicit)
        pv = self.pv_device
        ver = getattr(pv, "weights_version", None)
        if callable(ver):
            return ver
        owner = getattr(pv, "__self__", None)
        if owner is not None and hasattr(owner, "version"):
            return lambda: owner.version
        if hasattr(pv, "bind_outputs"):
            # one of ours without a version source: eager tick() is fine (the Python wrapper reloads), anything that replays
            # launches or reuses stored bits (run_ticks_graph, eval_memo) is refused where it is asked for
            self._weights_version_missing = True
        return None

    def _graph_key(self, n):
        ver = self._weights_version
        return (int(n), self.engine.params_key(), ver() if ver is not None else None)

    def run_ticks_graph(self, n=16, timed=False, stamped=False):
        """n ticks replayed as ONE HIP graph on the current stream (the tick kernel, the forward's 13 launches with the value
        branch's fork / join, and at the end a 16-byte copy of the engine's progress words into pinned host memory).  Returns at
        once; progress_lagged() reads the words one replay later, so polling never drains the device.  The graph is keyed on
        everything a launch has baked in: n, the engine's by-value parameters (training flag, simulation budget, per-launch select
        budget) and the evaluator's weight version — a change drops it, one eager tick re-packs the weights and the batch is
        captured again (same protocol as Player._search_batch).  A failing capture raises: there is no silent eager fallback.
        timed=True brackets the replay with HIP timing events on its stream and appends them to self.replay_events.
        stamped=True replays a second capture of the same n ticks with three device-clock stamps per tick (Engine.stamp: before the tree
        kernel, between the two, after the forward) and a copy of the stamps to pinned memory at its end: read_stamps() returns, one
        replay late, the (tree kernel, forward) durations as they ran INSIDE the graph — timing events cannot be captured on ROCm."""
        torch = self.torch
        if getattr(self, "_weights_version_missing", False):
            raise EngineError("run_ticks_graph: pv_device takes bind_outputs but exposes no weights_version (a replayed graph "
                              "would keep evaluating with stale weights); pass SelfPlayEngine(..., weights_version=lambda: net.version)")
        key = self._graph_key(n)
        if stamped:
            if 3 * n > STAMP_SLOTS:
                raise EngineError("run_ticks_graph(stamped=True): at most %d ticks per replay" % (STAMP_SLOTS // 3))
            if self._graph_stamped is None or self._graph_stamped[0] != key:
                self._graph_stamped = None
                if self._graph is None or self._graph[0] != key:
                    self.run_ticks_graph(n)               # (the plain capture does the warm-up tick and the allocations)
                if self._stamp_host is None:
                    self._stamp_host = torch.zeros(STAMP_SLOTS, dtype=torch.int64, pin_memory=True)
                    self._stamp_event = torch.cuda.Event()
                    self.engine.stamp(0, torch.cuda.current_stream(self.dev).cuda_stream)     # (allocates the stamp array: not inside a capture)
                torch.cuda.synchronize(self.dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for i in range(n):
                        self.tick(stamp_base=3 * i)
                    st = torch.cuda.current_stream(self.dev).cuda_stream
                    self.engine.stamps_async(self._stamp_host.data_ptr(), st)
                    self.engine.progress_async(self._prog_host.data_ptr(), st)
                self.ticks -= n
                self._graph_stamped = (self._graph_key(n), g, n)
            if self._stamp_pending:                       # the previous stamped replay's stamps, before this one overwrites them
                self._collect_stamps()
            self._graph_stamped[1].replay()
            self._stamp_event.record(torch.cuda.current_stream(self.dev))
            self._stamp_pending = True
            self.ticks += n
            self._prog_turn ^= 1
            self._prog_events[self._prog_turn].record(torch.cuda.current_stream(self.dev))
            self._prog_replays += 1
            return
        if self._graph is None or self._graph[0] != key:
            self._graph = None
            if self._prog_host is None:
                self._prog_host = torch.zeros(2, dtype=torch.int64, pin_memory=True)
                self._prog_events = [torch.cuda.Event(), torch.cuda.Event()]
            self.tick()                              # outside the capture: weight reload, lazy allocations, side-stream creation
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls can invalidate the capture — a process that also runs RCCL has a watchdog
            # thread querying events while we capture (multi-GPU bench: the process group is up before the first graph)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(n):
                    self.tick()
                st = torch.cuda.current_stream(self.dev).cuda_stream
                self.engine.progress_async(self._prog_host.data_ptr(), st)
            self.ticks -= n                          # (capture recorded the launches, it did not run them)
            self._graph = (self._graph_key(n), g)
        if timed:                                    # HIP events around the replay, on the stream it runs on (bench.py reads them)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.dev))
        self._graph[1].replay()
        if timed:
            e1.record(torch.cuda.current_stream(self.dev))
            self.replay_events.append((n, e0, e1))
        self.ticks += n
        self._prog_turn ^= 1
        self._prog_events[self._prog_turn].record(torch.cuda.current_stream(self.dev))
        self._prog_replays += 1

    def _collect_stamps(self):
        self._stamp_event.synchronize()
        t = self._stamp_host.numpy().astype(np.int64)[:3 * self._graph_stamped[2]].reshape(-1, 3)
        self.stamp_samples.append(np.stack([(t[:, 1] - t[:, 0]) * 1e-5, (t[:, 2] - t[:, 1]) * 1e-5], axis=1))     # ms (10-ns ticks)
        self._stamp_pending = False

    def read_stamps(self):
        """-> float64[k][2]: (tree kernel ms, forward ms) of every tick of every stamped replay so far, as they ran inside the graph
        (each interval includes the launch boundaries of the two stamp kernels around it).  Waits for the last stamped replay only."""
        if self._stamp_pending:
            self._collect_stamps()
        return np.concatenate(self.stamp_samples) if self.stamp_samples else np.zeros((0, 2))

    def progress_lagged(self):
        """(plies committed, episodes finished) as of the end of the PREVIOUS run_ticks_graph() replay or later: waits for that
        replay's event only — the newest replay keeps the device busy meanwhile."""
        if self._prog_replays < 2:
            return 0, 0
        self._prog_events[self._prog_turn ^ 1].synchronize()
        return int(self._prog_host[0]), int(self._prog_host[1])

    def check(self):
        stream = self.torch.cuda.current_stream(self.dev).cuda_stream
        return self.engine.status(stream)

    def counters(self):
        return self.engine.counters(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def progress(self):
        return self.engine.progress(self.torch.cuda.current_stream(self.dev).cuda_stream)

    # ---- finished-episode hand-off: pack on the device, written straight into pinned host memory ----
    def _pack_plies(self, cap):
        """Ply capacity of a pack buffer for `cap` episodes: ~40 per episode (they average 26 at 11x11; what does not fit
        waits for the next post) but never less than ONE maximum-length episode (S*S plies) — af_pack_scan takes a prefix in
        (game, sequence) order, so an episode longer than the whole buffer would block every later one for ever."""
        return max(cap * 40, int(self.engine.max_plies))

    def _outbox(self, cap):
        box = self._boxes.get(cap)
        if box is None:
            torch = self.torch
            max_plies = self._pack_plies(cap)
            ints = self.engine.pack_ints(cap, max_plies)
            box = dict(cap=cap, max_plies=max_plies, buf=torch.zeros(ints, dtype=torch.int32, pin_memory=True),
                       event=torch.cuda.Event(), posted=False)
            self._boxes[cap] = box
        return box

    def post_episodes(self, cap=256):
        """Asynchronous half of the hand-off: two small kernels on the tick stream compact the finished episodes into a
        pinned host buffer (the device writes it directly: no copy, no synchronisation).  collect_episodes() reads it."""
        box = self._outbox(cap)
        if box["posted"]:
            raise EngineError("post_episodes: the previous post has not been collected")
        stream = self.torch.cuda.current_stream(self.dev)
        self.engine.pack_episodes(box["buf"].data_ptr(), cap, box["max_plies"], stream.cuda_stream)
        box["event"].record(stream)
        box["posted"] = True
        return box

    def collect_episodes(self, cap=256, unpack=True):
        """-> raw episode dicts of the last post_episodes(cap) (waits for its event only).  unpack=False: the packed int32
        buffer itself (numpy view of the pinned memory, valid until the next post_episodes(cap); buf[0] = episodes,
        buf[1] = plies: a caller that only counts reads the header and does no per-episode work), or None."""
        box = self._outbox(cap)
        if not box["posted"]:
            return [] if unpack else None
        box["event"].synchronize()
        box["posted"] = False
        return unpack_episodes(box["buf"].numpy(), cap) if unpack else box["buf"].numpy()

    def pop_raw(self, cap=256):
        self.post_episodes(cap)
        return self.collect_episodes(cap)

    def post_episodes_device(self, cap=256):
        """Multi-GPU form of post_episodes: the packed buffer stays in device memory (a torch int32 tensor) so that it
        can be handed to RCCL as it is (alphafive_amd.dist.gather_packed)."""
        key = ("dev", cap)
        buf = self._boxes.get(key)
        if buf is None:
            buf = self.torch.zeros(self.engine.pack_ints(cap, self._pack_plies(cap)), dtype=self.torch.int32, device=self.dev)
            self._boxes[key] = buf
        self.engine.pack_episodes(buf.data_ptr(), cap, self._pack_plies(cap), self.torch.cuda.current_stream(self.dev).cuda_stream)
        return buf

    def pop_episodes(self, cap=256):
        """-> list of (game_record, result) exactly as main.gen_data puts on its queue (main.py:94)."""
        return [assemble_episode(r, self.S, self.cfg.gamma) for r in self.pop_raw(cap)]

    def close(self):
        self.engine.close()
