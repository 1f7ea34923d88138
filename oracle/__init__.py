"""ctypes binding of the CPU oracle (oracle/af_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (alphafive_amd/) never imports
this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libaf_oracle.so")

RNG_MT, RNG_PHILOX = 0, 1
PV_PSEUDO, PV_CALLBACK = 0, 1
STATE_CAP = 272


class Config(C.Structure):
    _fields_ = [("board_size", C.c_int), ("goal", C.c_int), ("sims", C.c_int), ("upper_sims", C.c_int),
                ("init_temp", C.c_double), ("gamma", C.c_double), ("tau_decay", C.c_double),
                ("tau_decay_r", C.c_double), ("alpha", C.c_double), ("c_puct", C.c_double)]

    @classmethod
    def from_cfg(cls, cfg):
        """cfg: any attribute bag with the reference's config.py names."""
        return cls(cfg.board_size, cfg.goal, cfg.simulation_per_step, cfg.upper_simulation_per_step,
                   cfg.init_temp, cfg.gamma, cfg.tau_decay_rate, cfg.tau_decay_rate_r,
                   cfg.dirichlet_alpha, cfg.c_puct)


PV_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("af_oracle.c", "af_oracle.h")] + \
          [os.path.join(_HERE, "..", "include", "af_noise.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src if os.path.exists(s))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "libaf_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp, i32p, f32p, u8p, f64p, u64p, u32p = (C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                                 C.POINTER(C.c_uint8), C.POINTER(C.c_double),
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint32))
        L.afo_create.restype = vp
        L.afo_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, PV_FN,
                                 vp, C.c_uint32, C.c_uint32]
        L.afo_destroy.argtypes = [vp]
        L.afo_reset.argtypes = [vp]
        L.afo_set_training.argtypes = [vp, C.c_int]
        L.afo_set_simulations.argtypes = [vp, C.c_int, C.c_int]
        L.afo_set_value_f64.argtypes = [vp, C.c_int]
        L.afo_node_get_w64.argtypes = [vp, C.c_char_p, f64p]
        L.afo_tree_dump_w64.argtypes = [vp, C.c_int, f64p]
        L.afo_get_action.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, f32p, C.POINTER(C.c_int), i32p]
        L.afo_run.argtypes = [vp, C.c_int, C.c_char_p, f32p, C.POINTER(C.c_int), C.POINTER(C.c_int), i32p,
                              f64p, f32p, f64p]
        L.afo_tree_size.argtypes = [vp]
        L.afo_node_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int), i32p, f32p, f32p, u8p]
        L.afo_tree_dump.argtypes = [vp, C.c_int, u64p, i32p, i32p, f32p, f32p, u8p]
        L.afo_stats.argtypes = [vp, u64p]
        L.afo_tie_stats.argtypes = [vp, u64p]
        L.afo_tau.restype = C.c_double
        L.afo_tau.argtypes = [vp]
        L.afo_board_to_state.argtypes = [C.POINTER(C.c_int8), C.c_int, C.c_char_p, C.c_int]
        L.afo_state_to_board.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int8)]
        L.afo_is_game_over.argtypes = [C.POINTER(C.c_int8), C.c_int, C.c_int, f64p]
        L.afo_legal_actions.argtypes = [C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_int)]
        L.afo_board_to_inputs.argtypes = [C.POINTER(C.c_int8), C.c_int, C.c_int, f32p]
        L.afo_step.argtypes = [C.POINTER(C.c_int8), C.c_int, C.c_int]
        L.afo_construct_weights.argtypes = [C.c_int, C.c_double, f32p]
        L.afo_pairwise_sum_f32.restype = C.c_float
        L.afo_pairwise_sum_f32.argtypes = [f32p, C.c_int]
        L.afo_pseudonet.argtypes = [f32p, C.c_int, C.c_uint32, C.c_uint32, f32p, f32p]
        L.afo_np_u32.restype = C.c_uint32
        L.afo_np_u32.argtypes = [vp]
        L.afo_py_u32.restype = C.c_uint32
        L.afo_py_u32.argtypes = [vp]
        L.afo_np_dirichlet.argtypes = [vp, C.c_double, C.c_int, f64p]
        L.afo_noise_philox_dirichlet.argtypes = [C.c_double, u64p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, f64p]
        L.afo_log.restype = C.c_double
        L.afo_log.argtypes = [C.c_double]
        L.afo_exp.restype = C.c_double
        L.afo_exp.argtypes = [C.c_double]
        L.afo_powf.restype = C.c_float
        L.afo_powf.argtypes = [C.c_float, C.c_float]
        L.afo_logf.restype = C.c_float
        L.afo_logf.argtypes = [C.c_float]
        L.afo_expf.restype = C.c_float
        L.afo_expf.argtypes = [C.c_float]
        L.afo_philox.argtypes = [u32p, u32p, u32p]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


# ---------------------------------------------------------------- rules / codecs
def board_to_state(board):
    b = np.ascontiguousarray(board, np.int8)
    out = C.create_string_buffer(STATE_CAP)
    n = lib().afo_board_to_state(_p(b, C.c_int8), b.shape[0], out, STATE_CAP)
    assert n >= 0
    return out.value.decode()


def state_to_board(state, S):
    b = np.zeros((S, S), np.int8)
    rc = lib().afo_state_to_board(state.encode(), S, _p(b, C.c_int8))
    assert rc == 0
    return b


def is_game_over(board, goal):
    b = np.ascontiguousarray(board, np.int8)
    v = C.c_double(0)
    over = lib().afo_is_game_over(_p(b, C.c_int8), b.shape[0], goal, C.byref(v))
    return bool(over), v.value


def legal_actions(board):
    b = np.ascontiguousarray(board, np.int8)
    S = b.shape[0]
    cells = (C.c_int * (S * S))()
    L = lib().afo_legal_actions(_p(b, C.c_int8), S, cells)
    return [(cells[i] // S, cells[i] % S) for i in range(L)]


def board_to_inputs(board, last_action=None):
    b = np.ascontiguousarray(board, np.int8)
    S = b.shape[0]
    out = np.zeros((3, S, S), np.float32)
    lc = -1 if last_action is None else last_action[0] * S + last_action[1]
    lib().afo_board_to_inputs(_p(b, C.c_int8), S, lc, _p(out, C.c_float))
    return out


def step(board, action):
    b = np.ascontiguousarray(board, np.int8).copy()
    S = b.shape[0]
    lib().afo_step(_p(b, C.c_int8), S, action[0] * S + action[1])
    return b


def construct_weights(T, gamma):
    w = np.zeros(T, np.float32)
    lib().afo_construct_weights(T, gamma, _p(w, C.c_float))
    return w


def pairwise_sum_f32(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.float32(lib().afo_pairwise_sum_f32(_p(a, C.c_float), a.size))


def pseudonet(planes, salt=0, peak=0):
    """planes float32[B,3,S,S] -> (policy float32[B,C], value float32[B])"""
    x = np.ascontiguousarray(planes, np.float32)
    B, _, S, _ = x.shape
    Cc = S * S
    pol = np.zeros((B, Cc), np.float32)
    val = np.zeros(B, np.float32)
    for b in range(B):
        v = C.c_float(0)
        lib().afo_pseudonet(_p(x[b], C.c_float), Cc, salt, peak, _p(pol[b], C.c_float), C.byref(v))
        val[b] = v.value
    return pol, val


# ---------------------------------------------------------------- player
class OraclePlayer:
    """genData/player.py:Player restated; pv = None -> built-in pseudo-net, or a Python
    callable float32[1,3,S,S] -> (float32[1,C], float32[1]) like the reference's pv_fn."""

    def __init__(self, cfg, training=True, rng_mode=RNG_PHILOX, seed=0, game_id=0, pv_fn=None,
                 pseudo_salt=0, pseudo_peak=0, value_f64=False):
        self.value_f64 = bool(value_f64)          # the reference's pipe path (networkAPI.py:72): w, q fp64
        self.cfg = Config.from_cfg(cfg) if not isinstance(cfg, Config) else cfg
        self.S = self.cfg.board_size
        self.C = self.S * self.S
        self._cb = PV_FN(self._thunk) if pv_fn is not None else PV_FN()
        self._pv = pv_fn
        self.h = lib().afo_create(C.byref(self.cfg), int(training), rng_mode, seed, game_id,
                                  PV_CALLBACK if pv_fn is not None else PV_PSEUDO, self._cb, None,
                                  pseudo_salt, pseudo_peak)
        assert self.h
        if self.value_f64:
            lib().afo_set_value_f64(self.h, 1)

    def _thunk(self, planes, policy, value, user):
        x = np.ctypeslib.as_array(planes, shape=(1, 3, self.S, self.S))
        p, v = self._pv(x.copy())
        np.ctypeslib.as_array(policy, shape=(self.C,))[:] = np.asarray(p, np.float32).reshape(-1)
        value[0] = np.float32(np.asarray(v).reshape(-1)[0])

    def close(self):
        if self.h:
            lib().afo_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        lib().afo_reset(self.h)

    def set_training(self, t):
        lib().afo_set_training(self.h, int(t))

    def set_simulations(self, sims, upper):
        lib().afo_set_simulations(self.h, int(sims), int(upper))

    @property
    def tau(self):
        return lib().afo_tau(self.h)

    def get_action(self, state, last_action=None, random_a=False):
        pol = np.zeros(self.C, np.float32)
        vis = np.zeros(self.C, np.int32)
        a = C.c_int(-1)
        lc = -1 if last_action is None else last_action[0] * self.S + last_action[1]
        rc = lib().afo_get_action(self.h, state.encode(), lc, int(random_a), _p(pol, C.c_float), C.byref(a),
                                  _p(vis, C.c_int32))
        if rc < 0:
            raise RuntimeError(f"afo_get_action failed rc={rc}")
        action = (a.value // self.S, a.value % self.S)
        return (None if rc == 1 else pol.reshape(self.S, self.S)), action, vis

    def run(self, max_T=None):
        T_cap = max_T or self.C + 1
        states = C.create_string_buffer(T_cap * STATE_CAP)
        pol = np.zeros((T_cap, self.C), np.float32)
        last = np.zeros(T_cap, np.int32)
        act = np.zeros(T_cap, np.int32)
        vis = np.zeros((T_cap, self.C), np.int32)
        val = np.zeros(T_cap, np.float64)
        wts = np.zeros(T_cap, np.float32)
        fv = C.c_double(0)
        T = lib().afo_run(self.h, T_cap, states, _p(pol, C.c_float), _p(last, C.c_int), _p(act, C.c_int),
                          _p(vis, C.c_int32), _p(val, C.c_double), _p(wts, C.c_float), C.byref(fv))
        if T < 0:
            raise RuntimeError(f"afo_run failed rc={T}")
        raw = states.raw
        recs = []
        for i in range(T):
            s = raw[i * STATE_CAP:(i + 1) * STATE_CAP].split(b"\0", 1)[0].decode()
            la = None if last[i] < 0 else (int(last[i]) // self.S, int(last[i]) % self.S)
            recs.append((s, pol[i].reshape(self.S, self.S).copy(), la, float(val[i]), np.float32(wts[i])))
        return recs, dict(actions=act[:T].copy(), visits=vis[:T].copy(), final_value=fv.value)

    def tree_size(self):
        return lib().afo_tree_size(self.h)

    def node(self, state):
        sum_n = C.c_int(0)
        n = np.zeros(self.C, np.int32)
        w = np.zeros(self.C, np.float32)
        p = np.zeros(self.C, np.float32)
        f = np.zeros(self.C, np.uint8)
        rc = lib().afo_node_get(self.h, state.encode(), C.byref(sum_n), _p(n, C.c_int32), _p(w, C.c_float),
                                _p(p, C.c_float), _p(f, C.c_uint8))
        if rc != 1:
            return None
        out = dict(sum_n=sum_n.value, n=n, w=w, p=p, f32=f)
        if self.value_f64:
            out["w64"] = np.zeros(self.C, np.float64)
            lib().afo_node_get_w64(self.h, state.encode(), _p(out["w64"], C.c_double))
        return out

    def tree_dump(self):
        cnt = self.tree_size()
        keys = np.zeros((cnt, 8), np.uint64)
        sum_n = np.zeros(cnt, np.int32)
        n = np.zeros((cnt, self.C), np.int32)
        w = np.zeros((cnt, self.C), np.float32)
        p = np.zeros((cnt, self.C), np.float32)
        f = np.zeros((cnt, self.C), np.uint8)
        lib().afo_tree_dump(self.h, cnt, _p(keys, C.c_uint64), _p(sum_n, C.c_int32), _p(n, C.c_int32),
                            _p(w, C.c_float), _p(p, C.c_float), _p(f, C.c_uint8))
        out = dict(keys=keys, sum_n=sum_n, n=n, w=w, p=p, f32=f)
        if self.value_f64:
            out["w64"] = np.zeros((cnt, self.C), np.float64)
            lib().afo_tree_dump_w64(self.h, cnt, _p(out["w64"], C.c_double))
        return out

    def stats(self):
        out = np.zeros(5, np.uint64)
        lib().afo_stats(self.h, _p(out, C.c_uint64))
        return dict(sims=int(out[0]), selects=int(out[1]), expands=int(out[2]), terminals=int(out[3]),
                    plies=int(out[4]))

    def tie_stats(self):
        """How often a uniform pick had more than one candidate so far (the only places where the RNG decides a search)."""
        out = np.zeros(3, np.uint64)
        lib().afo_tie_stats(self.h, _p(out, C.c_uint64))
        return dict(select=int(out[0]), forced=int(out[1]), best=int(out[2]))

    def np_u32(self):
        return lib().afo_np_u32(self.h)

    def py_u32(self):
        return lib().afo_py_u32(self.h)

    def np_dirichlet(self, alpha, L):
        out = np.zeros(L, np.float64)
        lib().afo_np_dirichlet(self.h, alpha, L, _p(out, C.c_double))
        return out
