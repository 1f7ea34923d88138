"""DeviceRandomStack — utils.RandomStack (reference utils.py:14-146) with the positions resident in HBM
(libaf_replay.so, include/af_replay.h).  Same call surface and the same random draws in the same order as the
reference, so a seeded run produces the same batches as the host class bit for bit (tests/test_gpu_replay.py);
get_data() returns device tensors the trainer consumes without a host round trip.

push() keeps the reference's scalar bookkeeping on the host (one episode at a time, a few Python `random` draws)
and uploads the accepted episode once; push_packed() / iter_push_packed() take the engine's packed DEVICE hand-off buffer
instead: the host reads its header only and the plies are decoded on the device (no 5-tuples, no per-ply Python, no
upload); get_data() draws (which positions, quarter turns, flip) on the host and runs the gather + 8-fold symmetry +
board_to_inputs encoding as one kernel launch.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import utils

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("AF_REPLAY_LIB") or os.path.join(_PKG, "_lib", "libaf_replay.so")
_lib = None


class ReplayError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise ImportError(f"{_LIBPATH} not built (python -m alphafive_amd.build)")
        # torch first: its wheel bundles its own HIP runtime; a process in which /opt/rocm's copy was pulled in earlier (by this
        # library's DT_NEEDED) ends up with two runtimes, and the second one finds no device
        import torch  # noqa: F401
        L = C.CDLL(_LIBPATH)
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
        L.af_replay_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.af_replay_destroy.argtypes = [vp]
        L.af_replay_destroy.restype = None
        L.af_replay_append.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int8), fp, ip, fp, fp]
        L.af_replay_drop_front.argtypes = [vp, C.c_int32]
        L.af_replay_size.argtypes = [vp]
        L.af_replay_sample.argtypes = [vp, vp, C.c_int32, ip, ip, ip, vp, vp, vp, vp]
        L.af_replay_append_packed.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32]
        L.af_replay_set_weights.argtypes = [vp, fp, C.c_int32]
        L.af_replay_check.argtypes = [vp, vp]
        L.af_replay_strerror.argtypes = [C.c_int]
        L.af_replay_strerror.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise ReplayError(f"{what}: {lib().af_replay_strerror(rc).decode()} (code {rc})")
    return rc


class _PackedEpisode(object):
    """Stands for the list of 5-tuples RandomStack.push() receives when the episode lives in a packed DEVICE buffer
    (af_engine_pack_episodes): push() only needs its length; _store() hands (buffer, index) to af_replay_append_packed."""
    __slots__ = ("buf", "max_eps", "index", "T")

    def __init__(self, buf, max_eps, index, T):
        self.buf, self.max_eps, self.index, self.T = buf, max_eps, index, T

    def __len__(self):
        return self.T


class DeviceRandomStack(utils.RandomStack):
    def __init__(self, board_size, length=2000, device=0, max_episode=None):
        super().__init__(board_size, length)
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        # a push may append one episode twice before the eviction brings the size back to `length`
        max_episode = max_episode or board_size * board_size
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _check(lib().af_replay_create(board_size, length + 2 * max_episode, idx, C.byref(self._h)), "af_replay_create")
        self.data = None                       # positions live on the device
        self._wtab_gamma = None
        self._max_T = max_episode

    # ---- storage hooks ----
    def _size(self):
        return int(lib().af_replay_size(self._h))

    def _store(self, data):
        if isinstance(data, _PackedEpisode):    # device-to-device: one launch decodes the episode out of the packed buffer
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _check(lib().af_replay_append_packed(self._h, stream, data.buf.data_ptr(), data.max_eps, data.index, data.T),
                   "af_replay_append_packed")
            return
        S, n = self.board_size, len(data)
        boards = np.empty((n, S * S), np.int8)
        pol = np.empty((n, S * S), np.float32)
        last = np.empty(n, np.int32)
        val = np.empty(n, np.float32)
        wts = np.empty(n, np.float32)
        for i, (state, p, la, v, w) in enumerate(data):
            boards[i] = utils.state_to_board(state, S).reshape(-1)
            pol[i] = np.asarray(p, np.float32).reshape(-1)
            last[i] = -1 if la is None else la[0] * S + la[1]
            val[i], wts[i] = v, w
        stream = torch.cuda.current_stream(self.device).cuda_stream
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        _check(lib().af_replay_append(self._h, stream, n, boards.ctypes.data_as(C.POINTER(C.c_int8)), pol.ctypes.data_as(fp),
                                      last.ctypes.data_as(ip), val.ctypes.data_as(fp), wts.ctypes.data_as(fp)),
               "af_replay_append")

    def _drop_front(self, n):
        _check(lib().af_replay_drop_front(self._h, n), "af_replay_drop_front")

    # ---- device-to-device hand-off (main.py:60-61 `stack.push(*q.get())` without the episodes leaving the GPU) ----
    def _ensure_weights(self, gamma):
        if self._wtab_gamma == gamma:
            return
        T = self._max_T
        tab = np.zeros((T + 1, T), np.float32)
        for t in range(1, T + 1):
            tab[t, :t] = utils.construct_weights(t, gamma=gamma)        # utils.py:286-296, numpy float32 arithmetic = the spec
        _check(lib().af_replay_set_weights(self._h, tab.ctypes.data_as(C.POINTER(C.c_float)), T), "af_replay_set_weights")
        self._wtab_gamma = gamma

    def push_packed(self, buf, max_eps, gamma):
        return list(self.iter_push_packed(buf, max_eps, gamma))

    def iter_push_packed(self, buf, max_eps, gamma):
        """Push every episode of a packed DEVICE hand-off buffer (SelfPlayEngine.post_episodes_device(max_eps)) in order.
        The host reads only the buffer's header (episode lengths and final values: 4 + 5*max_eps ints) and runs
        RandomStack.push's accept / duplicate / evict draws on (T, result) — same draws, same order as pushing the 5-tuples;
        the plies themselves are decoded on the device (af_replay_append_packed).  Yields push()'s result episode by episode
        (a generator, so that a training loop can draw batches between two pushes exactly where main.py:60-68 does).
        The buffer must stay untouched until the appends queued on the current stream have run (the engine's next pack
        comes later on the same stream, so the natural order post -> push_packed -> ticks -> post is safe)."""
        assert buf.is_cuda and buf.dtype == torch.int32
        self._ensure_weights(float(gamma))
        head = buf[:4 + 5 * max_eps].cpu().numpy()                       # waits for the pack kernels only
        n = int(head[0])
        finals = head[4 + 4 * max_eps:4 + 5 * max_eps].view(np.float32)
        for e in range(n):
            T = int(head[4 + 4 * e + 2])
            fv = float(finals[e])
            result = utils.DRAW if fv == 0.0 else (utils.BLACK_WIN if T % 2 == 1 else utils.WHITE_WIN)   # main.py:85-93
            yield self.push(_PackedEpisode(buf, max_eps, e, T), result)

    def check(self):
        """Raises if a packed append found its buffer not to hold what the header said (synchronises the stream)."""
        _check(lib().af_replay_check(self._h, torch.cuda.current_stream(self.device).cuda_stream), "af_replay_check")

    # ---- persistence is a host concern: use utils.RandomStack for the reference's pickles ----
    def save(self, s=""):
        raise NotImplementedError("DeviceRandomStack keeps positions in HBM; use utils.RandomStack for data_buffer/*.pkl")

    load = save

    def get_data(self, batch_size=1):
        """utils.py:118-146.  Draws: np.random.choice(len, num, replace=False), then per sample
        np.random.choice([0,1,2,3]) (np stream) and random.choice([1,2]) (Python stream) — exactly the reference's."""
        import random as _random
        S = self.board_size
        size = self._size()
        num = min(batch_size, size)
        idx = np.random.choice(size, size=num, replace=False).astype(np.int32)
        turns = np.empty(num, np.int32)
        flip = np.empty(num, np.int32)
        for i in range(num):
            turns[i] = np.random.choice([0, 1, 2, 3])
            flip[i] = 1 if _random.choice([1, 2]) == 1 else 0
        boards = torch.empty((num, 3, S, S), dtype=torch.float32, device=self.device)
        weights = torch.empty((num,), dtype=torch.float32, device=self.device)
        values = torch.empty((num,), dtype=torch.float32, device=self.device)
        policies = torch.empty((num, S * S), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ip = C.POINTER(C.c_int32)
        _check(lib().af_replay_sample(self._h, stream, num, idx.ctypes.data_as(ip), turns.ctypes.data_as(ip),
                                      flip.ctypes.data_as(ip), boards.data_ptr(), weights.data_ptr(), values.data_ptr(),
                                      policies.data_ptr()), "af_replay_sample")
        return boards, weights, values, policies

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().af_replay_destroy(self._h)
            except Exception:       # interpreter shutdown
                pass
            self._h = None

    __del__ = close
