"""DeviceRandomStack — utils.RandomStack (reference utils.py:14-146) with the positions resident in HBM
(libaf_replay.so, include/af_replay.h).  Same call surface and the same random draws in the same order as the
reference, so a seeded run produces the same batches as the host class bit for bit (tests/test_gpu_replay.py);
get_data() returns device tensors the trainer consumes without a host round trip.

push() keeps the reference's scalar bookkeeping on the host (one episode at a time, a few Python `random` draws)
and uploads the accepted episode once; get_data() draws (which positions, quarter turns, flip) on the host and
runs the gather + 8-fold symmetry + board_to_inputs encoding as one kernel launch.
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
        L = C.CDLL(_LIBPATH)
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
        L.af_replay_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.af_replay_destroy.argtypes = [vp]
        L.af_replay_destroy.restype = None
        L.af_replay_append.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int8), fp, ip, fp, fp]
        L.af_replay_drop_front.argtypes = [vp, C.c_int32]
        L.af_replay_size.argtypes = [vp]
        L.af_replay_sample.argtypes = [vp, vp, C.c_int32, ip, ip, ip, vp, vp, vp, vp]
        L.af_replay_strerror.argtypes = [C.c_int]
        L.af_replay_strerror.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise ReplayError(f"{what}: {lib().af_replay_strerror(rc).decode()} (code {rc})")
    return rc


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

    # ---- storage hooks ----
    def _size(self):
        return int(lib().af_replay_size(self._h))

    def _store(self, data):
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
