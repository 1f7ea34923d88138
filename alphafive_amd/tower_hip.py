"""ctypes binding of libaf_tower.so (include/af_tower_bf16.h): the hand-written bf16 MFMA residual tower of
BASELINE configs[4].  Raises if the library is missing — DeepResNet.select_backend decides what to do then."""
import ctypes as C
import os

import numpy as np
import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("AF_TOWER_LIB") or os.path.join(_PKG, "_lib", "libaf_tower.so")
_lib = None


class TowerError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise ImportError(f"{_LIBPATH} not built (python -m alphafive_amd.build)")
        import torch  # noqa: F401  (torch first: see engine.py:lib)
        L = C.CDLL(_LIBPATH)
        vp, fp = C.c_void_p, C.POINTER(C.c_float)
        L.af_tower_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.af_tower_destroy.argtypes = [vp]
        L.af_tower_destroy.restype = None
        L.af_tower_set_block.argtypes = [vp, C.c_int32, fp, fp, fp, fp, fp, fp]
        L.af_tower_set_stem.argtypes = [vp, fp, fp]
        L.af_tower_set_heads.argtypes = [vp, fp, fp, fp, fp]
        L.af_tower_stem.argtypes = [vp, vp, vp, vp, C.c_int32]
        L.af_tower_heads.argtypes = [vp, vp, vp, vp, vp, C.c_int32]
        L.af_tower_set_dense.argtypes = [vp, fp, fp, fp, fp, fp, fp]
        L.af_tower_dense.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32]
        L.af_tower_pix.argtypes = [vp]
        L.af_tower_plane_elems.argtypes = [vp]
        L.af_tower_plane_elems.restype = C.c_int64
        L.af_tower_forward.argtypes = [vp, vp, vp, vp, C.c_int32]
        L.af_tower_flops_per_position.argtypes = [vp]
        L.af_tower_flops_per_position.restype = C.c_int64
        L.af_tower_tune.argtypes = [C.c_int32, C.c_int32]
        L.af_tower_strerror.argtypes = [C.c_int]
        L.af_tower_strerror.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise TowerError(f"{what}: {lib().af_tower_strerror(rc).decode()} (code {rc})")


def tune(key, value):
    _check(lib().af_tower_tune(key, value), "af_tower_tune")


class HipTower(object):
    """blocks = list of dicts with torch tensors res/c1/c2 = (weight OIHW, bias), as DeepResNet.tower holds them."""

    def __init__(self, blocks, board_size, width, max_batch, device, stem=None, vconv=None, pconv=None, dense=None):
        self.S, self.width, self.max_batch, self.device = board_size, width, max_batch, torch.device(device)
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _check(lib().af_tower_create(board_size, width, len(blocks), idx, C.byref(self._h)), "af_tower_create")
        fp = C.POINTER(C.c_float)

        def host(t):
            a = np.ascontiguousarray(t.detach().float().cpu().numpy(), np.float32)
            return a, a.ctypes.data_as(fp)
        for b, blk in enumerate(blocks):
            keep = [host(blk[k][i]) for k in ("c1", "c2", "res") for i in (0, 1)]
            _check(lib().af_tower_set_block(self._h, b, *[p for _, p in keep]), f"af_tower_set_block({b})")
        if stem is not None:
            (_, wp), (_, bp) = keep = [host(stem[0]), host(stem[1])]
            _check(lib().af_tower_set_stem(self._h, wp, bp), "af_tower_set_stem")
        if vconv is not None:
            keep = [host(t) for t in (vconv[0], vconv[1], pconv[0], pconv[1])]
            _check(lib().af_tower_set_heads(self._h, *[p for _, p in keep]), "af_tower_set_heads")
        self.has_dense = dense is not None
        if dense is not None:                          # (vfc1_w [4C][64], vfc1_b, vfc2_w [64][1], vfc2_b, pfc_w [16C][C], pfc_b)
            keep = [host(t) for t in dense]
            _check(lib().af_tower_set_dense(self._h, *[p for _, p in keep]), "af_tower_set_dense")
        self.policy = torch.empty((max_batch, board_size ** 2), dtype=torch.float32, device=self.device)
        self.value = torch.empty((max_batch,), dtype=torch.float32, device=self.device)
        self.vin = torch.empty((max_batch, 4 * board_size ** 2), dtype=torch.bfloat16, device=self.device)
        self.pin = torch.empty((max_batch, 16 * board_size ** 2), dtype=torch.bfloat16, device=self.device)
        self.pix = int(lib().af_tower_pix(self._h))
        self.flops_per_position = int(lib().af_tower_flops_per_position(self._h))
        # C8 activations [B][width/8][PIX][8]; zero borders are never written by the kernels
        self.x = torch.zeros((max_batch, width // 8, self.pix, 8), dtype=torch.bfloat16, device=self.device)
        self.g = torch.zeros_like(self.x)
        S = board_size
        self._xin = self.x[:, :, S:S + S * S, :].unflatten(2, (S, S))      # [B, width/8, S, S, 8] view of the board pixels

    def load_nchw(self, h):
        """h: bf16 [B, width, S, S] -> interior of the C8 buffer."""
        B = h.shape[0]
        self._xin[:B].copy_(h.view(B, self.width // 8, 8, self.S, self.S).permute(0, 1, 3, 4, 2))

    def store_nchw(self, B):
        return self._xin[:B].permute(0, 1, 4, 2, 3).reshape(B, self.width, self.S, self.S)

    def stem(self, planes):
        """planes fp32 [B,3,S,S] (contiguous) -> the C8 buffer (af_tower_stem_kernel)."""
        B = planes.shape[0]
        assert planes.is_contiguous() and planes.dtype == torch.float32 and B <= self.max_batch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().af_tower_stem(self._h, stream, planes.data_ptr(), self.x.data_ptr(), B), "af_tower_stem")

    def heads(self, B):
        """-> (vin bf16 [B, 4*S*S], pin bf16 [B, 16*S*S]) = ELU(1x1 conv) of the tower output, flattened NCHW."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().af_tower_heads(self._h, stream, self.x.data_ptr(), self.vin.data_ptr(), self.pin.data_ptr(), B),
               "af_tower_heads")
        return self.vin[:B], self.pin[:B]

    def dense(self, B):
        """vin / pin of heads() -> (policy fp32 [B, S*S] softmax, value fp32 [B]) on the MFMA dense kernel."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().af_tower_dense(self._h, stream, self.vin.data_ptr(), self.pin.data_ptr(), self.policy.data_ptr(),
                                    self.value.data_ptr(), B), "af_tower_dense")
        return self.policy[:B], self.value[:B]

    def bind_outputs(self, policy, value):
        """Write results straight into caller-owned device tensors (the engine's: no copy on the tick path)."""
        assert policy.is_contiguous() and value.is_contiguous() and policy.dtype == torch.float32
        self.policy, self.value = policy, value

    def forward(self, B):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().af_tower_forward(self._h, stream, self.x.data_ptr(), self.g.data_ptr(), B), "af_tower_forward")

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().af_tower_destroy(self._h)
            except Exception:       # interpreter shutdown
                pass
            self._h = None

    __del__ = close
