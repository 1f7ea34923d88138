"""ctypes binding of libaf_net.so (include/af_net.h): the hand-written MFMA forward pass.
Used by ResNet.select_backend("hip"/"auto"); raises if the library is missing (no fallback here —
the caller decides whether the torch-op path is acceptable)."""
import ctypes as C
import os

import numpy as np
import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("AF_NET_LIB") or os.path.join(_PKG, "_lib", "libaf_net.so")
_lib = None


class NetError(RuntimeError):
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
        vp = C.c_void_p
        L.af_net_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.af_net_destroy.argtypes = [vp]
        L.af_net_destroy.restype = None
        L.af_net_set_variable.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int64]
        L.af_net_finalize.argtypes = [vp]
        L.af_net_forward.argtypes = [vp, vp, vp, C.c_int32, vp, vp]
        L.af_net_flops_per_position.argtypes = [vp]
        L.af_net_flops_per_position.restype = C.c_int64
        L.af_net_small_forward_error.argtypes = [vp]
        L.af_net_strerror.argtypes = [C.c_int]
        L.af_net_strerror.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise NetError(f"{what}: {lib().af_net_strerror(rc).decode()} (code {rc})")


class HipNet(object):
    """One af_net handle sized for `max_batch` positions on `device`."""

    def __init__(self, variables, board_size, max_batch, device):
        self.S, self.max_batch, self.device = board_size, max_batch, torch.device(device)
        self._h = C.c_void_p()
        self._version = 0
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _check(lib().af_net_create(board_size, max_batch, idx, C.byref(self._h)), "af_net_create")
        self.load(variables)
        self.flops_per_position = int(lib().af_net_flops_per_position(self._h))
        self.policy = torch.empty((max_batch, board_size * board_size), dtype=torch.float32, device=self.device)
        self.value = torch.empty((max_batch,), dtype=torch.float32, device=self.device)

    def load(self, variables):
        """(Re)load a weight set: af_net_set_variable for every tensor, then af_net_finalize repacks and uploads."""
        torch.cuda.synchronize(self.device)             # no forward may still be reading the old packed weights
        for name, arr in variables.items():
            a = np.ascontiguousarray(arr, np.float32)
            _check(lib().af_net_set_variable(self._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size),
                   f"af_net_set_variable({name})")
        _check(lib().af_net_finalize(self._h), "af_net_finalize")
        self._version += 1

    def weights_version(self):
        """Count of load() calls: what a captured graph over this handle (SelfPlayEngine.run_ticks_graph) is keyed on."""
        return self._version

    def bind_outputs(self, policy, value):
        """Write results straight into caller-owned device tensors (no copy on the tick path)."""
        assert policy.is_contiguous() and value.is_contiguous() and policy.dtype == torch.float32
        assert policy.shape[0] >= self.max_batch or policy.shape[0] == value.shape[0]
        self.policy, self.value = policy, value

    def __call__(self, planes):
        B = planes.shape[0]
        assert B <= self.max_batch and planes.is_contiguous() and planes.dtype == torch.float32
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().af_net_forward(self._h, stream, planes.data_ptr(), B, self.policy.data_ptr(), self.value.data_ptr()),
               "af_net_forward")
        return self.policy[:B], self.value[:B]

    def small_forward_error(self):
        """1 if a role of the single-launch small-batch forward ever gave up waiting for its producers (synchronises)."""
        return int(lib().af_net_small_forward_error(self._h))

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().af_net_destroy(self._h)
            except Exception:       # interpreter shutdown
                pass
            self._h = None

    __del__ = close


def make_eval(resnet):
    """-> callable planes -> (prob, value) backed by the HIP kernels, or None when not applicable."""
    if resnet.device.type != "cuda" or not (3 <= resnet.board_size <= 15):
        return None
    lib()
    state = {"net": None, "out": None, "version": None}

    def pv(planes):
        B = planes.shape[0]
        if state["net"] is None or state["net"].max_batch < B:
            if state["net"] is not None:
                state["net"].close()
            state["net"] = HipNet(resnet.variables, resnet.board_size, B, resnet.device)
            state["version"] = getattr(resnet, "version", None)
            if state["out"] is not None and state["out"][0].shape[0] >= B:
                state["net"].bind_outputs(*state["out"])
        elif state["version"] != getattr(resnet, "version", None):
            state["net"].load(resnet.variables)          # set_variables / restore / load_npz since the last call (a weight update)
            state["version"] = getattr(resnet, "version", None)
        return state["net"](planes)

    def bind_outputs(policy, value):
        state["out"] = (policy, value)
        if state["net"] is not None and policy.shape[0] >= state["net"].max_batch:
            state["net"].bind_outputs(policy, value)

    def close():
        if state["net"] is not None:
            state["net"].close()
            state["net"] = None

    pv.bind_outputs = bind_outputs
    pv.close = close
    pv.weights_version = lambda: getattr(resnet, "version", None)      # what a captured graph of this evaluator is keyed on
    return pv


# 3x3 + folded 1x1 convolution MACs per pixel of the five residual blocks (network.py:52-88): the part of the forward
# pass that runs on af_conv_f16s, each MAC issued as three fp16 MFMA MACs (hi*hi, hi*lo, lo*hi)
_SPLIT_MAC_PER_PIXEL = sum(9 * ci * co + 9 * co * co + ci * co for ci, co in ((32, 64), (64, 128), (128, 32), (128, 64), (64, 32)))
# HBM bytes per position of that path (S32 activations: a 32-channel slab is 16,384 B as the LDS-DMA reads it and 15,488 B
# as the epilogue writes it (121 pixels x 8 unit rows x 16 B); weights stay in registers):
#   slab reads   1 | 2+1 | 2 | 4+2 | 4 | 1 | 4 | 2+4 | 2 | 1 = 30  (3x3 input + block input of a folded 1x1 projection; block2's conv2
#                reads its position from two workgroup kinds, the second read is an L2 hit)
#   slab writes  stem 1 | 2 | 2 | 4 | 4 | 1 | - | 2 | 2 | 1 | - = 19
#   blocks 3 and 5: the separately computed projection, fp32 in accumulator layout (16,384 B), written by conv1 and read by conv2
#   head inputs: each branch's last conv also applies the head's 1x1 convolution and writes 4 (value) / 16 (policy) channels of
#                121 pixels as fp16 halves (hi + lo), read once by the dense kernels; input planes; outputs
# r5: blocks 3 and 5 run as ONE kernel each (af_block_f16s): their 32-channel intermediate and their projection never leave the CU —
#   slab reads 30 -> 28, slab writes 19 -> 17, no fp32 projection buffers
_SPLIT_BYTES_PER_POSITION = (28 * 16384 + 17 * 15488 + 2 * (4 + 16) * 121 * 2 * 2 + 3 * 121 * 4 + 122 * 4)


_conv_mode = 5          # af_net_tune(0, .) as last set through tune(): 5 = the split-operand path (the library's default on 11x11 / 15x15)


def roofline_info(board_size=11):
    if board_size == 15 and _conv_mode == 5:
        # the same kernels on two half-board pseudo-positions per board (8 pixel tiles of 32 for 225 pixels): MFMA work per
        # position = 256 / 121 of the 11x11 figure; HBM slabs are 32 KB (reads: 2 x 20 KB windows, writes 2 x 15.5 / 12.4 KB)
        return {"backend": "hip (af_conv_f16s.hip on 15x15: two half-board pseudo-positions per board, fp16 split operands, fp32 "
                           "accumulation; MFMA stem and fused heads as on 11x11)",
                "kernel": "af_net_forward = af_stem_mfma_f16s<Geo<15>> + 10x af_conv_f16s<Geo<15>> (the last conv of each branch also "
                          "applies the head's 1x1 convolution) + af_value_fc_f16s<Geo<15>> + af_policy_fc_f16s<Geo<15>> (whole "
                          "forward timed)",
                "peak_tflops": 2500.0, "issued_flop_per_position": 2 * 3 * _SPLIT_MAC_PER_PIXEL * 256,
                "algorithmic_bytes_per_position": (30 * 2 * 20480 + 19 * 30720 + 2 * 2 * 2 * 16384 + 2 * (4 + 16) * 225 * 2 * 2 + 3 * 225 * 4 + 226 * 4)}
    if board_size == 11 and _conv_mode == 5:
        return {"backend": "hip (af_conv_f16s.hip: stem, convs and heads on v_mfma_f32_32x32x16_f16 with fp16 split operands and "
                           "fp32 accumulation; convs weight-stationary with an LDS-DMA slab ring)",
                "kernel": "af_net_forward = af_stem_mfma_f16s + 6x af_conv_f16s + 2x af_block_f16s (blocks 3 and 5: both convolutions "
                          "and the head's 1x1 convolution in one kernel, the intermediate in LDS) + af_value_fc_f16s + af_policy_fc_f16s "
                          "(whole forward timed; the convolution kernels carry 99 % of the algorithmic FLOPs, each MAC issued as 3 fp16 MFMA MACs)",
                "peak_tflops": 2500.0, "issued_flop_per_position": 2 * 3 * _SPLIT_MAC_PER_PIXEL * 121,
                "algorithmic_bytes_per_position": _SPLIT_BYTES_PER_POSITION}
    return {"backend": "hip (af_net.hip: fp32 MFMA Winograd F(2x2,3x3) convs, fused transforms/bias/ELU/residual)",
            "kernel": "af_net_forward = af_stem_conv + 10x af_conv_wino + af_value_head + af_policy_head "
                      "(whole forward timed; af_conv_wino carries 97 % of the algorithmic FLOPs; achieved = "
                      "direct-convolution FLOPs / time, the MFMAs issued are 2.25x fewer on the 3x3 layers)",
            "peak_tflops": 157.3}


def tune(key, value):
    """Benchmark / A-B knob (af_net_tune, include/af_net.h): key 0 = conv path (5 fp16 split-operand implicit GEMM = default on
    11x11 / 15x15, 1 fp32 MFMA Winograd); key 7 = A/B and profiling bits of the split-operand path; key 9 = fused heads (1) or the
    fp32 head kernels (0); key 4 / 5 = side stream / MFMA policy head of the fp32 path.  Unknown keys and values raise NetError."""
    global _conv_mode
    lib().af_net_tune.argtypes = [C.c_int32, C.c_int32]
    _check(lib().af_net_tune(key, value), "af_net_tune")
    if key == 0:
        _conv_mode = int(value)
