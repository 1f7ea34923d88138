"""BASELINE configs[4]: a deeper policy/value net — N residual blocks of the reference's block type
(network.py:52-56: 1x1 projection || 3x3+ELU -> 3x3, add, ELU) at constant width, the reference's two heads,
evaluated in bf16.  The whole forward runs on the hand-written MFMA kernels of csrc/af_tower_bf16.hip
(`select_backend("hip")`, the bench default): 5x5 stem, residual tower (99 % of the FLOPs), the heads' 1x1
convolutions and the three dense layers + softmax.  `eval_device` is the
all-PyTorch reference path.  Performance-only configuration (no checkpoint exists for
it, random init, no bit parity); it plugs into SelfPlayEngine through the same
planes[G,3,S,S] -> (prob[G,C], value[G]) evaluator seam as the fp32 net.
"""
import numpy as np
import torch
import torch.nn.functional as F


class _HipEvaluator(object):
    """planes -> (prob, value) on the hand-written kernels; bind_outputs lets the engine own the result tensors."""

    def __init__(self, net):
        self.net = net

    def __call__(self, x):
        return self.net.eval_hip(x)

    def bind_outputs(self, policy, value):
        if policy.shape[0] >= self.net._tower.max_batch:
            self.net._tower.bind_outputs(policy, value)

    def weights_version(self):
        """What a captured graph of this evaluator is keyed on (SelfPlayEngine.run_ticks_graph): the tower packs its weights
        once, at select_backend(); every (re)build of the tower bumps the net's pack counter (a monotone count, not id(): CPython
        may hand a freed tower's id to its successor)."""
        return self.net._pack_version


class DeepResNet(object):
    def __init__(self, board_size, blocks=8, width=128, device="cuda", dtype=torch.bfloat16, seed=0):
        self.board_size, self.blocks, self.width = board_size, blocks, width
        self.device, self.dtype = torch.device(device), dtype
        g = torch.Generator().manual_seed(seed)
        C = board_size * board_size

        def glorot(*shape):                      # OIHW / [in,out], tf.layers default initialiser
            rf = int(np.prod(shape[2:])) if len(shape) == 4 else 1
            fan_in, fan_out = (shape[1] * rf, shape[0] * rf) if len(shape) == 4 else (shape[0], shape[1])
            lim = float(np.sqrt(6.0 / (fan_in + fan_out)))
            return ((torch.rand(shape, generator=g) * 2 - 1) * lim).to(self.device, dtype)

        z = lambda n: torch.zeros(n, device=self.device, dtype=dtype)  # noqa: E731
        self.stem = (glorot(width, 3, 5, 5), z(width))
        self.tower = [dict(res=(glorot(width, width, 1, 1), z(width)), c1=(glorot(width, width, 3, 3), z(width)),
                           c2=(glorot(width, width, 3, 3), z(width))) for _ in range(blocks)]
        self.vconv, self.pconv = (glorot(4, width, 1, 1), z(4)), (glorot(16, width, 1, 1), z(16))
        self.vfc1, self.vfc2 = (glorot(4 * C, 64), z(64)), (glorot(64, 1), z(1))
        self.pfc = (glorot(16 * C, C), z(C))

    def flops_per_position(self):
        HW, w = self.board_size ** 2, self.width
        mac = 75 * w * HW + self.blocks * (w * w + 18 * w * w) * HW + (4 + 16) * w * HW + 4 * HW * 64 + 64 + 16 * HW * HW
        return 2 * mac

    @torch.no_grad()
    def eval_device(self, x, dtype=None):
        """All-PyTorch evaluation in self.dtype (bf16), or — dtype=torch.float32 — of the SAME (bf16-valued) weights in fp32
        arithmetic with fp32 activations: the yardstick both bf16 paths' errors are measured against (tests/test_gpu_realnet.py)."""
        dt = self.dtype if dtype is None else dtype
        c = (lambda wb: (wb[0].to(dt), wb[1].to(dt)))
        B = x.shape[0]
        h = F.elu(F.conv2d(x.to(dt), *c(self.stem), padding=2))
        for blk in self.tower:
            r = F.conv2d(h, *c(blk["res"]))
            g = F.elu(F.conv2d(h, *c(blk["c1"]), padding=1))
            h = F.elu(r + F.conv2d(g, *c(blk["c2"]), padding=1))
        v = F.elu(F.conv2d(h, *c(self.vconv))).reshape(B, -1)
        v = F.elu(v @ self.vfc1[0].to(dt) + self.vfc1[1].to(dt))
        v = torch.tanh((v @ self.vfc2[0].to(dt) + self.vfc2[1].to(dt)).float() / 2).squeeze(1)
        p = F.elu(F.conv2d(h, *c(self.pconv))).reshape(B, -1)
        p = torch.softmax((p @ self.pfc[0].to(dt) + self.pfc[1].to(dt)).float(), dim=1)
        return p, v

    # ---- hand-written tower (csrc/af_tower_bf16.hip) ----
    def select_backend(self, name, max_batch):
        """"hip": torch stem -> af_tower_forward -> torch heads (raises if libaf_tower.so is missing);
        "torch": the all-PyTorch path."""
        if name == "torch":
            return self.eval_device
        if name != "hip":
            raise ValueError(name)
        from . import tower_hip
        self._pack_version = getattr(self, "_pack_version", 0) + 1
        self._tower = tower_hip.HipTower(self.tower, self.board_size, self.width, max_batch, self.device,
                                         stem=self.stem, vconv=self.vconv, pconv=self.pconv,
                                         dense=(self.vfc1[0], self.vfc1[1], self.vfc2[0], self.vfc2[1], self.pfc[0], self.pfc[1]))
        return _HipEvaluator(self)

    @torch.no_grad()
    def eval_hip(self, x):
        """The whole forward on hand-written kernels: af_tower_stem -> af_tower_forward -> af_tower_heads (the 1x1 convs)
        -> af_tower_dense (the three dense layers + softmax on the MFMA)."""
        B, tw = x.shape[0], self._tower
        tw.stem(x.contiguous())
        tw.forward(B)
        tw.heads(B)
        return tw.dense(B)

    @torch.no_grad()
    def eval_hip_torch_dense(self, x):
        """A/B reference for af_tower_dense: the same pipeline with the dense layers on PyTorch ops."""
        B, tw = x.shape[0], self._tower
        tw.stem(x.contiguous())
        tw.forward(B)
        vin, pin = tw.heads(B)
        v = F.elu(vin @ self.vfc1[0] + self.vfc1[1])
        v = torch.tanh((v @ self.vfc2[0] + self.vfc2[1]).float() / 2).squeeze(1)
        p = torch.softmax((pin @ self.pfc[0] + self.pfc[1]).float(), dim=1)
        return p, v

    @torch.no_grad()
    def eval_hip_torch_ends(self, x):
        """The hand-written tower between PyTorch stem and heads (A/B reference for the stem / heads kernels)."""
        B, tw = x.shape[0], self._tower
        tw.load_nchw(F.elu(F.conv2d(x.to(self.dtype), self.stem[0], self.stem[1], padding=2)))
        tw.forward(B)
        h = tw.store_nchw(B)
        v = F.elu(F.conv2d(h, *self.vconv)).reshape(B, -1)
        v = F.elu(v @ self.vfc1[0] + self.vfc1[1])
        v = torch.tanh((v @ self.vfc2[0] + self.vfc2[1]).float() / 2).squeeze(1)
        p = F.elu(F.conv2d(h, *self.pconv)).reshape(B, -1)
        p = torch.softmax((p @ self.pfc[0] + self.pfc[1]).float(), dim=1)
        return p, v

    @torch.no_grad()
    def tower_reference(self, h):
        """The tower alone on PyTorch ops (bf16 in/out NCHW) — what af_tower_forward replaces."""
        for blk in self.tower:
            r = F.conv2d(h, *blk["res"])
            g = F.elu(F.conv2d(h, *blk["c1"], padding=1))
            h = F.elu(r + F.conv2d(g, *blk["c2"], padding=1))
        return h
