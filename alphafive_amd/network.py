"""Policy/value network of the reference (genData/network.py:52-97,163-165) rebuilt on
PyTorch-ROCm: 3xSxS -> (softmax policy[S*S], tanh(x/2) value).

    conv 5x5 3->32 +ELU                                         network.py:63
    residual(64), residual(128)                    "bone"       network.py:64-65
    value : residual(32) -> 1x1 conv 4 +ELU -> fc 64 +ELU -> fc 1 -> tanh(x/2)     :67-76
    policy: residual(64) -> residual(32) -> 1x1 conv 16 +ELU -> fc S*S -> softmax  :78-88
    residual(f,u) = ELU( conv1x1(f) + conv3x3(ELU(conv3x3(f))) )                   :52-56

Same call surface as the reference's ResNet for the self-play path: ResNet(board_size),
.eval(inputs) -> (prob, value) numpy, .restore(ckpt_path) (reads the TF checkpoint through
alphafive_amd.tensorbundle, no TensorFlow), .get_pipes(config), .close(), .graph.as_default().
On a cuda device .eval() / .eval_device() run the hand-written kernels (libaf_net.so, strictly: a missing library
raises) — so NetworkAPI's `agent_model.eval(batch)` (networkAPI.py:72), Player(pv_fn=net.eval) and the batched engine
share one arithmetic.  eval_torch() is the plain PyTorch-op evaluation of the same graph: the fp32 reference the
kernel tests compare with, the trainer's forward (alphafive_amd/train.py) and the only path of a device="cpu" net.
Parameters are kept under the checkpoint's variable names in TF layout (HWIO / [in,out]).
No torch.compile (it would emit Triton).
"""
import contextlib
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import tensorbundle

# (scope/name, kernel size, cin, cout) in graph order; residual blocks expand to _res/_conv1/_conv2
_BLOCKS = [("bone/block1", 32, 64), ("bone/block2", 64, 128), ("value/block3", 128, 32),
           ("policy/block4", 128, 64), ("policy/block5", 64, 32)]


def variable_shapes(board_size):
    C = board_size * board_size
    shapes = {"bone/conv1/kernel": (5, 5, 3, 32), "bone/conv1/bias": (32,)}
    for name, cin, cout in _BLOCKS:
        shapes[name + "_res/kernel"] = (1, 1, cin, cout)
        shapes[name + "_res/bias"] = (cout,)
        shapes[name + "_conv1/kernel"] = (3, 3, cin, cout)
        shapes[name + "_conv1/bias"] = (cout,)
        shapes[name + "_conv2/kernel"] = (3, 3, cout, cout)
        shapes[name + "_conv2/bias"] = (cout,)
    shapes.update({"value/conv/kernel": (1, 1, 32, 4), "value/conv/bias": (4,),
                   "value/fc1/kernel": (4 * C, 64), "value/fc1/bias": (64,),
                   "value/fc2/kernel": (64, 1), "value/fc2/bias": (1,),
                   "policy/conv/kernel": (1, 1, 32, 16), "policy/conv/bias": (16,),
                   "policy/fc/kernel": (16 * C, C), "policy/fc/bias": (C,)})
    return shapes


def random_variables(board_size, seed=0):
    """tf.layers defaults: glorot_uniform kernels, zero biases (network.py:53-55 passes no initializer)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shape in variable_shapes(board_size).items():
        if name.endswith("bias"):
            out[name] = np.zeros(shape, np.float32)
        else:
            rf = int(np.prod(shape[:-2])) if len(shape) == 4 else 1
            fan_in, fan_out = rf * shape[-2], rf * shape[-1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
    return out


class _Graph(object):
    """Stands in for the tf.Graph the reference's NetworkAPI enters (networkAPI.py:67)."""

    @contextlib.contextmanager
    def as_default(self):
        yield self


class ResNet(object):
    def __init__(self, board_size, graph=None, device=None, seed=0):
        self.board_size = board_size
        self.graph = graph if graph is not None else _Graph()
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self.api = None
        self.variables = {}
        self._t = {}
        import threading
        # eval_device(): one hand-written-kernel evaluator (packed weights + activation buffers for that thread's largest
        # batch: ~350 KB per position at 11x11) per calling thread, so that NetworkAPI's worker thread and the caller's never
        # share output buffers.  Kept in a registry keyed by thread: evaluators of threads that have ended are released the next
        # time one is built, close() releases all of them.
        self._hip_eval = {}
        self._hip_eval_lock = threading.Lock()
        self._eval_lock = threading.Lock()       # eval(): the returned views are copied to the host before the next call of ANY thread
        self.set_variables(random_variables(board_size, seed))

    # ---- weights ----
    def set_variables(self, variables):
        shapes = variable_shapes(self.board_size)
        for name, shape in shapes.items():
            if name not in variables:
                raise KeyError("missing variable %s" % name)
            if tuple(variables[name].shape) != shape:
                raise ValueError("variable %s has shape %s, expected %s" % (name, variables[name].shape, shape))
        self.variables = {k: np.ascontiguousarray(variables[k], np.float32) for k in shapes}
        t = {}
        for k, v in self.variables.items():
            tv = torch.from_numpy(v)
            if v.ndim == 4:
                tv = tv.permute(3, 2, 0, 1).contiguous()          # HWIO -> OIHW
            t[k] = tv.to(self.device)
        self._t = t
        self.version = getattr(self, "version", 0) + 1      # evaluators built from an older weight set reload (net_hip.make_eval)

    def restore(self, ckpt_path):
        """network.py:113-122: directory with a `checkpoint` file or a checkpoint prefix."""
        prefix = tensorbundle.resolve_checkpoint(ckpt_path)
        self.set_variables(tensorbundle.load_bundle(prefix))
        print("Successfully loaded:", prefix)

    def load_npz(self, path):
        with np.load(path) as z:
            self.set_variables({k: z[k] for k in z.files})

    def save_npz(self, path):
        np.savez(path, **self.variables)

    # ---- forward ----
    def _conv(self, x, name, act):
        k = self._t[name + "/kernel"]
        y = F.conv2d(x, k, self._t[name + "/bias"], padding=k.shape[-1] // 2)
        return F.elu(y) if act else y

    def _residual(self, f, name):
        res = self._conv(f, name + "_res", False)
        g = self._conv(f, name + "_conv1", True)
        g = self._conv(g, name + "_conv2", False)
        return F.elu(res + g)

    @torch.no_grad()
    def eval_torch(self, x):
        """Plain PyTorch ops (MIOpen convs / hipBLASLt GEMMs on ROCm): x float32[B,3,S,S] on self.device ->
        (prob float32[B,S*S], value float32[B]).  The reference evaluation, not the product path on a GPU."""
        B = x.shape[0]
        f = self._conv(x, "bone/conv1", True)
        f = self._residual(f, "bone/block1")
        f = self._residual(f, "bone/block2")
        v = self._residual(f, "value/block3")
        v = self._conv(v, "value/conv", True).reshape(B, -1)
        v = F.elu(v @ self._t["value/fc1/kernel"] + self._t["value/fc1/bias"])
        v = torch.tanh((v @ self._t["value/fc2/kernel"] + self._t["value/fc2/bias"]) / 2).squeeze(1)
        p = self._residual(f, "policy/block4")
        p = self._residual(p, "policy/block5")
        p = self._conv(p, "policy/conv", True).reshape(B, -1)
        p = torch.softmax(p @ self._t["policy/fc/kernel"] + self._t["policy/fc/bias"], dim=1)
        return p, v

    def eval_device(self, x):
        """x float32[B,3,S,S] on self.device -> (prob float32[B,S*S], value float32[B]) on device.  cuda: the hand-written
        kernels (the returned tensors are views of the calling thread's evaluator output buffers, valid until that thread's
        next call)."""
        if self.device.type == "cuda":
            # one evaluator (kernel handle + output buffers) PER THREAD: NetworkAPI's worker thread calling eval() and a
            # SelfPlayEngine(pv_device=net.eval_device) on the caller's thread never share output buffers
            return self._thread_evaluator()(x.contiguous())
        return self.eval_torch(x)

    def _thread_evaluator(self):
        import threading
        me = threading.current_thread()
        ent = self._hip_eval.get(me.ident)
        if ent is not None and ent[0] is me:
            return ent[1]
        with self._hip_eval_lock:
            for ident, (th, fn) in list(self._hip_eval.items()):
                if not th.is_alive() or ident == me.ident:     # ended threads (and a recycled ident) give their device memory back
                    fn.close()
                    del self._hip_eval[ident]
            fn = self.select_backend("hip")      # raises without libaf_net.so: no vendor-op fallback
            self._hip_eval[me.ident] = (me, fn)
        return fn

    def eval(self, inputs):
        """network.py:90-97: numpy float32[B,3,S,S] -> (prob[B,S*S], value[B]) numpy."""
        x = torch.from_numpy(np.ascontiguousarray(inputs, np.float32)).to(self.device)
        with self._eval_lock:                     # (the device evaluator returns views of its own output buffers)
            p, v = self.eval_device(x)
            return p.cpu().numpy(), v.cpu().numpy()

    # ---- backend selection for the batched engine
    def select_backend(self, name="hip"):
        """-> callable planes[B,3,S,S] -> (prob, value) on device.  "hip" (default): the hand-written MFMA kernels
        (libaf_net.so) — raises when the library is missing or the configuration is not covered, there is no silent
        fallback.  "torch": PyTorch-ROCm ops (MIOpen convs / hipBLASLt GEMMs), kept as the plain reference."""
        if name == "auto":                       # historical name: now means the product path, strictly
            name = "hip"
        if name == "hip":
            from . import net_hip
            fn = net_hip.make_eval(self)         # ImportError if libaf_net.so is not built
            if fn is None:
                raise RuntimeError("HIP net kernels need a cuda device and a board size of 3..15")
            self._backend = "hip"
            return fn
        if name != "torch":
            raise ValueError("unknown net backend %r" % (name,))
        self._backend = "torch"
        return self.eval_torch

    def flops_per_position(self):
        """2*MAC of one forward pass (direct convolution), any board size."""
        HW = self.board_size ** 2
        mac = 75 * 32 * HW + 32 * 4 * HW + 4 * HW * 64 + 64 + 32 * 16 * HW + 16 * HW * HW
        for _, cin, cout in _BLOCKS:
            mac += (9 * cin * cout + 9 * cout * cout + cin * cout) * HW
        return 2 * mac

    def roofline_info(self, pv=None):
        if getattr(self, "_backend", "torch") == "hip":
            from . import net_hip
            return net_hip.roofline_info(self.board_size)
        return {"backend": "torch-rocm (MIOpen convs + hipBLASLt GEMMs)", "peak_tflops": 157.3,
                "kernel": "net forward = 18 MIOpen conv launches + 3 GEMMs + elementwise (whole forward timed)"}

    # ---- network.py:124-134
    def get_pipes(self, config, reload=True):
        from .networkAPI import NetworkAPI
        if self.api is None:
            self.api = NetworkAPI(config, self)
            self.api.start(reload)
        return self.api.get_pipe(reload)

    def close(self):
        if self.api is not None:
            self.api.close()
        with self._hip_eval_lock:                # every thread's evaluator: kernel handles, packed weights, activation buffers
            for _, fn in self._hip_eval.values():
                fn.close()
            self._hip_eval.clear()
        # a HIP graph captured over a released evaluator's buffers must not be replayed: graphs are keyed on the weight version
        # (Player._search_batch, SelfPlayEngine.run_ticks_graph), so a net that is used again after close() re-captures
        self.version = getattr(self, "version", 0) + 1
