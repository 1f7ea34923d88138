"""Training step of the reference (genData/network.py:40-50 loss graph, main.py:38-39,62-75 loop) on
PyTorch-ROCm, so self-play -> replay -> update closes on one node without TensorFlow.  SURVEY §8f
rank 2: a caller of the hot path, not part of it.

    total = -mean(w * sum(pi * log_softmax(z)))  +  2 * mean(w * (v - z_v)^2)  +  4e-5 * sum_k ||k||^2 / 2
            (kernels only: variables whose name contains "bias" are excluded, network.py:48)
    optimiser: tf.train.AdamOptimizer(lr) semantics (beta1 .9, beta2 .999, eps 1e-8 added to sqrt(v) AFTER
               folding the bias corrections into the step size), lr from config.get_lr(step) (config.py:9,23-27)
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import tensorbundle
from .network import _BLOCKS, variable_shapes


def forward_train(params, x):
    """Differentiable forward on TF-layout parameters -> (logits[B,S*S], value[B])."""
    def conv(h, name, act):
        k = params[name + "/kernel"].permute(3, 2, 0, 1)
        y = F.conv2d(h, k, params[name + "/bias"], padding=k.shape[-1] // 2)
        return F.elu(y) if act else y

    def residual(h, name):
        return F.elu(conv(h, name + "_res", False) + conv(conv(h, name + "_conv1", True), name + "_conv2", False))

    B = x.shape[0]
    f = conv(x, "bone/conv1", True)
    f = residual(f, "bone/block1")
    f = residual(f, "bone/block2")
    v = residual(f, "value/block3")
    v = conv(v, "value/conv", True).reshape(B, -1)
    v = F.elu(v @ params["value/fc1/kernel"] + params["value/fc1/bias"])
    v = torch.tanh((v @ params["value/fc2/kernel"] + params["value/fc2/bias"]) / 2).squeeze(1)
    p = residual(f, "policy/block4")
    p = residual(p, "policy/block5")
    p = conv(p, "policy/conv", True).reshape(B, -1)
    return p @ params["policy/fc/kernel"] + params["policy/fc/bias"], v


def loss_terms(params, boards, distrib, winner, weights):
    """network.py:40-50 -> dict(total, cross_entropy, value_loss, entropy)."""
    logits, value = forward_train(params, boards)
    logsm = F.log_softmax(logits, dim=1)
    x_entropy = (distrib * logsm).sum(dim=1)
    value_sq = (value - winner) ** 2
    l2 = sum((p ** 2).sum() / 2 for n, p in params.items() if "bias" not in n and "bn" not in n)
    total = -(x_entropy * weights).mean() + 2.0 * (value_sq * weights).mean() + 4e-5 * l2
    entropy = -(F.softmax(logits, dim=1) * logsm).sum(dim=1).mean()
    return dict(total=total, cross_entropy=-x_entropy.mean(), value_loss=value_sq.mean(), entropy=entropy)


class Trainer(object):
    """Holds the variables as torch parameters (TF layout/names) + TF-style Adam slots.  The update of the 42 variables is nine
    multi-tensor launches instead of ~210 per-variable ones (at the reference's batch of 512 the step is launch-bound:
    7.4 -> 6.0 ms, tools/probe_train_step.py); step(..., metrics=False) returns None and never waits for the device."""

    def __init__(self, variables, board_size, device=None, beta1=0.9, beta2=0.999, eps=1e-8):
        self.board_size = board_size
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        shapes = variable_shapes(board_size)
        self.params = {k: torch.tensor(np.asarray(variables[k], np.float32), device=self.device, requires_grad=True)
                       for k in shapes}
        self.m = {k: torch.zeros_like(p) for k, p in self.params.items()}
        self.v = {k: torch.zeros_like(p) for k, p in self.params.items()}
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.t = 0
        self._lr_t = torch.zeros((), dtype=torch.float32, device=self.device)

    def _update(self, batch):
        """loss -> gradients -> Adam (tf.train.AdamOptimizer: lr_t folds the bias corrections, eps is added to sqrt(v)).
        Same arithmetic order as the per-variable loop it replaces: p -= (lr_t * m) / (sqrt(v) + eps)."""
        terms = loss_terms(self.params, *batch)
        ps = list(self.params.values())
        grads = list(torch.autograd.grad(terms["total"], ps))
        ms, vs = list(self.m.values()), list(self.v.values())
        with torch.no_grad():
            torch._foreach_mul_(ms, self.beta1)
            torch._foreach_add_(ms, grads, alpha=1.0 - self.beta1)
            torch._foreach_mul_(vs, self.beta2)
            torch._foreach_addcmul_(vs, grads, grads, value=1.0 - self.beta2)
            den = torch._foreach_sqrt(vs)
            torch._foreach_add_(den, self.eps)
            num = torch._foreach_mul(ms, self._lr_t)
            torch._foreach_div_(num, den)
            torch._foreach_sub_(ps, num)
        return {k: v.detach() for k, v in terms.items()}

    def step(self, boards, weights, values, policies, lr, metrics=True):
        """One optimiser step on a RandomStack.get_data batch (main.py:63-68). Returns the scalar metrics."""
        def to(a):      # numpy batches (utils.RandomStack) or device tensors (replay.DeviceRandomStack)
            if torch.is_tensor(a):
                return a.to(device=self.device, dtype=torch.float32)
            return torch.as_tensor(np.asarray(a, np.float32), device=self.device)
        batch = (to(boards), to(policies), to(values), to(weights))
        self.t += 1
        self._lr_t.fill_(float(lr * np.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)))
        terms = self._update(batch)
        return {k: float(v) for k, v in terms.items()} if metrics else None

    def variables(self):
        return {k: p.detach().cpu().numpy().copy() for k, p in self.params.items()}

    def save(self, ckpt_dir, step):
        """main.py:73-74 — a checkpoint the reference's ResNet.restore() can load."""
        os.makedirs(ckpt_dir, exist_ok=True)
        name = "alphaFive-%d" % step
        tensorbundle.save_bundle(os.path.join(ckpt_dir, name), self.variables())
        tensorbundle.write_checkpoint_state(ckpt_dir, name)


def train_loop(config, engine, net, stack, trainer, steps, log=print):
    """main.py:57-76 with the five gen_data processes replaced by the device batch `engine`
    (alphafive_amd.engine.SelfPlayEngine): every accepted episode triggers 4 minibatches once the buffer is full."""
    step = 1
    on_device = hasattr(stack, "iter_push_packed") and hasattr(engine, "post_episodes_device")
    cap = 256
    while step < steps:
        engine.run_ticks(256)
        engine.check()
        if on_device:       # episodes never leave the GPU: packed by the engine, decoded into the replay ring by one launch each
            pushes = stack.iter_push_packed(engine.post_episodes_device(cap), cap, config.gamma)
        else:
            pushes = (stack.push(data_record, result) for data_record, result in engine.pop_episodes())
        for r in pushes:                                # every finished episode reaches the buffer, also after the last step
            if r and stack.is_full() and step < steps:
                for i in range(4):                              # (only the last minibatch's scalars are logged: one sync per episode)
                    boards, weights, values, policies = stack.get_data(batch_size=config.batch_size)
                    metrics = trainer.step(boards, weights, values, policies, config.get_lr(step), metrics=i == 3)
                step += 1
                net.set_variables(trainer.variables())          # the engine's evaluator follows the trainer
                log("step: %d, xcross_loss: %0.3f, mse: %0.3f, entropy: %0.3f" %
                    (step, metrics["cross_entropy"], metrics["value_loss"], metrics["entropy"]))
                if step % 60 == 0:
                    trainer.save(config.ckpt_path, step)
        if on_device:
            # a packed append that found its buffer not to hold what the header said appends nothing and raises a device flag,
            # while the host bookkeeping has already advanced: surface it here, once per hand-off, before the ring is sampled again
            stack.check()
    return step
