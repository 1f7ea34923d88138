"""fp64 numpy restatement of the reference network's forward pass (genData/network.py:52-97,
163-165) — the checker for "value within 1e-5 fp32".  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against real TensorFlow: the arithmetic lives in TF 1.x (un-vendored, no
version pinned in the reference, not installable here) and the reference holds no stored
network outputs.  What is pinned: the checkpoint weights (ckpt/alphaFive-6960, read without TF)
and the graph definition restated here line by line: NCHW, HWIO kernels, SAME/VALID padding,
ELU(alpha=1), flatten in NCHW order, tanh(x/2), softmax.
"""
import numpy as np


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def _conv2d(x, kernel, bias, same=True):
    """x [B,Cin,H,W] f64, kernel HWIO -> [B,Cout,H,W]; SAME zero padding (odd kernels) or VALID 1x1."""
    kh, kw, cin, cout = kernel.shape
    B, _, H, W = x.shape
    ph, pw = (kh // 2, kw // 2) if same else (0, 0)
    xp = np.zeros((B, H + 2 * ph, W + 2 * pw, cin), np.float64)       # NHWC: every tap is one fp64 matrix product
    xp[:, ph:ph + H, pw:pw + W, :] = x.transpose(0, 2, 3, 1)
    k64 = kernel.astype(np.float64)
    out = np.zeros((B, H, W, cout), np.float64)
    for a in range(kh):
        for b in range(kw):
            out += xp[:, a:a + H, b:b + W, :] @ k64[a, b]
    return out.transpose(0, 3, 1, 2) + bias.astype(np.float64)[None, :, None, None]


def _residual(f, v, name):                                        # network.py:52-56
    res = _conv2d(f, v[name + "_res/kernel"], v[name + "_res/bias"], same=False)
    g = _elu(_conv2d(f, v[name + "_conv1/kernel"], v[name + "_conv1/bias"]))
    g = _conv2d(g, v[name + "_conv2/kernel"], v[name + "_conv2/bias"])
    return _elu(res + g)


def forward(variables, inputs):
    """variables: {tf name: ndarray}; inputs [B,3,S,S] -> (prob [B,S*S], value [B]) in float64."""
    v = variables
    x = np.asarray(inputs, np.float64)
    B = x.shape[0]
    f = _elu(_conv2d(x, v["bone/conv1/kernel"], v["bone/conv1/bias"]))           # :63
    f = _residual(f, v, "bone/block1")                                          # :64
    f = _residual(f, v, "bone/block2")                                          # :65
    val = _residual(f, v, "value/block3")                                       # :68
    val = _elu(_conv2d(val, v["value/conv/kernel"], v["value/conv/bias"]))      # :70
    val = val.reshape(B, -1)                                                    # :71-72 (NCHW flatten)
    val = _elu(val @ v["value/fc1/kernel"].astype(np.float64) + v["value/fc1/bias"])   # :73
    val = np.tanh((val @ v["value/fc2/kernel"].astype(np.float64) + v["value/fc2/bias"]) / 2)[:, 0]   # :76,163
    pol = _residual(f, v, "policy/block4")                                      # :79
    pol = _residual(pol, v, "policy/block5")                                    # :80
    pol = _elu(_conv2d(pol, v["policy/conv/kernel"], v["policy/conv/bias"]))    # :82
    logits = pol.reshape(B, -1) @ v["policy/fc/kernel"].astype(np.float64) + v["policy/fc/bias"]   # :85
    logits = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    return e / e.sum(axis=1, keepdims=True), val                                # :88


def loss_terms(variables, boards, distrib, winner, weights):
    """fp64 restatement of the loss graph (network.py:40-50): total, cross_entropy, value_loss, entropy."""
    v = variables
    x = np.asarray(boards, np.float64)
    B = x.shape[0]
    f = _elu(_conv2d(x, v["bone/conv1/kernel"], v["bone/conv1/bias"]))
    f = _residual(f, v, "bone/block1")
    f = _residual(f, v, "bone/block2")
    val = _residual(f, v, "value/block3")
    val = _elu(_conv2d(val, v["value/conv/kernel"], v["value/conv/bias"])).reshape(B, -1)
    val = _elu(val @ v["value/fc1/kernel"].astype(np.float64) + v["value/fc1/bias"])
    val = np.tanh((val @ v["value/fc2/kernel"].astype(np.float64) + v["value/fc2/bias"]) / 2)[:, 0]
    pol = _residual(f, v, "policy/block4")
    pol = _residual(pol, v, "policy/block5")
    pol = _elu(_conv2d(pol, v["policy/conv/kernel"], v["policy/conv/bias"]))
    logits = pol.reshape(B, -1) @ v["policy/fc/kernel"].astype(np.float64) + v["policy/fc/bias"]
    z = logits - logits.max(axis=1, keepdims=True)
    logsm = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    x_entropy = (np.asarray(distrib, np.float64) * logsm).sum(axis=1)                       # :41
    w = np.asarray(weights, np.float64)
    value_sq = (val - np.asarray(winner, np.float64)) ** 2                                  # :44
    l2 = sum((np.asarray(a, np.float64) ** 2).sum() / 2 for n, a in v.items() if "bias" not in n and "bn" not in n)  # :47-48
    total = -(x_entropy * w).mean() + 2.0 * (value_sq * w).mean() + 4e-5 * l2               # :50
    entropy = -(np.exp(logsm) * logsm).sum(axis=1).mean()
    return dict(total=total, cross_entropy=-x_entropy.mean(), value_loss=value_sq.mean(), entropy=entropy)
