#!/usr/bin/env python
"""Go / no-go for Winograd F(2x2,3x3) on fp16 split operands (VERDICT r4 next #4a) — CPU only, zero GPU minutes.

The three wide 3x3 layers of the 11x11 forward (block2.conv1 64->128, block2.conv2 128->128, block4.conv1 128->64: 62 % of the
MACs) are replaced, inside oracle/net_fp64.py's forward, by an emulation of the kernel a Winograd-on-split-operands design would run,
with the operand model of csrc/af_conv_f16s.hip:

  input   activation as the previous layer stores it: x = hi + lo, hi = fp16(x), lo = fp16(x - hi)  (22 bits)
          V = B^T d B in fp32 (two passes of single fp32 additions), then V -> hi / lo fp16 halves
  weights U = G g G^T in fp64, scaled by a power of two per layer, split from fp64 into hi / lo fp16 halves
  product three terms Uhi.Vhi + Uhi.Vlo + Ulo.Vhi; inside a 16-long k-step exact (fp64), between k-steps fp32 RNE accumulation
          (v_mfma_f32_32x32x16_f16 + fp32 accumulator), all three terms of all k-steps into one accumulator, main term first
  output  Y = A^T M A in fp32 additions, scaled back, + bias, ELU in fp64 (the epilogue's rounding is the same for both variants)

and, for comparison, by the DIRECT split-operand convolution under the same model (what af_conv_f16s computes today).  Everything else
of the forward stays fp64, so |dv| / |dp| against the pure fp64 forward is the error these three layers contribute.
Kill criterion (VERDICT): |dv| max > 5e-6 on the 512 test positions of tests/test_gpu_net.py.

    python tools/emul_wino_split.py [n_positions]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import net_fp64 as N                      # noqa: E402  (checker; this tool is not product code)

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def split16(x32):
    hi = x32.astype(np.float16)
    lo = (x32 - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def stored(x64):
    """An activation as the previous layer's epilogue leaves it in HBM: fp32 result -> hi + lo halves (22 bits)."""
    x32 = x64.astype(np.float32)
    hi, lo = split16(x32)
    return hi.astype(np.float32) + lo.astype(np.float32)


def pow2_scale(w):
    """Per-layer power of two that puts the largest weight near 2^14 (the halves stay fp16 normals where it matters)."""
    return 2.0 ** np.floor(np.log2(16384.0 / np.abs(w).max()))


def acc_products(Vs, Us, kstep=16):
    """Vs = (Vhi, Vlo) [P, T, Cin] fp16-valued f64; Us = (Uhi, Ulo) [P, Cin, Cout] -> [P, T, Cout] fp32:
    per k-step the three products summed exactly (fp64: a 16-long fp16xfp16 dot is exact in fp64), added to an fp32 accumulator."""
    Vhi, Vlo = Vs
    Uhi, Ulo = Us
    P, T, Cin = Vhi.shape
    acc = np.zeros((P, T, Uhi.shape[2]), np.float32)
    for k0 in range(0, Cin, kstep):
        sl = slice(k0, k0 + kstep)
        for a, b in ((Vhi, Uhi), (Vlo, Uhi), (Vhi, Ulo)):           # one MFMA each: fp32 rounding after every one
            acc = (acc.astype(np.float64) + a[:, :, sl] @ b[:, sl, :]).astype(np.float32)
    return acc


def conv3_direct_split(x64, kernel, bias):
    """x64 [B,Cin,11,11] fp64 (rounded to the stored 22-bit form here) -> pre-activation [B,Cout,11,11] fp64."""
    B, Cin, H, W = x64.shape
    d = stored(x64)
    dp = np.zeros((B, H + 2, W + 2, Cin), np.float32)
    dp[:, 1:-1, 1:-1] = d.transpose(0, 2, 3, 1)
    s = pow2_scale(kernel)
    w = kernel.astype(np.float64) * s                                # [3,3,Cin,Cout]
    whi = w.astype(np.float16).astype(np.float64)
    wlo = (w - whi).astype(np.float16).astype(np.float64)
    hi, lo = split16(dp)
    hi, lo = hi.astype(np.float64), lo.astype(np.float64)
    acc = np.zeros((B * H * W, kernel.shape[3]), np.float32)
    for k0 in range(0, Cin, 16):                                     # the kernel's order: slab (32 ch) outer, tap inner; here k-step outer
        for a in range(3):
            for b in range(3):
                vh = hi[:, a:a + H, b:b + W, k0:k0 + 16].reshape(-1, 16)
                vl = lo[:, a:a + H, b:b + W, k0:k0 + 16].reshape(-1, 16)
                for v, u in ((vh, whi), (vl, whi), (vh, wlo)):
                    acc = (acc.astype(np.float64) + v @ u[a, b, k0:k0 + 16]).astype(np.float32)
    out = acc.astype(np.float64) / s
    return out.reshape(B, H, W, -1).transpose(0, 3, 1, 2) + bias.astype(np.float64)[None, :, None, None]


def conv3_wino_split(x64, kernel, bias, split_weights_from="fp64", stats=None):
    B, Cin, H, W = x64.shape
    Cout = kernel.shape[3]
    d = stored(x64)                                                  # fp32 [B,Cin,11,11]
    nt = (H + 1) // 2                                                # 6 tiles per side
    dp = np.zeros((B, Cin, 2 * nt + 2, 2 * nt + 2), np.float32)      # 14x14, origin -1
    dp[:, :, 1:1 + H, 1:1 + W] = d
    # tiles [B,Cin,nt,nt,4,4]
    t = np.stack([np.stack([dp[:, :, i:i + 2 * nt:2, j:j + 2 * nt:2] for j in range(4)], -1) for i in range(4)], -2)
    # rows (fp32, one operation per element), then columns
    r = np.stack([t[..., 0, :] - t[..., 2, :], t[..., 1, :] + t[..., 2, :], t[..., 2, :] - t[..., 1, :], t[..., 1, :] - t[..., 3, :]], -2)
    V = np.stack([r[..., :, 0] - r[..., :, 2], r[..., :, 1] + r[..., :, 2], r[..., :, 2] - r[..., :, 1], r[..., :, 1] - r[..., :, 3]], -1)
    assert V.dtype == np.float32
    Vhi, Vlo = split16(V)
    if stats is not None:
        stats["V_absmax"] = max(stats.get("V_absmax", 0.0), float(np.abs(V).max()))
    # [16 points, B*nt*nt tiles, Cin]
    def pts(x):
        return x.astype(np.float64).transpose(4, 5, 0, 2, 3, 1).reshape(16, B * nt * nt, Cin)
    U = np.einsum("ia,abco,jb->ijco", G, kernel.astype(np.float64), G)          # [4,4,Cin,Cout] fp64
    s = pow2_scale(U)
    U = (U * s).reshape(16, Cin, Cout)
    if split_weights_from == "fp32":
        U = U.astype(np.float32).astype(np.float64)
    Uhi = U.astype(np.float16).astype(np.float64)
    Ulo = (U - Uhi).astype(np.float16).astype(np.float64)
    M = acc_products((pts(Vhi), pts(Vlo)), (Uhi, Ulo))                          # [16, tiles, Cout] fp32
    M = M.reshape(4, 4, B, nt, nt, Cout)
    # output transform in fp32: rows then columns
    r0 = (M[0] + M[1]) + M[2]
    r1 = (M[1] - M[2]) - M[3]                                                    # each [4(cols), B,nt,nt,Cout]
    def cols(rr):
        return (rr[0] + rr[1]) + rr[2], (rr[1] - rr[2]) - rr[3]
    y00, y01 = cols(r0)
    y10, y11 = cols(r1)
    assert y00.dtype == np.float32
    Y = np.zeros((B, 2 * nt, 2 * nt, Cout), np.float64)
    Y[:, 0::2, 0::2], Y[:, 0::2, 1::2], Y[:, 1::2, 0::2], Y[:, 1::2, 1::2] = y00, y01, y10, y11
    out = Y[:, :H, :W] / s
    return out.transpose(0, 3, 1, 2) + bias.astype(np.float64)[None, :, None, None]


def forward(variables, inputs, conv3):
    """net_fp64.forward with the three wide 3x3 convolutions computed by `conv3` (None: all fp64)."""
    v = variables
    c64 = lambda x, name: N._conv2d(x, v[name + "/kernel"], v[name + "/bias"])             # noqa: E731
    cw = (lambda x, name: conv3(x, v[name + "/kernel"], v[name + "/bias"])) if conv3 else c64  # noqa: E731
    x = np.asarray(inputs, np.float64)
    B = x.shape[0]
    f = N._elu(c64(x, "bone/conv1"))
    f = N._residual(f, v, "bone/block1")
    # block2: both 3x3 convolutions wide
    res = N._conv2d(f, v["bone/block2_res/kernel"], v["bone/block2_res/bias"], same=False)
    g = N._elu(cw(f, "bone/block2_conv1"))
    g = cw(g, "bone/block2_conv2")
    f = N._elu(res + g)
    val = N._residual(f, v, "value/block3")
    val = N._elu(N._conv2d(val, v["value/conv/kernel"], v["value/conv/bias"])).reshape(B, -1)
    val = N._elu(val @ v["value/fc1/kernel"].astype(np.float64) + v["value/fc1/bias"])
    val = np.tanh((val @ v["value/fc2/kernel"].astype(np.float64) + v["value/fc2/bias"]) / 2)[:, 0]
    # block4: conv1 wide
    res = N._conv2d(f, v["policy/block4_res/kernel"], v["policy/block4_res/bias"], same=False)
    g = N._elu(cw(f, "policy/block4_conv1"))
    g = c64(g, "policy/block4_conv2")
    pol = N._elu(res + g)
    pol = N._residual(pol, v, "policy/block5")
    pol = N._elu(N._conv2d(pol, v["policy/conv/kernel"], v["policy/conv/bias"]))
    logits = pol.reshape(B, -1) @ v["policy/fc/kernel"].astype(np.float64) + v["policy/fc/bias"]
    logits = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    return e / e.sum(axis=1, keepdims=True), val


def main():
    from test_gpu_net import _positions
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    with np.load(os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz")) as z:
        variables = {k: z[k] for k in z.files}
    x = _positions(11, 512)[:n]
    t0 = time.time()
    p64, v64 = N.forward(variables, x)
    stats = {}
    res = {}
    for name, fn in (("direct split (today's operand model)", conv3_direct_split),
                     ("Winograd F(2x2,3x3) split, weights split from fp64", lambda a, k, b: conv3_wino_split(a, k, b, "fp64", stats)),
                     ("Winograd F(2x2,3x3) split, weights rounded to fp32 first", lambda a, k, b: conv3_wino_split(a, k, b, "fp32"))):
        dv, dp = [], []
        for b0 in range(0, n, 32):
            p, v = forward(variables, x[b0:b0 + 32], fn)
            dv.append(np.abs(v - v64[b0:b0 + 32]))
            dp.append(np.abs(p - p64[b0:b0 + 32]).max(axis=1))
        dv, dp = np.concatenate(dv), np.concatenate(dp)
        res[name] = (dv, dp)
        print("%-58s |dv| max %.3g mean %.3g   |dp| max %.3g mean %.3g   (%d positions, %.0f s)"
              % (name, dv.max(), dv.mean(), dp.max(), dp.mean(), n, time.time() - t0), flush=True)
    print("transformed-input |V| max %.1f (fp16 max 65504)" % stats.get("V_absmax", 0.0))
    dvw = res["Winograd F(2x2,3x3) split, weights split from fp64"][0].max()
    print("verdict: %s (kill criterion |dv| max > 5e-6)" % ("SURVIVES" if dvw <= 5e-6 else "KILLED"))


if __name__ == "__main__":
    main()
