"""Deterministic integer-hash pseudo-net (test stand-in for ResNet.eval).

Spec (must equal oracle/af_oracle.c:afo_pseudonet): all arithmetic is 32-bit
integer; outputs are integers scaled by powers of two, hence exactly
representable in fp32 and bit-identical on numpy / torch-CPU / torch-GPU.

    mix32(x): x^=x>>16; x*=0x45d9f3b; x^=x>>16; x*=0x45d9f3b; x^=x>>16   (mod 2^32)
    h = salt + sum_c mine_c*mix32(c+1) + theirs_c*mix32(c+1001) + last_c*mix32(c+2001)
    m_c = mix32(h ^ mix32(c+3001));  k_c = 1 + (m_c & 1023) + (peak if (m_c>>10)&7 == 0 else 0)
    policy_c = k_c / 2^17 ;  value = ((mix32(h ^ 0x9e3779b9) & 0xffff) - 32768) / 2^16

vbits (default 16, what the oracle's built-in copy has): value = ((mix32(..) & (2^vbits - 1)) - 2^(vbits-1)) / 2^vbits.
With vbits = 24 the values use the whole fp32 mantissa, so an fp32 running sum of them rounds where an fp64 one does
not: the pipe-path goldens (W as a python float, networkAPI.py:72) use that.

sharp=1 (the tie-free traces, tests/golden/mcts_*_sharp.npz): policy_c = ((65536 + (m_c & 0xffff)) << ((m_c >> 16) & 7)) / 2^24 —
integers below 2^24, so still exact in fp32, but with a 16-bit fraction (two cells of a position hardly ever share a prior:
1.4 % of positions have such a pair at 11x11, against 7 pairs per position with the 10-bit k_c above) and priors spread over
seven octaves, so that a search has a clear favourite and max-visit ties at the root are rare.  With no uniform pick among
several candidates anywhere (oracle tie_stats() all zero) a search is a function of the net alone, whatever the generator.
"""
import numpy as np

M32 = 0xFFFFFFFF


def _mix32_np(x):
    x = x.astype(np.int64) & M32
    x ^= x >> 16
    x = (x * 0x45d9f3b) & M32
    x ^= x >> 16
    x = (x * 0x45d9f3b) & M32
    x ^= x >> 16
    return x


def pseudonet_np(planes, salt=0, peak=0, vbits=16, sharp=0):
    x = np.asarray(planes)
    B = x.shape[0]
    C = x.shape[2] * x.shape[3]
    bits = (x.reshape(B, 3, C) != 0).astype(np.int64)
    c = np.arange(C, dtype=np.int64)
    tab = np.stack([_mix32_np(c + 1), _mix32_np(c + 1001), _mix32_np(c + 2001)])       # [3,C]
    h = (salt + (bits * tab[None]).sum(axis=(1, 2))) & M32                              # [B]
    m = _mix32_np(h[:, None] ^ _mix32_np(c + 3001)[None, :])                            # [B,C]
    k = 1 + (m & 0x3FF) + np.where(((m >> 10) & 7) == 0, peak, 0)
    policy = k.astype(np.float32) * np.float32(1.0 / 131072.0)
    if sharp:
        policy = ((65536 + (m & 0xFFFF)) << ((m >> 16) & 7)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    mv = _mix32_np(h ^ 0x9E3779B9)
    value = ((mv & ((1 << vbits) - 1)) - (1 << (vbits - 1))).astype(np.float32) * np.float32(1.0 / (1 << vbits))
    return policy, value


def _mix32_t(x):
    x = x & M32
    x = x ^ (x >> 16)
    x = (x * 0x45d9f3b) & M32
    x = x ^ (x >> 16)
    x = (x * 0x45d9f3b) & M32
    x = x ^ (x >> 16)
    return x


def pseudonet_torch(planes, salt=0, peak=0, vbits=16, sharp=0):
    """planes: float32[B,3,S,S] torch tensor (any device) -> (policy[B,C], value[B]) float32."""
    import torch
    x = planes
    B = x.shape[0]
    C = x.shape[2] * x.shape[3]
    dev = x.device
    bits = (x.reshape(B, 3, C) != 0).to(torch.int64)
    c = torch.arange(C, dtype=torch.int64, device=dev)
    tab = torch.stack([_mix32_t(c + 1), _mix32_t(c + 1001), _mix32_t(c + 2001)])
    h = (salt + (bits * tab[None]).sum(dim=(1, 2))) & M32
    m = _mix32_t(h[:, None] ^ _mix32_t(c + 3001)[None, :])
    k = 1 + (m & 0x3FF) + torch.where(((m >> 10) & 7) == 0, torch.full_like(m, peak), torch.zeros_like(m))
    policy = k.to(torch.float32) * (1.0 / 131072.0)
    if sharp:
        policy = ((65536 + (m & 0xFFFF)) << ((m >> 16) & 7)).to(torch.float32) * (1.0 / 16777216.0)
    mv = _mix32_t(h ^ 0x9E3779B9)
    value = ((mv & ((1 << vbits) - 1)) - (1 << (vbits - 1))).to(torch.float32) * (1.0 / (1 << vbits))
    return policy, value
