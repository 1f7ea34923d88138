"""Network rebuild (alphafive_amd.network) on torch-CPU vs the fp64 restatement; checkpoint reader."""
import os

import numpy as np
import pytest

from alphafive_amd import tensorbundle
from alphafive_amd.network import ResNet, variable_shapes
from conftest import GOLDEN
from oracle import net_fp64

W = os.path.join(GOLDEN, "alphaFive-6960.weights.npz")


def _positions(S, B, seed=0):
    rng = np.random.RandomState(seed)
    xs = np.zeros((B, 3, S, S), np.float32)
    for b in range(B):
        cells = rng.permutation(S * S)[:2 * b + 1]
        for t, c in enumerate(cells):
            xs[b, t % 2, c // S, c % S] = 1
        xs[b, 2, cells[-1] // S, cells[-1] % S] = 1
    return xs


def test_fp32_forward_within_1e5_of_fp64_restatement():
    net = ResNet(11, device="cpu")
    net.load_npz(W)
    xs = _positions(11, 6)
    p, v = net.eval(xs)
    p64, v64 = net_fp64.forward(net.variables, xs)
    assert p.dtype == np.float32 and p.shape == (6, 121) and v.shape == (6,)
    assert np.abs(v - v64).max() < 1e-5 and np.abs(p - p64).max() < 1e-5
    assert np.allclose(p.sum(1), 1, atol=1e-5)
    # SURVEY §8c sanity values for the shipped checkpoint
    p0, v0 = net.eval(np.zeros((1, 3, 11, 11), np.float32))
    assert abs(float(v0[0]) - 0.10429) < 1e-5 and int(p0.argmax()) == 5 * 11 + 8 and abs(p0.max() - 0.017847) < 1e-6


def test_other_board_sizes_and_shape_checks():
    net = ResNet(15, device="cpu", seed=3)
    xs = _positions(15, 2)
    p, v = net.eval(xs)
    p64, v64 = net_fp64.forward(net.variables, xs)
    assert p.shape == (2, 225) and np.abs(v - v64).max() < 1e-5 and np.abs(p - p64).max() < 1e-5
    bad = dict(net.variables)
    bad["policy/fc/kernel"] = bad["policy/fc/kernel"][:, :100]
    with pytest.raises(ValueError):
        net.set_variables(bad)


@pytest.mark.skipif(not os.path.exists("/root/reference/ckpt/checkpoint"), reason="reference checkpoint not present")
def test_tensorbundle_reader_on_reference_checkpoint():
    prefix = tensorbundle.resolve_checkpoint("/root/reference/ckpt")
    assert prefix.endswith("alphaFive-6960")
    vs = tensorbundle.load_bundle(prefix)
    shapes = variable_shapes(11)
    assert set(vs) == set(shapes) and sum(v.size for v in vs.values()) == 754910
    with np.load(W) as z:
        for k in shapes:
            assert vs[k].shape == shapes[k] and (vs[k] == z[k]).all()
    with pytest.raises(FileNotFoundError):
        tensorbundle.resolve_checkpoint("/nonexistent/ckpt")


def test_crc32c_known_answer():
    assert tensorbundle.crc32c(b"123456789") == 0xE3069283
    assert tensorbundle.crc32c(b"") == 0


def test_tensorbundle_writer_roundtrip(tmp_path):
    net = ResNet(7, device="cpu", seed=5)
    prefix = str(tmp_path / "alphaFive-120")
    tensorbundle.save_bundle(prefix, net.variables)
    tensorbundle.write_checkpoint_state(str(tmp_path), "alphaFive-120")
    assert tensorbundle.resolve_checkpoint(str(tmp_path)) == prefix
    back = tensorbundle.load_bundle(prefix)
    assert set(back) == set(net.variables)
    for k, v in net.variables.items():
        assert back[k].dtype == np.float32 and back[k].shape == v.shape and (back[k] == v).all()
    net2 = ResNet(7, device="cpu", seed=9)
    net2.restore(str(tmp_path))
    x = _positions(7, 3)
    assert (net2.eval(x)[0] == net.eval(x)[0]).all()


@pytest.mark.skipif(not os.path.exists("/root/reference/ckpt/checkpoint"), reason="reference checkpoint not present")
def test_tensorbundle_writer_reproduces_the_reference_checkpoint_bytes(tmp_path):
    """Re-writing the tensors of ckpt/alphaFive-6960 must give byte-identical .data and .index files
    (tensor order, offsets, masked crc32c per tensor, SSTable block/restart/footer layout)."""
    src = "/root/reference/ckpt/alphaFive-6960"
    prefix = str(tmp_path / "alphaFive-6960")
    tensorbundle.save_bundle(prefix, tensorbundle.load_bundle(src))
    for ext in (".data-00000-of-00001", ".index"):
        assert open(prefix + ext, "rb").read() == open(src + ext, "rb").read(), ext
