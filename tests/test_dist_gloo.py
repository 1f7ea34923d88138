"""N>1 path on CPU: world_size-2 and -8 gloo runs of the episode gather, weight broadcast and move counter (8 = the rank count of
BASELINE configs[2]: one process per GPU of a node)."""
import os
import socket

import pytest

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alphafive_amd import dist as afdist


def _episode(rank, i, T, C=36, kw2=4):
    rng = np.random.RandomState(rank * 100 + i)
    return dict(game=i, seq=i % 3, T=T, final_value=-1.0 if i % 2 else 0.0,
                keys=rng.randint(0, 2 ** 62, size=(T, kw2)).astype(np.uint64),
                policies=rng.rand(T, C).astype(np.float32), visits=rng.randint(0, 500, size=(T, C)).astype(np.int32),
                lasts=rng.randint(-1, C, size=T).astype(np.int32), actions=rng.randint(0, C, size=T).astype(np.int32))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    mine = [_episode(rank, i, 5 + 3 * i + rank) for i in range(2 + rank)]
    got = afdist.gather_episodes(mine, world, rank, dev, game_offset=rank * 1000)
    total = afdist.all_reduce_sum(sum(e["T"] for e in mine), dev)
    from alphafive_amd.network import ResNet
    net = ResNet(6, device="cpu", seed=rank)          # different weights per rank before the broadcast
    afdist.broadcast_weights(net, src=0)
    digest = float(sum(np.abs(v).sum() for v in net.variables.values()))
    # the pipelined form bench.py uses: collect -> pack -> post per step, flush at the end; payloads lag two steps
    M, R = 6, 2 * 4 + 2 * 36 + 2
    g = afdist.EpisodeGather(world, rank, dev, M, R, games_per_rank=1000)
    piped, lag = [], []
    for step in range(5):
        got_now = g.collect()
        lag.append(len(got_now))
        piped += got_now
        mine_s = [_episode(rank, 10 * step + i, 4 + i + step) for i in range((step + rank) % 4)]      # ragged, sometimes empty
        g.post(torch.from_numpy(afdist.pack_episodes(mine_s, M)))
    piped += g.flush()
    q.put((rank, [(e["game"], e["seq"], e["T"], e["final_value"], float(e["policies"].sum()),
                   int(e["visits"].sum()), int(e["keys"].sum() % 1000003)) for e in got], total, digest,
           [(e["game"], e["T"], int(e["visits"].sum())) for e in piped], lag, g.bytes_received))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gather_broadcast_allreduce(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = {}
    for _ in range(world):
        r = q.get(timeout=240)
        res[r[0]] = r
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    expect = []
    for rank in range(world):
        for i in range(2 + rank):
            e = _episode(rank, i, 5 + 3 * i + rank)
            expect.append((e["game"] + rank * 1000, e["seq"], e["T"], e["final_value"], float(e["policies"].sum()),
                           int(e["visits"].sum()), int(e["keys"].sum() % 1000003)))
    assert sorted(res[0][1]) == sorted(expect)          # rank 0 holds everyone's episodes, bit-exact
    assert all(res[r][1] == [] for r in range(1, world))
    assert all(res[r][2] == sum(t[2] for t in expect) for r in range(world))
    assert all(res[r][3] == res[0][3] for r in range(world))   # identical weights after the broadcast
    want = []
    for step in range(5):                               # pipelined gather: everything arrives at rank 0, in (step, rank) order
        for rank in range(world):
            for i in range((step + rank) % 4):
                e = _episode(rank, 10 * step + i, 4 + i + step)
                want.append((e["game"] + rank * 1000, e["T"], int(e["visits"].sum())))
    assert res[0][4] == want and all(res[r][4] == [] for r in range(1, world))
    assert res[0][5][:2] == [0, 0]                      # nothing can arrive before two steps have passed (sizes, then payload)
    assert res[0][6] > 0 and all(res[r][6] == 0 for r in range(1, world))   # only rank 0 receives payload bytes


def test_pack_unpack_roundtrip_empty_and_ragged():
    assert afdist.unpack_episodes(afdist.pack_episodes([])) == []
    eps = [_episode(0, i, T) for i, T in enumerate([1, 9, 36])]
    back = afdist.unpack_episodes(afdist.pack_episodes(eps))
    for a, b in zip(eps, back):
        assert a["game"] == b["game"] and a["T"] == b["T"] and a["final_value"] == b["final_value"]
        for k in ("keys", "policies", "visits", "lasts", "actions"):
            assert (a[k] == b[k]).all()
