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
    M, R = 6, 2 * 2 * 4 + 2 * 36 + 2
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


def _steady_worker(rank, world, port, q):
    """The steady-state shape of BASELINE configs[2]: every rank hands over ~160 episodes of ~26 plies at 11x11 per step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    C, K2, M = 121, 4, 512
    R = 2 * K2 + 2 * C + 2
    rng = np.random.RandomState(rank)
    bufs = []
    for step in range(3):                               # three distinct steps' buffers, built outside the measured calls
        eps = []
        for i in range(160):
            T = int(rng.randint(18, 35))
            eps.append(dict(game=i, seq=step, T=T, final_value=-1.0, keys=np.zeros((T, K2), np.uint64),
                            policies=np.full((T, C), 1.0 / C, np.float32), visits=np.full((T, C), 4, np.int32),
                            lasts=np.zeros(T, np.int32), actions=np.full(T, step, np.int32)))
        packed = afdist.pack_episodes(eps, M)
        full = torch.zeros(4 + 5 * M + 160 * 36 * R, dtype=torch.int32)      # like the engine's pack buffer: full capacity
        full[:len(packed)] = torch.from_numpy(packed)
        bufs.append((full, sum(e["T"] for e in eps)))
    g = afdist.EpisodeGather(world, rank, dev, M, R, games_per_rank=4096)
    n_eps = n_plies = 0
    allocs = []
    steps = 9
    for step in range(steps):
        got = g.collect(unpack=False)
        n_eps += sum(p.n for p in got)
        n_plies += sum(p.plies for p in got)
        g.post(bufs[step % 3][0])
        allocs.append(g.allocations)
    last = g.flush(unpack=False)
    n_eps += sum(p.n for p in last)
    n_plies += sum(p.plies for p in last)
    sample = last[-1].episodes()[:2] if rank == 0 else []
    wall_ms, host_ms = g.handoff_ms_per_step()
    q.put((rank, n_eps, n_plies, sum(b[1] for b in bufs) * (steps // 3), wall_ms, host_ms, allocs,
           [(e["game"], e["T"], int(e["actions"][0])) for e in sample]))
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_handoff_is_header_only_at_the_steady_state_shape():
    """at 8 ranks x 160 episodes x ~26 plies per step rank 0's share of the hand-off must not scale with the
    episodes: collect(unpack=False) + post() do no per-episode work (<= 2 ms of host work per step outside the collective
    calls, which are enqueue-only under RCCL), buffers are allocated once, the counts come from the packed headers, and the
    records are still there for whoever unpacks them."""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_steady_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    _, n_eps, n_plies, _, wall_ms, host_ms, allocs, sample = res[0]
    assert n_eps == 8 * 160 * 9 and n_plies == sum(res[r][3] for r in range(world))     # every rank's plies, header-counted
    assert all(res[r][1] == 0 for r in range(1, world))
    assert host_ms <= 2.0, f"rank 0 spends {host_ms:.2f} ms of host work per step in the hand-off"
    assert allocs[-1] == allocs[3], f"buffers re-allocated in the steady state: {allocs}"
    assert sample and sample[0][0] == 7 * 4096 and sample[0][2] == 2                   # last rank's records, game ids global
    print("rank 0 hand-off per step: %.2f ms wall (gloo: blocking collectives), %.3f ms host work" % (wall_ms, host_ms))


def test_pack_unpack_roundtrip_empty_and_ragged():
    assert afdist.unpack_episodes(afdist.pack_episodes([])) == []
    eps = [_episode(0, i, T) for i, T in enumerate([1, 9, 36])]
    back = afdist.unpack_episodes(afdist.pack_episodes(eps))
    for a, b in zip(eps, back):
        assert a["game"] == b["game"] and a["T"] == b["T"] and a["final_value"] == b["final_value"]
        for k in ("keys", "policies", "visits", "lasts", "actions"):
            assert (a[k] == b[k]).all()


def test_a_stale_view_of_the_pinned_ring_raises_instead_of_reading_overwritten_episodes():
    """collect(unpack=False) hands out views into a RING-deep pinned ring; a consumer that lags by RING collects must
    get an error, not another step's episodes; copy() detaches a view; posting without collecting is bounded."""
    import torch
    g = afdist.EpisodeGather(1, 0, torch.device("cpu"), 4, 2 * 4 + 2 * 36 + 2, games_per_rank=0)

    def step_buf(step):
        eps = [_episode(0, step, 3 + step)]
        return torch.from_numpy(afdist.pack_episodes(eps, 4))
    views, copies = [], []
    for step in range(6):
        got = g.collect(unpack=False)
        views += got
        copies += [p.copy() for p in got]
        g.post(step_buf(step))
    views += g.flush(unpack=False)
    assert len(views) == 6
    assert [int(c.episodes()[0]["T"]) for c in copies] == [3, 4, 5, 6]             # detached copies stay what they were (the pipeline runs two steps behind)
    assert views[-1].n == 1 and views[-2].n == 1 and views[-3].episodes()[0]["T"] == 6     # the newest RING views are live
    for v in views[:3]:                                                             # older ones: their slots were re-used
        with pytest.raises(RuntimeError, match="recycled"):
            v.episodes()
        with pytest.raises(RuntimeError, match="recycled"):
            v.n
    h = afdist.EpisodeGather(1, 0, torch.device("cpu"), 4, 2 * 4 + 2 * 36 + 2)
    for step in range(afdist.EpisodeGather.RING):
        h.post(step_buf(step))
    with pytest.raises(RuntimeError, match="without a collect"):
        h.post(step_buf(9))
