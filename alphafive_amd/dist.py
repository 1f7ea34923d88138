"""Multi-GPU glue: games shard embarrassingly (one process per GPU, game id = rank*G + g, no
collective on the tick path).  The only exchanges are the ones SURVEY §8e lists:

  broadcast_weights   one broadcast of the 3 MB fp32 weight set from rank 0 (per weight update)
  gather_packed       finished episodes -> rank 0: the engine packs them on the device into one
                      int32 buffer (af_engine_pack_episodes); sizes all-gather, then one padded
                      all-gather of the used prefixes; rank 0 keeps the result — the device-side
                      replacement of main.py:51,94's multiprocessing.Queue hand-off
  gather_episodes     the same for callers that hold episode dicts
  all_reduce_sum      the moves counter for the metric

Backend: torch.distributed "nccl" (= RCCL over xGMI) on GPUs; "gloo" in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import packed_used_ints, unpack_episodes as _unpack_packed


def pack_episodes(eps):
    """Host-side builder of the engine's packed hand-off layout (include/af_engine.h af_engine_pack_episodes) with
    max_episodes = len(eps): list of raw episode dicts -> int32 numpy buffer.  (The device produces this layout itself;
    this builder serves callers that hold episode dicts, e.g. the CPU tests.)"""
    n = len(eps)
    K = int(eps[0]["keys"].shape[1]) if n else 4
    Cc = int(eps[0]["policies"].shape[1]) if n else 0
    R = 2 * K + 2 * Cc + 2
    plies = sum(int(e["T"]) for e in eps)
    buf = np.zeros(4 + 5 * n + plies * R, np.int32)
    buf[:4] = (n, plies, K, Cc)
    p0 = 0
    for i, e in enumerate(eps):
        T = int(e["T"])
        buf[4 + 4 * i:8 + 4 * i] = (e["game"], e["seq"], T, p0)
        buf[4 + 4 * n + i] = np.float32(e["final_value"]).view(np.int32)
        rec = buf[4 + 5 * n + p0 * R:4 + 5 * n + (p0 + T) * R].reshape(T, R)
        rec[:, :2 * K] = np.ascontiguousarray(e["keys"], np.uint64).view(np.int32).reshape(T, 2 * K)
        rec[:, 2 * K:2 * K + Cc] = np.ascontiguousarray(e["policies"], np.float32).view(np.int32)
        rec[:, 2 * K + Cc:2 * K + 2 * Cc] = e["visits"]
        rec[:, 2 * K + 2 * Cc] = e["lasts"]
        rec[:, 2 * K + 2 * Cc + 1] = e["actions"]
        p0 += T
    return buf


def unpack_episodes(buf, max_eps=None):
    buf = np.ascontiguousarray(buf, np.int32)
    return _unpack_packed(buf, int(buf[0]) if max_eps is None else max_eps)


def gather_packed(buf, max_eps, world, rank, device, games_per_rank=0):
    """buf: this rank's packed hand-off buffer (torch int32, device or host; layout of af_engine_pack_episodes with
    the given max_eps).  Rank 0 gets every rank's episodes (game ids made global: + r * games_per_rank), the other
    ranks get [].  One sizes all-gather and one padded all-gather of the used prefixes (RCCL over xGMI on GPUs:
    the packed device buffer goes to the collective as it is, no host round trip on the sending side)."""
    hdr = buf[:4].cpu().numpy()                       # waits for the pack kernels of this buffer only
    used = packed_used_ints(hdr, max_eps)

    def local(b, r):
        eps = _unpack_packed(b, max_eps)
        if games_per_rank:
            for e in eps:
                e["game"] += r * games_per_rank
        return eps

    if world == 1:
        return local(buf[:used].cpu().numpy(), 0)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([used], dtype=torch.int64, device=device))
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes)
    payload = buf[:width].to(device)
    if payload.numel() < width:                       # a rank whose buffer is smaller than the widest payload
        payload = torch.cat([payload, torch.zeros(width - payload.numel(), dtype=torch.int32, device=device)])
    bufs = [torch.empty(width, dtype=torch.int32, device=device) for _ in range(world)]
    dist.all_gather(bufs, payload.contiguous())
    if rank != 0:
        return []
    out = []
    for r in range(world):
        out += local(bufs[r][:sizes[r]].cpu().numpy(), r)
    return out


def gather_episodes(eps, world, rank, device, game_offset=None):
    """Episode-dict form of gather_packed (callers that already hold raw dicts): every rank passes its finished
    episodes; rank 0 gets everyone's (game ids made global with game_offset = rank * games_per_rank when given)."""
    if game_offset:
        eps = [dict(e, game=e["game"] + game_offset) for e in eps]
    if world == 1:
        return eps
    # every rank may hold a different number of episodes: normalise to a common max_eps first
    n = torch.tensor([len(eps)], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    max_eps = int(n.item())
    own = pack_episodes(eps)
    # re-home the ply region to the common max_eps
    k = len(eps)
    buf = np.zeros(4 + 5 * max_eps + (own.size - 4 - 5 * k), np.int32)
    buf[:4] = own[:4]
    buf[4:4 + 4 * k] = own[4:4 + 4 * k]
    buf[4 + 4 * max_eps:4 + 4 * max_eps + k] = own[4 + 4 * k:4 + 5 * k]
    buf[4 + 5 * max_eps:] = own[4 + 5 * k:]
    if k == 0:
        buf[2], buf[3] = 0, 0
    return gather_packed(torch.from_numpy(buf), max_eps, world, rank, device)


def broadcast_weights(net, src=0):
    """Replicate rank `src`'s network variables on every rank (ncclBroadcast per tensor)."""
    dev = net.device if (net.device.type == "cuda" and dist.get_backend() == "nccl") else torch.device("cpu")
    new = {}
    for name in sorted(net.variables):
        t = torch.from_numpy(net.variables[name]).to(dev)
        dist.broadcast(t, src=src)
        new[name] = t.cpu().numpy()
    net.set_variables(new)


def all_reduce_sum(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
