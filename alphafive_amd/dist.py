"""Multi-GPU glue: games shard embarrassingly (one process per GPU, game id = rank*G + g, no
collective on the tick path).  The only exchanges are the ones SURVEY §8e lists:

  broadcast_weights   one broadcast of the 3 MB fp32 weight set from rank 0 (per weight update)
  gather_episodes     finished episodes -> rank 0 (variable length: sizes all-gather, then one
                      padded all-gather; rank 0 keeps the result) — the device-side
                      replacement of main.py:51,94's multiprocessing.Queue hand-off
  all_reduce_sum      the moves counter for the metric

Backend: torch.distributed "nccl" (= RCCL over xGMI) on GPUs; "gloo" in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

_HDR = 4   # game, seq, T, final_value bits


def pack_episodes(eps):
    """list of raw episode dicts (engine.pop_episodes_raw) -> one int32 numpy buffer."""
    parts = [np.asarray([len(eps)], np.int32)]
    for e in eps:
        T = int(e["T"])
        hdr = np.asarray([e["game"], e["seq"], T, np.float32(e["final_value"]).view(np.int32)], np.int32)
        parts += [hdr, np.asarray([e["keys"].shape[1], e["policies"].shape[1]], np.int32),
                  np.ascontiguousarray(e["keys"], np.uint64).view(np.int32).reshape(-1),
                  np.ascontiguousarray(e["policies"], np.float32).view(np.int32).reshape(-1),
                  np.ascontiguousarray(e["visits"], np.int32).reshape(-1),
                  np.ascontiguousarray(e["lasts"], np.int32), np.ascontiguousarray(e["actions"], np.int32)]
    return np.concatenate(parts)


def unpack_episodes(buf):
    buf = np.ascontiguousarray(buf, np.int32)
    n, at, out = int(buf[0]), 1, []
    for _ in range(n):
        game, seq, T, fvb = (int(x) for x in buf[at:at + _HDR])
        at += _HDR
        kw2, C = int(buf[at]), int(buf[at + 1])
        at += 2
        keys = buf[at:at + 2 * T * kw2].view(np.uint64).reshape(T, kw2).copy()
        at += 2 * T * kw2
        pol = buf[at:at + T * C].view(np.float32).reshape(T, C).copy()
        at += T * C
        vis = buf[at:at + T * C].reshape(T, C).copy()
        at += T * C
        lasts = buf[at:at + T].copy()
        at += T
        actions = buf[at:at + T].copy()
        at += T
        out.append(dict(game=game, seq=seq, T=T, final_value=float(np.int32(fvb).view(np.float32)), keys=keys,
                        policies=pol, visits=vis, lasts=lasts, actions=actions))
    return out


def gather_episodes(eps, world, rank, device, game_offset=None):
    """Every rank passes its finished episodes; rank 0 gets everyone's (game ids made global with
    game_offset = rank * games_per_rank when given), other ranks get []."""
    if game_offset:
        eps = [dict(e, game=e["game"] + game_offset) for e in eps]
    if world == 1:
        return eps
    payload = torch.from_numpy(pack_episodes(eps)).to(device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([payload.numel()], dtype=torch.int64, device=device))
    sizes = [int(s.item()) for s in sizes]
    # one padded all-gather (the collective every backend has; RCCL point-to-point would open a communicator per
    # rank pair on first use): <= a few MB per rank and step, off the tick path
    width = max(sizes)
    padded = torch.zeros(width, dtype=torch.int32, device=device)
    padded[:payload.numel()] = payload
    bufs = [torch.empty(width, dtype=torch.int32, device=device) for _ in range(world)]
    dist.all_gather(bufs, padded)
    if rank != 0:
        return []
    out = list(eps)
    for r in range(1, world):
        out += unpack_episodes(bufs[r][:sizes[r]].cpu().numpy())
    return out


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
