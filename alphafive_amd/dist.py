"""Multi-GPU glue: games shard embarrassingly (one process per GPU, game id = rank*G + g, no
collective on the tick path).  The only exchanges are the ones SURVEY §8e lists:

  broadcast_weights   one broadcast of the 3 MB fp32 weight set from rank 0 (per weight update)
  EpisodeGather       finished episodes -> rank 0 ONLY (main.py:51,94's Queue hand-off on the device):
                      the engine packs them into one int32 device buffer (af_engine_pack_episodes); the used
                      sizes travel as a tiny all-gather issued with the pack and are read one step later, then
                      one gather (grouped send/recv towards rank 0) of the used prefixes; rank 0 copies the
                      payload to pinned host memory asynchronously and unpacks it another step later.  Nothing
                      on the step path waits for the device (no .item()/.cpu() of fresh data).
  gather_packed       the blocking form of the same gather (tests, one-off callers)
  gather_episodes     the same for callers that hold episode dicts
  all_reduce_sum      the moves counter for the metric

Backend: torch.distributed "nccl" (= RCCL over xGMI) on GPUs; "gloo" in the CPU tests.
"""
import time as _time

import numpy as np
import torch
import torch.distributed as dist

from .engine import packed_used_ints, unpack_episodes as _unpack_packed


def pack_episodes(eps, max_eps=None):
    """Host-side builder of the engine's packed hand-off layout (include/af_engine.h af_engine_pack_episodes) with
    max_episodes = max_eps (default len(eps)): list of raw episode dicts -> int32 numpy buffer.  (The device produces this
    layout itself; this builder serves callers that hold episode dicts, e.g. the CPU tests.)"""
    n = len(eps)
    M = n if max_eps is None else int(max_eps)
    assert n <= M
    K = int(eps[0]["keys"].shape[1]) if n else 4
    Cc = int(eps[0]["policies"].shape[1]) if n else 0
    R = 2 * K + 2 * Cc + 2
    plies = sum(int(e["T"]) for e in eps)
    buf = np.zeros(4 + 5 * M + plies * R, np.int32)
    buf[:4] = (n, plies, K, Cc)
    p0 = 0
    for i, e in enumerate(eps):
        T = int(e["T"])
        buf[4 + 4 * i:8 + 4 * i] = (e["game"], e["seq"], T, p0)
        buf[4 + 4 * M + i] = np.float32(e["final_value"]).view(np.int32)
        rec = buf[4 + 5 * M + p0 * R:4 + 5 * M + (p0 + T) * R].reshape(T, R)
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


class PackedEpisodes(object):
    """One rank's finished episodes as rank 0 received them: a numpy int32 VIEW of the packed hand-off layout
    (af_engine_pack_episodes) in one of EpisodeGather's pinned host buffers.  Reading `n` / `plies` touches the 4-int header
    only; episodes() unpacks into raw episode dicts (game ids made global) — call it when the records are needed (a trainer's
    push), from any thread, before the buffer is recycled: the view stays valid through the collect() after the one that
    returned it (the second one after it overwrites the buffer) and RAISES when read later than that (the ring slot carries a
    generation counter); copy() detaches it."""
    __slots__ = ("buf", "max_eps", "rank", "games_per_rank", "_gen", "_slot")

    def __init__(self, buf, max_eps, rank, games_per_rank, gen=None, slot=None):
        self.buf, self.max_eps, self.rank, self.games_per_rank = buf, max_eps, rank, games_per_rank
        self._gen, self._slot = gen, slot        # generation of the ring slot this view lives in (None: detached copy)

    def _live(self):
        """A view handed to a slower thread must not be read after its ring slot was given to a later step: the slot's
        generation counter (bumped when the slot is re-used) must still be the one this view was made with."""
        if self._slot is not None and self._slot[0] != self._gen:
            raise RuntimeError("PackedEpisodes view read after its pinned ring slot was recycled (%d collects later); "
                               "copy() it before handing it to a thread that may lag" % (self._slot[0] - self._gen))

    @property
    def n(self):
        self._live()
        return int(self.buf[0])

    @property
    def plies(self):
        self._live()
        return int(self.buf[1])

    def lengths(self):
        """T of every episode (header only)."""
        return self.buf[4:4 + 4 * self.n].reshape(-1, 4)[:, 2]

    def copy(self):
        self._live()
        out = PackedEpisodes(self.buf.copy(), self.max_eps, self.rank, self.games_per_rank)
        self._live()                             # (recycled while copying: the copy is torn)
        return out

    def episodes(self):
        self._live()
        eps = _unpack_packed(self.buf, self.max_eps)
        self._live()                             # (recycled while unpacking: the records are torn)
        if self.games_per_rank:
            for e in eps:
                e["game"] += self.rank * self.games_per_rank
        return eps


def _run_collectives(world):
    """Collectives run whenever a process group is up — also at world size 1 (the RCCL smoke test on a 1-GPU box)."""
    return dist.is_available() and dist.is_initialized() and (world > 1 or dist.get_world_size() == 1)


class EpisodeGather(object):
    """Pipelined gather of packed episode buffers to rank 0.  Per step, in this order on every rank:

        got = g.collect(unpack=False)          # advances the buffers posted earlier; rank 0 gets finished payloads
        buf = sp.post_episodes_device(cap)     # this step's pack kernels (they overwrite the engine's pack buffer)
        g.post(buf)                            # sizes all-gather on the device, read back asynchronously
        ...
        got += g.flush(unpack=False)           # at the end: drains everything (blocking)

    collect() must come before the next pack: the gather it issues reads the previously posted buffer (the collective is
    ordered with the current stream, so the pack that follows cannot overtake it).
    Rank 0's share of a step is O(world) small calls and no per-episode work: sizes (one event that retired a step ago),
    one gather into a receive buffer that is allocated once, `world` asynchronous device-to-pinned copies of the used row prefixes
    into a ring of pinned buffers that is allocated once, and — with unpack=False — PackedEpisodes views whose headers carry the counts
    (bench.py reads n / plies only; a trainer unpacks where it needs the records, see PackedEpisodes).  unpack=True
    (default, tests and one-off callers) returns raw episode dicts as before.
    max_eps / rec_ints describe the pack layout (af_engine_pack_episodes: used = 4 + 5*max_eps + plies*rec_ints).
    `device`: where the collectives run — the GPU for "nccl" (= RCCL over xGMI), cpu for "gloo".
    stats: host seconds spent in collect() / post() on this rank, split into time inside torch.distributed calls
    (enqueue-only under RCCL, blocking under gloo) and everything else (`host_s`, thread CPU time)."""

    RING = 3                                  # pinned payload buffers: delivered views outlive two more collects

    def __init__(self, world, rank, device, max_eps, rec_ints, games_per_rank=0):
        self.world, self.rank, self.device = world, rank, torch.device(device)
        self.max_eps, self.rec_ints, self.games_per_rank = max_eps, rec_ints, games_per_rank
        self.collective = world > 1 or _run_collectives(world)
        self._tickets = []
        self.bytes_received = 0           # rank 0: payload bytes that crossed the fabric towards it
        self._pin = torch.cuda.is_available()
        self._sizes_host = [torch.zeros(world, dtype=torch.int64, pin_memory=self._pin) for _ in range(self.RING + 1)]
        self._sizes_dev = [None] * (self.RING + 1)
        self._turn = 0
        self._recv = None                 # rank 0: [world, cap] int32 on `device`
        self._host = [None] * self.RING   # rank 0: pinned [world, cap] int32
        self._host_gen = [[0] for _ in range(self.RING)]      # per slot: bumped when the slot is handed to a new step
        self._host_turn = 0
        self.allocations = 0              # buffer (re)allocations so far: constant once the run has seen its widest step
        self.stats = {"steps": 0, "wall_s": 0.0, "host_s": 0.0, "comm_s": 0.0}

    # ---- buffers: grown geometrically, i.e. allocated once per size class ----
    def _ensure(self, width):
        """-> (receive buffer [world, >= width] on `device`, pinned host buffer of this turn).  On a cpu `device` (gloo) the
        collective receives straight into the host ring: there is no second copy."""
        k = self._host_turn
        on_host = self.device.type != "cuda"
        have = self._host[k].shape[1] if self._host[k] is not None else 0
        want = max(have, self._recv.shape[1] if self._recv is not None else 0)
        if want < width:
            want = int(width * 1.5) + 1024
        if not on_host and (self._recv is None or self._recv.shape[1] < want):
            self._recv = torch.empty((self.world, want), dtype=torch.int32, device=self.device)
            self.allocations += 1
        if have < want:
            self._host[k] = torch.empty((self.world, want), dtype=torch.int32, pin_memory=self._pin)
            self.allocations += 1
        return (self._host[k] if on_host else self._recv), self._host[k]

    def _comm(self, fn, *a, **kw):
        t0 = _time.thread_time()
        out = fn(*a, **kw)
        self.stats["comm_s"] += _time.thread_time() - t0
        return out

    def post(self, buf):
        """buf: the packed int32 buffer of this step (device tensor).  Issues the sizes exchange; returns at once."""
        w0, c0 = _time.perf_counter(), _time.thread_time()
        if len(self._tickets) >= self.RING:      # the sizes ring has RING + 1 entries and a ticket owns one until it is gathered
            raise RuntimeError("EpisodeGather.post: %d buffers posted without a collect() in between (at most %d)"
                               % (len(self._tickets) + 1, self.RING))
        k = self._turn = (self._turn + 1) % len(self._sizes_host)
        used = (buf[1:2].to(torch.int64) * self.rec_ints + (4 + 5 * self.max_eps))          # stays on the device
        if self.collective:
            if self._sizes_dev[k] is None:
                self._sizes_dev[k] = torch.empty(self.world, dtype=torch.int64, device=self.device)
            allsz = self._sizes_dev[k]
            self._comm(dist.all_gather_into_tensor, allsz, used.to(self.device))
        else:
            allsz = used
        host = self._sizes_host[k]
        if allsz.device.type == "cuda":
            host.copy_(allsz, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(allsz.device))
        else:
            host.copy_(allsz)
            ev = None
        self._tickets.append({"stage": 0, "buf": buf, "sizes": host, "ev": ev})
        self.stats["wall_s"] += _time.perf_counter() - w0
        self.stats["host_s"] += _time.thread_time() - c0

    def _gather(self, t):
        if t["ev"] is not None:
            t["ev"].synchronize()                 # recorded a whole step ago
        sizes = [int(x) for x in t["sizes"].tolist()]
        width = max(sizes)
        payload = t["buf"][:width]
        if payload.device != self.device:
            payload = payload.to(self.device)
        if payload.numel() < width:               # a caller-built buffer smaller than the widest payload (gather_episodes)
            payload = torch.cat([payload, torch.zeros(width - payload.numel(), dtype=torch.int32, device=self.device)])
        t.update(stage=1, buf=None, sizes=sizes, host=None)
        if self.rank != 0:
            if self.collective:
                self._comm(dist.gather, payload.contiguous(), None, dst=0)     # grouped send towards rank 0: no other rank receives
            t["stage"] = 2
            return
        recv, host = self._ensure(width)
        slot = self._host_gen[self._host_turn]
        slot[0] += 1                              # views of the step that used this slot RING collects ago are stale from here on
        t["slot"], t["gen"] = slot, slot[0]
        self._host_turn = (self._host_turn + 1) % self.RING
        if self.collective:
            self._comm(dist.gather, payload.contiguous(), [recv[r, :width] for r in range(self.world)], dst=0)
        else:
            recv[0, :width].copy_(payload)
        self.bytes_received += sum(sizes[1:]) * 4
        ev = None
        if recv.device.type == "cuda":
            for r in range(self.world):               # contiguous row prefixes: plain asynchronous device-to-pinned copies (a strided
                host[r, :sizes[r]].copy_(recv[r, :sizes[r]], non_blocking=True)      # 2-D copy_ would stage through a temporary)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(recv.device))
        t["host"], t["ev"] = host, ev

    def _deliver(self, t, unpack):
        if t["host"] is None:
            return []
        if t["ev"] is not None:
            t["ev"].synchronize()
        h = t["host"].numpy()
        out = [PackedEpisodes(h[r, :t["sizes"][r]], self.max_eps, r, self.games_per_rank, t["gen"], t["slot"]) for r in range(self.world)]
        if not unpack:
            return out
        eps = []
        for pk in out:
            eps += pk.episodes()
        return eps

    def collect(self, drain=False, unpack=True):
        """Advance every posted buffer by one stage (all of them to the end with drain=True); -> rank 0: what has arrived
        (raw episode dicts, or one PackedEpisodes per rank and step with unpack=False), other ranks: []."""
        w0, c0 = _time.perf_counter(), _time.thread_time()
        out, keep = [], []
        for t in self._tickets:
            if t["stage"] == 0:
                self._gather(t)
                if not drain:
                    keep.append(t)
                    continue
            out += self._deliver(t, unpack)
        self._tickets = keep
        self.stats["steps"] += 1
        self.stats["wall_s"] += _time.perf_counter() - w0
        self.stats["host_s"] += _time.thread_time() - c0
        return out

    def flush(self, unpack=True):
        return self.collect(drain=True, unpack=unpack)

    def handoff_ms_per_step(self):
        """-> (wall ms, host-work ms outside torch.distributed calls) this rank spent in collect() + post() per step."""
        n = max(1, self.stats["steps"])
        return 1e3 * self.stats["wall_s"] / n, 1e3 * (self.stats["host_s"] - self.stats["comm_s"]) / n


def gather_packed(buf, max_eps, world, rank, device, games_per_rank=0):
    """Blocking form: buf = this rank's packed hand-off buffer (torch int32, device or host; layout of
    af_engine_pack_episodes with the given max_eps).  Rank 0 gets every rank's episodes (game ids made global:
    + r * games_per_rank), the other ranks get [].  One sizes all-gather, one gather of the used prefixes to rank 0."""
    hdr = buf[:4].cpu().numpy()                       # waits for the pack kernels of this buffer only
    rec = 2 * int(hdr[2]) + 2 * int(hdr[3]) + 2
    g = EpisodeGather(world, rank, device, max_eps, rec, games_per_rank)
    g.post(buf)
    return g.flush()


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
    buf = pack_episodes(eps, max_eps)
    return gather_packed(torch.from_numpy(buf), max_eps, world, rank, device)


def broadcast_weights(net, src=0):
    """Replicate rank `src`'s network variables on every rank: ONE broadcast of the flattened weight set (3 MB at 11x11;
    main.py:73-75 hands new weights to its workers through a checkpoint file)."""
    dev = net.device if (net.device.type == "cuda" and dist.get_backend() == "nccl") else torch.device("cpu")
    names = sorted(net.variables)
    flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(net.variables[k], np.float32).reshape(-1) for k in names])).to(dev)
    dist.broadcast(flat, src=src)
    flat = flat.cpu().numpy()
    new, at = {}, 0
    for k in names:
        n = net.variables[k].size
        new[k] = flat[at:at + n].reshape(net.variables[k].shape).copy()
        at += n
    net.set_variables(new)


def all_reduce_sum(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
