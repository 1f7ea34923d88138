"""SURVEY §8e parity check in the only form a 1-GPU box allows: two ranks share GPU 0 (gloo rendezvous), each owns G
games with game id = rank*G + g and hands its finished episodes to rank 0 through the device pack + all-gather path;
the multiset rank 0 ends up with must equal what ONE engine of 2G games produces — i.e. an N-rank run is N independent
shards of the same games.  Also: `bench.py --gpus 2` (the driver's launch line, AF_BENCH_SHARE_GPU=1) prints one
consistent JSON line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, make_cfg

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(args, extra_env=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_produce_the_episodes_of_one_big_engine(tmp_path):
    import pseudonet
    import zlib
    from alphafive_amd.engine import SelfPlayEngine
    G, ticks = 128, 6000
    out = str(tmp_path / "rank0.json")
    r = _launch([os.path.join(REPO, "tests", "multirank_worker.py"), out, str(G), str(ticks)])
    assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as f:
        two = json.load(f)
    cfg = make_cfg(board_size=7, goal=4, simulation_per_step=40, upper_simulation_per_step=60)
    sp = SelfPlayEngine(cfg, 2 * G, lambda x: pseudonet.pseudonet_torch(x, 555, 8192), device=0, seed=2025)
    eps = []
    for _ in range(ticks // 500):
        sp.run_ticks(500)
        sp.check()
        while True:
            raws = sp.pop_raw(cap=2 * G)
            eps += raws
            if len(raws) < 2 * G:
                break
    moves = sp.progress()[0]
    sp.close()

    def digest(e):
        h = zlib.crc32(np.ascontiguousarray(e["keys"]).tobytes())
        h = zlib.crc32(np.ascontiguousarray(e["policies"]).tobytes(), h)
        h = zlib.crc32(np.ascontiguousarray(e["visits"]).tobytes(), h)
        h = zlib.crc32(np.ascontiguousarray(e["actions"]).tobytes(), h)
        return [int(e["game"]), int(e["seq"]), int(e["T"]), float(e["final_value"]), int(h)]

    one = sorted(digest(e) for e in eps)
    assert len(one) > 2 * G                                  # several episodes per game
    assert one == two["episodes"]                            # the same multiset of episodes, bit for bit (crc of every record)
    assert {d[0] for d in two["episodes"]} == set(range(2 * G))      # both shards are there
    assert two["moves"] == moves                              # the all-reduced move counter


def test_bench_two_ranks_prints_one_consistent_line():
    r = _launch([os.path.join(REPO, "bench.py"), "--gpus", "2", "--games", "96", "--board", "6", "--sims", "30", "--upper", "40",
                 "--steps", "36", "--warmup", "4", "--no-cpu-baseline"], extra_env={"AF_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 36 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["value"] is not None and d["value"] > 0
    # whole-job aggregate: both ranks' plies over the max-over-ranks time
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * 96) < 0.25 * 2 * 96
    assert d["config"]["episodes_gathered"] >= 2 * 96 * 0.5  # rank 0 received episodes of both shards
    _check_self_verifying_keys(d, 2, "gloo", "torch.distributed.run / external")


def _check_self_verifying_keys(d, world, backend, launcher):
    c = d["config"]
    assert c["ranks_seen"] == world and c["backend"] == backend and c["launcher"] == launcher
    assert len(c["devices"]) == world and len(c["per_rank"]) == world and [r["rank"] for r in c["per_rank"]] == list(range(world))
    assert all(r["episodes_finished"] > 0 and r["plies"] > 0 for r in c["per_rank"])
    # rank 0 holds every episode any rank finished (after the untimed drain), counted from the headers it received
    assert c["gathered_equals_finished"] and c["episodes_gathered_total"] == c["episodes_finished_total_all_ranks"] > 0
    assert sum(r["episodes_finished"] for r in c["per_rank"]) == c["episodes_finished_in_timed_region"]


def test_plain_python_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` started the way the driver starts `--gpus 1` (no torch.distributed.run
    around it, no RANK / WORLD_SIZE in the environment) launches its two ranks itself and prints ONE contract line that
    carries what the process group really was."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", AF_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--games", "96", "--board", "6", "--sims", "30",
                        "--upper", "40", "--steps", "36", "--warmup", "4", "--no-cpu-baseline"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 36 and d["warmup"] == 4 and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * 96) < 0.25 * 2 * 96
    _check_self_verifying_keys(d, 2, "gloo", "bench.py spawn_ranks")


def test_rccl_branch_runs_on_the_device(tmp_path):
    """No multi-GPU box is available to the build, but the "nccl" (= RCCL) code path can still execute: a world-size-1 group
    on GPU 0 runs the weight broadcast, the move-counter all-reduce and the pipelined episode gather on device tensors."""
    out = str(tmp_path / "nccl.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "nccl_smoke_worker.py"), out], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    with open(out) as f:
        d = json.load(f)
    assert d["backend"] == "nccl" and d["weights_same"] and d["moves"] == 1234.0
    assert len(d["ref"]) > 64 and d["got"] == d["ref"]            # same episodes, same order, bit for bit (crc of every record)


def test_bench_eight_ranks_dry_run_on_one_gpu():
    """BASELINE configs[2]'s shape (8 ranks, games sharded by id, finished episodes gathered to rank 0) has
    never met 8 devices; what can be de-risked on one is everything but the fabric: `python bench.py --gpus 8` spawning its own
    eight ranks (sharing GPU 0, gloo), every rank seen by the group, rank 0 holding exactly the episodes each rank's device counter
    says it finished — attributed to the right source rank, global game ids up to 8 G - 1 — and the pinned ring allocated once
    (EpisodeGather raises on an overrun or a stale view, so rc 0 = it never happened)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", AF_BENCH_SHARE_GPU="1")
    G = 256
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--games", str(G), "--board", "7", "--sims", "40",
                        "--upper", "60", "--steps", "44", "--warmup", "4", "--no-cpu-baseline"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 8 * G) < 0.25 * 8 * G          # whole-job plies over the max-over-ranks time
    _check_self_verifying_keys(d, 8, "gloo", "bench.py spawn_ranks")
    c = d["config"]
    assert c["episodes_gathered_by_source_rank"] == c["episodes_finished_by_rank_total"]   # per source rank, not just in total
    assert min(c["episodes_gathered_by_source_rank"]) > 0
    assert 7 * G <= c["max_global_game_id_gathered"] < 8 * G
    assert c["gather_ring_allocations"] <= 24                                         # a few size classes x 3 ring slots, not one per step (48 steps)
