"""Worker of tests/test_gpu_multirank.py: one rank of an N-rank self-play run that shares GPU 0 (gloo rendezvous).
Every rank owns G games (game id = rank*G + g), ticks a fixed number of times, and hands its finished episodes to
rank 0 through the packed device buffer + alphafive_amd.dist.EpisodeGather; rank 0 writes their digest."""
import json
import os
import sys
import zlib

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import pseudonet  # noqa: E402
from conftest import make_cfg  # noqa: E402


def digest(e):
    h = zlib.crc32(np.ascontiguousarray(e["keys"]).tobytes())
    h = zlib.crc32(np.ascontiguousarray(e["policies"]).tobytes(), h)
    h = zlib.crc32(np.ascontiguousarray(e["visits"]).tobytes(), h)
    h = zlib.crc32(np.ascontiguousarray(e["actions"]).tobytes(), h)
    return [int(e["game"]), int(e["seq"]), int(e["T"]), float(e["final_value"]), int(h)]


def main():
    out, G, ticks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    from alphafive_amd import dist as afdist
    from alphafive_amd.engine import SelfPlayEngine
    cfg = make_cfg(board_size=7, goal=4, simulation_per_step=40, upper_simulation_per_step=60)
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, 555, 8192), device=0, seed=2025, first_game_id=rank * G)
    eps, cap = [], 2 * G                      # a game holds at most two finished episodes: nothing is ever held back
    gat = afdist.EpisodeGather(world, rank, torch.device("cpu"), cap, 2 * sp.engine.KW2 + 2 * 49 + 2, games_per_rank=G)
    for _ in range(ticks // 500):             # the pipelined hand-off of bench.py: collect -> pack -> post
        sp.run_ticks(500)
        sp.check()
        eps += gat.collect()
        gat.post(sp.post_episodes_device(cap))
    eps += gat.flush()
    moves = afdist.all_reduce_sum(sp.progress()[0], torch.device("cpu"))
    if rank == 0:
        with open(out, "w") as f:
            json.dump({"episodes": sorted(digest(e) for e in eps), "moves": moves}, f)
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
