"""Worker of tests/test_gpu_multirank.py::test_rccl_branch_runs_on_the_device: a world-size-1 "nccl" (= RCCL) process group
on GPU 0, through which every collective of the multi-GPU path runs once on device tensors — weight broadcast, move-counter
all-reduce, and the pipelined episode gather (sizes all-gather + gather to rank 0) fed by the engine's pack kernels — so
that the RCCL branch of alphafive_amd.dist / bench.py has executed before an 8-GPU node ever sees it.  The gathered
episodes must equal what a twin engine hands out through the pinned-host path."""
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
    for k in ("policies", "visits", "actions", "lasts"):
        h = zlib.crc32(np.ascontiguousarray(e[k]).tobytes(), h)
    return [int(e["game"]), int(e["seq"]), int(e["T"]), float(e["final_value"]), int(h)]


def main():
    out = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from alphafive_amd import dist as afdist
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    net = ResNet(6, device=dev, seed=4)
    before = {k: v.copy() for k, v in net.variables.items()}
    afdist.broadcast_weights(net, src=0)
    same = all((net.variables[k] == before[k]).all() for k in before)
    moves = afdist.all_reduce_sum(1234, dev)
    G, cap = 64, 128
    cfg = make_cfg(board_size=6, goal=4, simulation_per_step=30, upper_simulation_per_step=40)
    mk = lambda: SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, 77, 4096), device=0, seed=8, first_game_id=0)
    a, b = mk(), mk()
    gat = afdist.EpisodeGather(1, 0, dev, cap, 2 * a.engine.KW2 + 2 * 36 + 2, games_per_rank=G)
    assert gat.collective
    got, ref = [], []
    for it in range(10):
        for sp in (a, b):
            if sp is a and it >= 2:             # the HIP-graph loop with the RCCL process group (and its watchdog thread) alive:
                for _ in range(15):             # bench.py's N > 1 situation — capture and replays must work next to the collectives
                    sp.run_ticks_graph(20)
            else:
                sp.run_ticks(301 if (sp is b and it == 2) else 300)      # (a's first replay is preceded by one eager warm-up tick)
            sp.check()
        got += gat.collect()
        gat.post(a.post_episodes_device(cap))
        ref += b.pop_raw(cap)
    got += gat.flush()
    with open(out, "w") as f:
        json.dump({"weights_same": bool(same), "moves": moves, "got": [digest(e) for e in got], "ref": [digest(e) for e in ref],
                   "backend": dist.get_backend(), "bytes_received": gat.bytes_received}, f)
    a.close()
    b.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
