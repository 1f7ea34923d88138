"""Every game, not a sample: configs[1] at full size (4096 games, 11x11, 500 sims/move, cap 642, pseudo-net) run until every game
has finished an episode, then EVERY first episode (and every second one that exists) is replayed by the C oracle on the host cores
and compared bit for bit (tests/test_gpu_fullsize.py does this for 8 sampled games inside the test budget).
Env: BOARD / SIMS / UPPER (11 / 500 / 642; 15 / 800 / 942 = configs[3]), G (4096), MEMO (0 / 1: the evaluation memo on), PROCS (host processes, default = usable cores)."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
SALT, PEAK, SEED = 777, 8192, 42
BOARD, SIMS, UPPER = int(os.environ.get("BOARD", 11)), int(os.environ.get("SIMS", 500)), int(os.environ.get("UPPER", 642))


def check_game(item):
    g, eps = item
    import oracle
    from conftest import make_cfg
    from test_gpu_fullsize import _assert_episode_equals_oracle
    cfg = make_cfg(board_size=BOARD, simulation_per_step=SIMS, upper_simulation_per_step=UPPER)
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g, pseudo_salt=SALT, pseudo_peak=PEAK)
    n = 0
    try:
        for raw in eps[:2]:
            _assert_episode_equals_oracle(raw, orc, BOARD, cfg.gamma)
            n += 1
    except AssertionError as e:
        return g, n, "MISMATCH in episode %d: %s" % (n, e)
    return g, n, None


def main():
    import torch
    import pseudonet
    from conftest import make_cfg
    from alphafive_amd.engine import SelfPlayEngine
    from test_gpu_fullsize import _play_until_every_game_finished
    G, memo = int(os.environ.get("G", 4096)), os.environ.get("MEMO", "0") == "1"
    cfg = make_cfg(board_size=BOARD, simulation_per_step=SIMS, upper_simulation_per_step=UPPER)
    t0 = time.time()
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED, weights_version=0,
                        eval_memo=dict(log2_buckets=18, max_stones=5) if memo else None)
    got = _play_until_every_game_finished(sp, G, max_rounds=1000)
    ct = sp.counters()
    ms = sp.engine.memo_stats() if memo else None
    ticks = sp.ticks
    sp.close()
    t_gpu = time.time() - t0
    assert len(got) == G, "some games never finished"
    procs = int(os.environ.get("PROCS", len(os.sched_getaffinity(0))))
    t1 = time.time()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(check_game, sorted(got.items()), chunksize=16)
    bad = [r for r in res if r[2]]
    print(json.dumps({"board": BOARD, "sims": SIMS, "upper": UPPER, "games": G, "memo": memo, "ticks": ticks, "episodes_finished": int(ct["episodes"]),
                      "episodes_compared_with_the_oracle": int(sum(r[1] for r in res)), "games_with_two_episodes_compared": sum(r[1] == 2 for r in res),
                      "mismatches": len(bad), "first_mismatches": [(r[0], r[2][:120]) for r in bad[:5]],
                      "plies_compared": int(sum(sum(e["T"] for e in eps[:2]) for eps in got.values())),
                      "memo_stats": ms, "gpu_s": round(t_gpu, 1), "oracle_s": round(time.time() - t1, 1), "oracle_processes": procs}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
