"""Every game, not a sample: configs[1]'s settings (11x11, 500 sims/move, cap 642, pseudo-net) run until every game has finished an
episode, then EVERY first episode (and every second one that exists) is replayed by the C oracle on the host cores and compared
bit for bit.  tests/test_gpu_fullsize.py::test_every_game_of_a_512_game_engine_equals_the_oracle calls sweep() at 512 games inside
the test budget; the command line runs the full 4096 (3.5-4 min of host time on 16 cores, profiles/r5_40).
Env: BOARD / SIMS / UPPER (11 / 500 / 642; 15 / 800 / 942 = configs[3]), G (4096), MEMO (0 / 1: the evaluation memo on), PROCS (host
processes, default = usable cores)."""
import json
import multiprocessing as mp
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
SALT, PEAK, SEED = 777, 8192, 42


def check_game(item):
    g, eps, board, sims, upper = item                 # (eps: the episodes to compare, already cut to length)
    import oracle
    from conftest import make_cfg
    from test_gpu_fullsize import _assert_episode_equals_oracle
    cfg = make_cfg(board_size=board, simulation_per_step=sims, upper_simulation_per_step=upper)
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=SEED, game_id=g, pseudo_salt=SALT, pseudo_peak=PEAK)
    n = 0
    try:
        for raw in eps[:2]:
            _assert_episode_equals_oracle(raw, orc, board, cfg.gamma)
            n += 1
    except AssertionError as e:
        return g, n, "MISMATCH in episode %d: %s" % (n, e)
    return g, n, None


def sweep(G=4096, board=11, sims=500, upper=642, memo=False, procs=None, first_game_id=0, second_every=1):
    import pseudonet
    from conftest import make_cfg
    from alphafive_amd.engine import SelfPlayEngine
    from test_gpu_fullsize import _play_until_every_game_finished
    cfg = make_cfg(board_size=board, simulation_per_step=sims, upper_simulation_per_step=upper)
    t0 = time.time()
    sp = SelfPlayEngine(cfg, G, lambda x: pseudonet.pseudonet_torch(x, SALT, PEAK), device=0, seed=SEED, weights_version=0,
                        first_game_id=first_game_id, eval_memo=dict(log2_buckets=18, max_stones=5) if memo else None)
    got = _play_until_every_game_finished(sp, G, max_rounds=1000)
    ct = sp.counters()
    ms = sp.engine.memo_stats() if memo else None
    ticks = sp.ticks
    sp.close()
    t_gpu = time.time() - t0
    assert len(got) == G, "some games never finished"
    procs = procs or len(os.sched_getaffinity(0))
    t1 = time.time()
    # the first episode of every game; the second one (if finished) of every `second_every`-th game
    items = [(first_game_id + g, eps[:2 if g % second_every == 0 else 1], board, sims, upper) for g, eps in sorted(got.items())]
    # (processes, not threads: 256 host threads on the ctypes oracle measured 74 s against 46 s for 256 spawned processes — the players'
    # Python glue serialises on the GIL; the sweep's floor is its longest game, ~40 s for a 121-ply game of the pseudo-net)
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(check_game, items, chunksize=max(1, min(16, G // (4 * procs))))
    bad = [r for r in res if r[2]]
    return {"board": board, "sims": sims, "upper": upper, "games": G, "first_game_id": first_game_id, "memo": memo, "ticks": ticks,
            "episodes_finished": int(ct["episodes"]),
            "episodes_compared_with_the_oracle": int(sum(r[1] for r in res)), "games_with_two_episodes_compared": sum(r[1] == 2 for r in res),
            "mismatches": len(bad), "first_mismatches": [(r[0], r[2][:120]) for r in bad[:5]],
            "plies_compared": int(sum(sum(e["T"] for e in it[1]) for it in items)),
            "memo_stats": ms, "gpu_s": round(t_gpu, 1), "oracle_s": round(time.time() - t1, 1), "oracle_processes": procs}


def main():
    rep = sweep(G=int(os.environ.get("G", 4096)), board=int(os.environ.get("BOARD", 11)), sims=int(os.environ.get("SIMS", 500)),
                upper=int(os.environ.get("UPPER", 642)), memo=os.environ.get("MEMO", "0") == "1",
                procs=int(os.environ["PROCS"]) if "PROCS" in os.environ else None)
    print(json.dumps(rep))
    return 1 if rep["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
