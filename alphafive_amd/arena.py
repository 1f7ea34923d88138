"""Batched checkpoint arena — the match loop of the reference's choose_best_player.py:38-60 for many
games at once (a caller of the hot path).

Reference protocol, per game i: both players are reset (:43-44), player (i % 2) moves first (:48), players
alternate, each calling get_action(state, last_action=action, random_a=True) on its OWN tree with
training=False (:32-33,52) — temperature play with tau *= tau_decay_rate_r per own move (player.py:108-109) —
until is_game_over; a non-draw is a win for the player who just moved (:62-63).

Here every game is one slot of two EXTERNAL-mode engines (one per player / weight set).  At ply k the mover
of game i is player (i + k) % 2, so each ply activates the even games in one engine and the odd games in the
other; a mover's engine runs its tick loop (tree kernel -> leaf batch -> its own net) until all of its active
games have decided their move.
"""
import numpy as np

from . import engine as _eng
from . import utils


def play_matches(cfg, pv0, pv1, n_games, device=0, seed0=0, seed1=1, node_cap=0, max_plies=None):
    """pv0 / pv1: device evaluators (planes float32[G,3,S,S] -> (prob[G,C], value[G]) torch tensors).
    Returns dict(wins=[w0, w1], draws, moves=[[cell,...] per game], lengths)."""
    import torch
    S, G = cfg.board_size, n_games
    C = S * S
    dev = torch.device("cuda", device)
    engines = [_eng.Engine(cfg, G, device=device, mode=_eng.MODE_EXTERNAL, training=False, seed=s, node_cap=node_cap)
               for s in (seed0, seed1)]
    pvs = [pv0, pv1]
    planes = torch.zeros((G, 3, S, S), dtype=torch.float32, device=dev)
    policy = [torch.zeros((G, C), dtype=torch.float32, device=dev) for _ in range(2)]
    value = [torch.zeros((G,), dtype=torch.float32, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream(dev).cuda_stream
    boards = [np.zeros((S, S), np.int8) for _ in range(G)]
    last = [None] * G
    over = [False] * G
    moves = [[] for _ in range(G)]
    wins, draws = [0, 0], 0
    ply = 0
    limit = max_plies or C
    while not all(over) and ply < limit:
        for p in (0, 1):
            active = [i for i in range(G) if not over[i] and (i + ply) % 2 == p]
            if not active:
                continue
            e = engines[p]
            keys = np.stack([_eng.state_to_key(utils.board_to_state(boards[i]), S) for i in active])
            lcs = [-1 if last[i] is None else last[i][0] * S + last[i][1] for i in active]
            e.set_roots(active, keys, lcs, random_a=True, reset_tree=(ply < 2), stream=stream)   # Player.reset() before each game
            while True:
                e.tick(policy[p].data_ptr(), value[p].data_ptr(), planes.data_ptr(), stream)
                st = e.status(stream)
                if all(st[i] == _eng.STATUS_MOVE_DONE for i in active):
                    break
                pr, va = pvs[p](planes)
                policy[p].copy_(pr.reshape(G, C))
                value[p].copy_(va.reshape(G))
            cells = e.move_results(active, stream=stream)[0]
            for i, cell in zip(active, cells):
                cell = int(cell)
                a = (cell // S, cell % S)
                moves[i].append(cell)
                boards[i] = utils.step(boards[i], a)
                last[i] = a
                done, v = utils.is_game_over(boards[i], cfg.goal)
                if done:
                    over[i] = True
                    if v == 0.0:
                        draws += 1
                    else:
                        wins[p] += 1                     # the player who just moved (choose_best_player.py:62-63)
        ply += 1
    for e in engines:
        e.close()
    return dict(wins=wins, draws=draws, moves=moves, lengths=[len(m) for m in moves])
