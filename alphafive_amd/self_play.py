"""Headless version of the reference's AI-vs-AI demo loop (self_play.py:79-106) — BASELINE config 1:
Player(training=False, pv_fn=net.eval); every frame `action = None` is passed as last_action
(self_play.py:95-97), the tree is never reset and is shared by both colours; stop at game over.

    python -m alphafive_amd.self_play --ckpt /path/to/ckpt [--sims 500 --upper 642]
"""
import argparse
import time

from . import config as default_config
from . import utils
from .network import ResNet
from .player import Player


def play(cfg, net, seed=0, verbose=True):
    player = Player(cfg, training=False, pv_fn=net.eval, seed=seed)
    state_str = player.get_init_state()
    game_over, turn, moves = False, 0, []
    t0 = time.time()
    while not game_over:
        action = None
        _, action = player.get_action(state_str, last_action=action)
        board = utils.step(utils.state_to_board(state_str, cfg.board_size), action)
        state_str = utils.board_to_state(board)
        game_over, value = utils.is_game_over(board, cfg.goal)
        moves.append(action)
        turn += 1
        if verbose:
            print(f"ply {turn:3d} {'black' if turn % 2 else 'white'} -> {action}", flush=True)
    dt = time.time() - t0
    player.close()
    return moves, value, dt


def main():
    import types
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=default_config.ckpt_path)
    ap.add_argument("--weights-npz", default=None)
    ap.add_argument("--sims", type=int, default=500)          # choose_best_player.py:25
    ap.add_argument("--upper", type=int, default=default_config.upper_simulation_per_step)
    args = ap.parse_args()
    cfg = types.SimpleNamespace(**{k: getattr(default_config, k) for k in dir(default_config)
                                   if not k.startswith("_") and k != "get_lr"})
    cfg.simulation_per_step, cfg.upper_simulation_per_step = args.sims, args.upper
    net = ResNet(cfg.board_size)
    if args.weights_npz:
        net.load_npz(args.weights_npz)
    else:
        net.restore(args.ckpt)
    moves, value, dt = play(cfg, net)
    print(f"game finished: {len(moves)} plies in {dt:.1f} s = {len(moves) / dt:.2f} moves/s")


if __name__ == "__main__":
    main()
