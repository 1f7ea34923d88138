"""Host-side mirror of the reference's utils.py game helpers (utils.py:149-296) and replay
record conventions.  Same names, arguments and results (bit-for-bit; checked against golden
vectors recorded from the reference in tests/test_host_utils.py), written on numpy
vector ops.  The device engine has its own bitboard versions of these rules
(csrc/af_engine.hip); these exist for the API boundary (state strings, records).

Board convention (utils.py:185,194,275-283): int8[S,S], +1 = stone of the player to move,
-1 = opponent, 0 = empty.
"""
import numpy as np

BLACK_WIN = 1      # utils.py:9-11
WHITE_WIN = -1
DRAW = 0


def board_to_state(board):
    """utils.py:156-175 — run-length text key: empties as chr(ord('a')+run), '3' mine, '1' theirs, '/' per row."""
    rows = []
    for row in np.asarray(board):
        out, run = [], 0
        for v in row.tolist():
            if v == 0:
                run += 1
                continue
            if run:
                out.append(chr(97 + run))
                run = 0
            out.append(str(v + 2))
        if run:
            out.append(chr(97 + run))
        rows.append("".join(out))
    return "/".join(rows) + "/"


def state_to_board(state, board_size):
    """utils.py:178-196."""
    board = np.zeros((board_size, board_size), np.int8)
    for i, row in enumerate(state.split("/")[:board_size]):
        j = 0
        for ch in row:
            if ch.isalpha():
                j += ord(ch) - 97
            else:
                board[i, j] = int(ch) - 2
                j += 1
    return board


def is_game_over(board, goal):
    """utils.py:199-235 — (over, value from the player-to-move's view).  The reference scans cells
    row-major and at each cell tests down, right, down-right, up-right windows of `goal`; the first
    hit in that order wins (matters only if both colours have a line)."""
    b = np.asarray(board, np.int32)
    h, w = b.shape
    best = None   # (cell*4+dir, value)
    if goal <= h:
        win = np.lib.stride_tricks.sliding_window_view
        cands = []
        if h >= goal:
            s = win(b, goal, axis=0).sum(-1)                               # down: anchor (i,j), i<=h-goal
            cands.append((0, s, 0, 0))
            s = win(b, goal, axis=1).sum(-1)                               # right: anchor (i,j), j<=w-goal
            cands.append((1, s, 0, 0))
            idx = np.arange(goal)
            d = win(b, (goal, goal))                                       # [h-g+1, w-g+1, g, g]
            cands.append((2, d[:, :, idx, idx].sum(-1), 0, 0))             # down-right from (i,j)
            cands.append((3, d[:, :, idx[::-1], idx].sum(-1), goal - 1, 0))  # up-right from (i+g-1, j)
        for direction, s, di, dj in cands:
            hit = np.argwhere(np.abs(s) == goal)
            for i, j in hit:
                key = ((i + di) * w + (j + dj)) * 4 + direction
                if best is None or key < best[0]:
                    best = (key, 1.0 if s[i, j] > 0 else -1.0)
    if best is not None:
        return True, best[1]
    if not (b == 0).any():
        return True, 0.0
    return False, 0.0


def get_legal_actions(board):
    """utils.py:238-245 — empties in row-major order; this order defines edge index everywhere."""
    ii, jj = np.nonzero(np.asarray(board) == 0)
    return [(int(i), int(j)) for i, j in zip(ii, jj)]


def board_to_inputs(board, type_=np.float32, last_action=None):
    """utils.py:256-272 — planes [mine, theirs, one-hot(last_action)]."""
    b = np.asarray(board)
    out = np.zeros((3,) + b.shape, dtype=type_)
    out[0] = b == 1
    out[1] = b == -1
    if last_action is not None:
        out[2, last_action[0], last_action[1]] = 1
    return out


def step(board, action):
    """utils.py:275-283 — place a stone for the player to move, flip perspective (mutates like the reference)."""
    board[action[0], action[1]] = 1
    return -board


def construct_weights(length, gamma=0.95):
    """utils.py:286-296 — per-ply training weights, fp32: w[T-1]=1, w[i]=w[i+1]*gamma, then T*w/sum(w)."""
    length = int(length)
    g = np.full(length, np.float32(gamma), np.float32)
    g[0] = np.float32(1.0)
    w = np.multiply.accumulate(g, dtype=np.float32)[::-1].copy()        # sequential fp32 products
    return length * w / np.sum(w)


def softmax(x):
    e = np.exp(x - np.max(x))
    return e / np.sum(e)


class RandomStack(object):
    """Replay buffer with the reference's behaviour and call surface (utils.py:14-146): FIFO of
    per-position 5-tuples `(state, policy[S,S], last_action, value, weight)`; push() rejects short
    games at random (utils.py:81), duplicates episodes of the under-represented winner
    (utils.py:86-100) and evicts from the front with partial-episode bookkeeping (utils.py:101-115);
    get_data() samples without replacement and applies one of the 8 board symmetries, remapping
    last_action (utils.py:118-146).  Draws from the same global streams in the same order as the
    reference (`random.random`, `np.random.choice`, `random.choice`), so a seeded run is reproducible
    against it (tests/test_host_utils.py)."""

    def __init__(self, board_size, length=2000):
        import time as _time
        self.data = []
        self.board_size = board_size
        self.length = length
        self.white_win = 0
        self.black_win = 0
        self.data_len = []
        self.result = []
        self.total_length = 0
        self.num = 0
        self.time = _time.time()
        self.self_play_black_win = 0
        self.self_play_white_win = 0

    # ---- persistence: same three pickles as utils.py:29-57 ----
    _FILES = (("data", "data"), ("data_len", "data_len"), ("result", "result"))

    def save(self, s=""):
        import pickle
        for attr, stem in self._FILES:
            with open(f"data_buffer/{stem}{s}.pkl", "wb") as f:
                pickle.dump(getattr(self, attr), f)

    def load(self, s=""):
        import pickle
        for attr, stem in self._FILES:
            with open(f"data_buffer/{stem}{s}.pkl", "rb") as f:
                setattr(self, attr, pickle.load(f))
        self.white_win = self.result.count(WHITE_WIN)
        self.black_win = self.result.count(BLACK_WIN)
        print("load data successfully, with length %d" % len(self.data))
        print("black: white = %d: %d in the memory" % (self.black_win, self.white_win))

    def isEmpty(self):
        return self._size() == 0

    def is_full(self):
        return self._size() >= self.length

    # ---- storage hooks (alphafive_amd.replay.DeviceRandomStack keeps the positions in HBM instead) ----
    def _size(self):
        return len(self.data)

    def _store(self, data):
        self.data.extend(data)

    def _drop_front(self, n):
        del self.data[:n]

    def _append_episode(self, data, result):
        self._store(data)
        self.data_len.append(len(data))
        self.result.append(result)

    def push(self, data, result):
        import random as _random
        import time as _time
        n = len(data)
        self.total_length += n
        self.num += 1
        if result == BLACK_WIN:
            self.self_play_black_win += 1
        elif result == WHITE_WIN:
            self.self_play_white_win += 1
        if self.total_length >= 100:
            now = _time.time()
            print("black: white = %d: %d in the memory, avg_length: %0.1f avg: %0.3fs per piece" % (
                self.black_win, self.white_win, self.total_length / self.num, (now - self.time) / self.total_length))
            print("self-play black: %d, white: %d" % (self.self_play_black_win, self.self_play_white_win))
            self.total_length = self.num = 0
            self.time = now
        # short games are dropped with probability -0.0682*T + 1.364 (T=9: 0.75, T>=20: 0)
        if _random.random() <= -0.0682 * n + 1.364:
            return False
        self._append_episode(data, result)
        if result == BLACK_WIN:
            self.black_win += 1
            if _random.random() < (self.white_win - self.black_win) / (self.black_win * 1.3):
                self._append_episode(data, result)
                self.black_win += 1
        elif result == WHITE_WIN:
            self.white_win += 1
            if _random.random() < (self.black_win - self.white_win) / (self.white_win * 1.02):
                self._append_episode(data, result)
                self.white_win += 1
        beyond = self._size() - self.length
        if beyond > 0:
            self._drop_front(beyond)
            while beyond >= self.data_len[0]:           # whole episodes fall out of the front
                beyond -= self.data_len.pop(0)
                gone = self.result.pop(0)
                if gone == BLACK_WIN:
                    self.black_win -= 1
                elif gone == WHITE_WIN:
                    self.white_win -= 1
            self.data_len[0] -= beyond                  # the front episode is cut short
        return True

    def get_data(self, batch_size=1):
        import random as _random
        S = self.board_size
        num = min(batch_size, len(self.data))
        idx = np.random.choice(len(self.data), size=num, replace=False)
        boards = np.empty((num, 3, S, S), dtype=np.float32)
        weights = np.empty((num,), dtype=np.float32)
        values = np.empty((num,), dtype=np.float32)
        policies = np.empty((num, S, S), dtype=np.float32)
        for i, ix in enumerate(idx):
            state, p, la, v, w = self.data[ix]
            board = state_to_board(state, S)
            k = np.random.choice([0, 1, 2, 3])          # quarter turns
            board = np.rot90(board, k=k, axes=(0, 1))
            p = np.rot90(p, k=k, axes=(0, 1))
            if la is not None:
                for _ in range(int(k)):                 # one quarter turn: (i, j) -> (S-1-j, i)
                    la = (S - 1 - la[1], la[0])
            if _random.choice([1, 2]) == 1:             # vertical flip
                board = np.flip(board, axis=0)
                p = np.flip(p, axis=0)
                if la is not None:
                    la = (S - 1 - la[0], la[1])
            boards[i] = board_to_inputs(board, last_action=la)
            weights[i] = w
            values[i] = v
            policies[i] = p
        return boards, weights, values, policies.reshape((num, S * S))
