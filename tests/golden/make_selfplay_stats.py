#!/usr/bin/env python
"""tests/golden/selfplay_stats_6960.npz: the replay buffer bookkeeping the reference saved next to its shipped checkpoint
(/root/reference/data_buffer/data_len6960.pkl, result6960.pkl — written by RandomStack.save at step 6960, utils.py:29-41): the
lengths and results of the 470 accepted self-play episodes that filled the 12,000-position buffer, i.e. games the TensorFlow net
of steps ~6490..6960 played through genData/player.py at 542 / 642 simulations, after RandomStack.push's acceptance, duplication
and eviction rules (utils.py:65-115).  The positions themselves (data6960.pkl) are not in the reference repository.
usage: python tests/golden/make_selfplay_stats.py [/root/reference]"""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    with open(os.path.join(ref, "data_buffer", "data_len6960.pkl"), "rb") as f:
        data_len = pickle.load(f)
    with open(os.path.join(ref, "data_buffer", "result6960.pkl"), "rb") as f:
        result = pickle.load(f)
    assert len(data_len) == len(result)
    out = os.path.join(HERE, "selfplay_stats_6960.npz")
    np.savez_compressed(out, data_len=np.asarray(data_len, np.int32), result=np.asarray(result, np.int32),
                        buffer_size=np.int32(12000), simulation_per_step=np.int32(542), upper_simulation_per_step=np.int32(642))
    dl, rs = np.asarray(data_len), np.asarray(result)
    print("%s: %d episodes, %d positions, mean length %.3f (%d..%d), black:white:draw %d:%d:%d" %
          (out, len(dl), dl.sum(), dl.mean(), dl.min(), dl.max(), (rs == 1).sum(), (rs == -1).sum(), (rs == 0).sum()))


if __name__ == "__main__":
    main()
