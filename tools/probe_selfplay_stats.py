"""Calibration of tests/test_gpu_selfplay_stats.py: buffer statistics of fixed-weight self-play for the 6960 weights and for
deliberately broken forwards (weight-space forms of classic bugs).  usage: python tools/probe_selfplay_stats.py [G]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_selfplay_stats as T

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with np.load(T.W) as z:
    variables = {k: z[k] for k in z.files}
f = T._fixture()
print("reference buffer: mean length %.2f, black share %.3f" % (f["mean_len"], f["black_share"]))
for how in ["good"] + sys.argv[2:]:
    v = variables if how == "good" else T.broken_variables(variables, how)
    for seed in (3, 4):
        st = T.selfplay_buffer_stats(v, G, seed=seed)
        print("%-12s seed %d: buffer mean length %.2f  black share %.3f | averaged over %d buffer states %.2f / %.3f | raw mean %.2f black %.3f draws %d max %d" %
              (how, seed, st["buffer_mean_len"], st["buffer_black_share"], st["snapshots"], st["avg_mean_len"], st["avg_black_share"],
               st["raw_mean_len"], st["raw_black_share"], st["draws"], st["len_max"]), flush=True)
