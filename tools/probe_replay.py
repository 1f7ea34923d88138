"""Scratch probe: RandomStack.get_data on the host vs DeviceRandomStack.get_data (one kernel launch)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from alphafive_amd import utils
from alphafive_amd.replay import DeviceRandomStack
from test_gpu_replay import _episodes
B = int(os.environ.get("B", 512))
eps = _episodes(11, 400, seed=1)
import io, contextlib
for cls in (utils.RandomStack, DeviceRandomStack):
    random.seed(0); np.random.seed(0)
    st = cls(11, 8000)
    with contextlib.redirect_stdout(io.StringIO()):
        for rec, res in eps: st.push(rec, res)
    for _ in range(3): st.get_data(B)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): out = st.get_data(B)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    print(f"{cls.__name__}: {st._size()} positions, get_data({B}) {dt*1e3:.3f} ms  ({B/dt:.0f} samples/s)")
