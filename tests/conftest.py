import os
import sys
import types

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / by the driver)")


def make_cfg(**kw):
    """Attribute bag with the reference's config.py names and default values (config.py:2-20)."""
    base = dict(board_size=11, goal=5, simulation_per_step=542, upper_simulation_per_step=642, init_temp=1.2,
                gamma=0.94, tau_decay_rate=0.94, tau_decay_rate_r=0.9, dirichlet_alpha=0.3, c_puct=5.0,
                buffer_size=12000, batch_size=512, max_processes=5)
    base.update(kw)
    return types.SimpleNamespace(**base)


def cfg_from_golden(z):
    keys = ["board_size", "goal", "simulation_per_step", "upper_simulation_per_step", "init_temp", "gamma",
            "tau_decay_rate", "tau_decay_rate_r", "dirichlet_alpha", "c_puct"]
    kw = {}
    for k in keys:
        v = z["cfg_" + k]
        kw[k] = int(v) if k in ("board_size", "goal", "simulation_per_step", "upper_simulation_per_step") else float(v)
    return make_cfg(**kw)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def run_in_threads(fn, items, workers=None):
    """fn(item) for every item on host threads (the C oracle runs outside the GIL: ctypes.CDLL calls release it, and every
    OraclePlayer owns its handle), results in order; the first exception is re-raised.  The GPU tests' oracle replays are
    per-game independent, and the GPU boxes have many more cores than one."""
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    if not items:
        return []
    n = workers or min(len(items), len(os.sched_getaffinity(0)))
    with ThreadPoolExecutor(max_workers=max(1, n)) as ex:
        return list(ex.map(fn, items))
