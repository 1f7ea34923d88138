"""NetworkAPI (genData/networkAPI.py call surface): k client threads speaking player.py:194-197's
protocol against one batching worker with a fake agent_model."""
import threading

import numpy as np

from alphafive_amd.networkAPI import NetworkAPI
from alphafive_amd.network import _Graph


class FakeModel(object):
    def __init__(self):
        self.graph = _Graph()
        self.batches = []

    def eval(self, data):
        assert data.dtype == np.float32 and data.ndim == 4
        self.batches.append(data.shape[0])
        s = data.reshape(data.shape[0], -1).sum(1)
        return np.tile(s[:, None], (1, 4)).astype(np.float32), (s * 0.5).astype(np.float32)


def test_replies_in_request_order_per_pipe():
    model = FakeModel()
    api = NetworkAPI(None, model)
    api.start(True)
    pipes = [api.get_pipe() for _ in range(3)]
    results = {}

    def client(i, pipe):
        out = []
        for m in range(40):
            x = np.full((3, 2, 2), i * 100 + m, np.float32)
            pipe.send([x])
            while not pipe.poll():
                pass
            policy, value = pipe.recv()[0]
            out.append((float(policy[0]), value))
        results[i] = out

    ts = [threading.Thread(target=client, args=(i, p)) for i, p in enumerate(pipes)]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    api.close()
    for i in range(3):
        assert len(results[i]) == 40
        for m, (p0, v) in enumerate(results[i]):
            assert p0 == 12.0 * (i * 100 + m) and v == 6.0 * (i * 100 + m) and isinstance(v, float)
    assert sum(model.batches) == 120 and max(model.batches) <= 3
