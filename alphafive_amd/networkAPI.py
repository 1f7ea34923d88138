"""Leaf-evaluation service with the reference's call surface (genData/networkAPI.py:10-83):

    NetworkAPI(cfg, agent_model) .start(reload) .get_pipe(reload) -> Connection .close()

One daemon thread multiplexes the pipe ends: it drains every ready pipe, stacks the pending
leaves into one float32[B,3,S,S] batch, calls agent_model.eval once and answers each pipe
with its slice, in arrival order, as [(policy_row, float(value)), ...] (networkAPI.py:43-78).
Clients speak the protocol of player.py:194-197 (send([x]); poll(); recv()[0]).
`agent_model` needs .graph.as_default() and .eval(ndarray) — alphafive_amd.network.ResNet
or the reference's own model object.
"""
from logging import getLogger
from multiprocessing import Pipe, connection
from threading import Thread

import numpy as np

logger = getLogger(__name__)


class NetworkAPI(object):
    def __init__(self, cfg=None, agent_model=None):
        self.agent_model = agent_model
        self.config = cfg
        self.pipes = []
        self.reload = True
        self.prediction_worker = None
        self.done = False

    def start(self, reload):
        self.reload = reload
        self.prediction_worker = Thread(target=self.predict_batch_worker, name="prediction_worker", daemon=True)
        self.prediction_worker.start()

    def get_pipe(self, reload=True):
        mine, theirs = Pipe()
        self.pipes.append(mine)
        self.reload = reload
        return theirs

    def _drain(self, ready):
        """-> (list of leaf arrays, [(pipe, count)]) for every message waiting on the ready pipes."""
        leaves, owners = [], []
        for pipe in ready:
            try:
                while pipe.poll():
                    msg = pipe.recv()
                    leaves.extend(msg)
                    owners.append((pipe, len(msg)))
            except (EOFError, OSError) as exc:        # peer went away: log and drop it (networkAPI.py:57-59)
                logger.error(f"EOF error: {exc}")
                pipe.close()
                if pipe in self.pipes:
                    self.pipes.remove(pipe)
        return leaves, owners

    def predict_batch_worker(self):
        while not self.done:
            live = [p for p in self.pipes if not p.closed]
            if not live:
                connection.wait([], timeout=0.001)
                continue
            try:
                ready = connection.wait(live, timeout=0.001)
            except OSError:
                continue
            if not ready:
                continue
            leaves, owners = self._drain(ready)
            if not leaves:
                continue
            batch = np.asarray(leaves, dtype=np.float32)
            with self.agent_model.graph.as_default():
                policy, value = self.agent_model.eval(batch)
            at = 0
            for pipe, count in owners:
                reply = [(policy[at + k], float(value[at + k])) for k in range(count)]
                at += count
                try:
                    pipe.send(reply)
                except (BrokenPipeError, OSError) as exc:
                    logger.error(f"send failed: {exc}")

    def close(self):
        self.done = True
        worker = self.prediction_worker
        if worker is not None and worker.is_alive():
            worker.join(timeout=1.0)      # let the worker leave connection.wait before the pipes go away
        for pipe in self.pipes:
            pipe.close()
