"""In-process rehearsal transport for gogp_amd.sharded.ShardedGP: G ranks as G threads of ONE
process sharing one GPU, exchanging host buffers through queues.  It lets the tests run
process grids (2x4, 4x4) that would need more GPU processes than a test box allows; the
library side is exactly the callback transport a multi-process host would use."""
import queue
import threading

import numpy as np


class Loopback:
    def __init__(self, world):
        self.world = world
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.barrier = threading.Barrier(world)
        self.lock = threading.Lock()
        self.acc = None
        self.sent_bytes = [0] * world

    def exchange(self, rank, ops):
        for peer, is_send, mv in ops:
            if is_send:
                self.q[(rank, peer)].put(bytes(mv))
                self.sent_bytes[rank] += len(mv)
        for peer, is_send, mv in ops:
            if not is_send:
                data = self.q[(peer, rank)].get(timeout=120)
                assert len(data) == len(mv), (rank, peer, len(data), len(mv))
                mv[:] = data

    def allreduce(self, rank, arr):
        with self.lock:
            if self.acc is None:
                self.acc = np.zeros_like(arr)
            self.acc += arr
        self.barrier.wait(timeout=120)
        arr[:] = self.acc
        self.barrier.wait(timeout=120)
        if rank == 0:
            self.acc = None
        self.barrier.wait(timeout=120)


def run_ranks(world, fn):
    """fn(rank, loopback) in `world` threads; re-raises the first failure."""
    lb = Loopback(world)
    errs = [None] * world
    outs = [None] * world

    def body(r):
        try:
            outs[r] = fn(r, lb)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e
            try:
                lb.barrier.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errs:
        if e is not None:
            raise e
    return outs, lb
