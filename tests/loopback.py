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
        # what every rank asked of the transport, in order: ("group", [(peer, is_send, nbytes), ...])
        # or ("allreduce", nbytes) -- input of check_rendezvous()
        self.log = [[] for _ in range(world)]

    def exchange(self, rank, ops):
        self.log[rank].append(("group", [(peer, bool(is_send), len(mv)) for peer, is_send, mv in ops]))
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
        self.log[rank].append(("allreduce", arr.nbytes))
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


def check_rendezvous(logs):
    """Would the recorded schedule complete on a transport WITHOUT buffering -- RCCL's grouped
    ncclSend / ncclRecv, where a group finishes only when each of its transfers has met the
    matching transfer of the peer's CURRENT group, and an all-reduce only when every rank is in
    it?  The queues above buffer sends, so a schedule with a cyclic wait (rank A's group k sends to
    B whose matching receive sits in its group k+1, and vice versa) would pass here and hang on
    the GPUs.  Returns None, or a description of the first state that cannot make progress."""
    world = len(logs)
    pos = [0] * world
    pending = [None] * world  # unmatched transfers of the rank's current group

    def load(r):
        while pos[r] < len(logs[r]):
            kind, payload = logs[r][pos[r]]
            if kind == "group":
                if not payload:
                    pos[r] += 1
                    continue
                pending[r] = list(payload)
            else:
                pending[r] = None
            return

    for r in range(world):
        load(r)
    while any(pos[r] < len(logs[r]) for r in range(world)):
        progress = False
        active = [r for r in range(world) if pos[r] < len(logs[r])]
        # all-reduce: every rank must have arrived (and all of them with the same size)
        if all(logs[r][pos[r]][0] == "allreduce" for r in active) and len(active) == world:
            sizes = {logs[r][pos[r]][1] for r in active}
            if len(sizes) != 1:
                return "all-reduce sizes differ: %s" % sorted(sizes)
            for r in active:
                pos[r] += 1
                load(r)
            continue
        for r in active:
            if logs[r][pos[r]][0] != "group":
                continue
            for op in list(pending[r]):
                peer, is_send, nb = op
                if pos[peer] >= len(logs[peer]) or logs[peer][pos[peer]][0] != "group":
                    continue
                want = (r, not is_send, nb)
                if want in pending[peer]:
                    pending[peer].remove(want)
                    pending[r].remove(op)
                    progress = True
        for r in active:
            if logs[r][pos[r]][0] == "group" and not pending[r]:
                pos[r] += 1
                load(r)
                progress = True
        if not progress:
            return "no progress at " + "; ".join(
                "rank %d step %d/%d %s" % (r, pos[r], len(logs[r]),
                                           (logs[r][pos[r]][0], pending[r]) if pos[r] < len(logs[r]) else "done")
                for r in range(world))
    return None


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
