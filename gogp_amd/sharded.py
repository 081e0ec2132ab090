"""One GP evaluation sharded over several GPUs (one process per GPU).

``ShardedGP`` is a ``gp.GP`` whose Absorb / Observe / Gradient run the 2-D block-cyclic
sharded evaluation of libgogp_hip (include/gogp_hip.h, "one evaluation sharded over several
GPUs"; gogp_amd/csrc/dist2d.hip): the ranks form a Pr x Pc grid, tile (I, J) of the Gram
matrix lives on rank (I mod Pr, J mod Pc), every rank allocates only its own tiles.  Every
rank constructs the object with the SAME kernel, data and arguments and calls the SAME
methods in the same order (collective semantics); every rank gets the same LML, gradient and
Alpha.

Transport:
  * ``"rccl"`` (default when the process group's backend is nccl): the library itself calls
    RCCL over xGMI -- this module only carries the 128-byte unique id from rank 0 to the others;
  * ``"callbacks"`` (default otherwise, e.g. gloo): host-buffer exchange through
    ``torch.distributed`` point-to-point calls -- rehearsals where several ranks share one GPU.
The reference has no counterpart (single process).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .gp import GP


def zlib_host() -> int:
    """A small integer that differs between hosts (crc32 of the host name)."""
    import socket
    import zlib
    return zlib.crc32(socket.gethostname().encode()) & 0xFFFFFF


class ShardedGP(GP):
    def __init__(self, *args, group=None, grid=None, transport=None, rank=None, world=None,
                 exchange=None, allreduce=None, **kw):
        """``exchange(rank, ops)`` / ``allreduce(rank, array)`` replace the torch.distributed
        calls of the callback transport (ops: list of (peer, is_send, memoryview)); with them
        ``rank`` / ``world`` give this object's place -- several ranks can then live in ONE
        process (one thread each), which is how the tests rehearse 2x4 and 4x4 grids on one GPU."""
        import torch
        import torch.distributed as dist
        have_pg = dist.is_available() and dist.is_initialized() and exchange is None and transport != "replay"
        backend = dist.get_backend(group) if have_pg else ("in-process" if exchange else "none")
        nworld = dist.get_world_size(group) if have_pg else (world or 1)
        if transport is None:
            transport = "rccl" if backend == "nccl" else "callbacks"
        if transport == "rccl" and nworld > 1:
            # one process per GPU: without an explicit device every rank would sit on device 0 and
            # ncclCommInitRank would fail (or hang) on the duplicate; bind rank -> GPU here
            import os
            if kw.get("device", -1) in (None, -1):
                kw["device"] = int(os.environ.get("LOCAL_RANK", dist.get_rank(group))) % max(1, torch.cuda.device_count())
            mine = torch.tensor([float(kw["device"]), float(zlib_host())], dtype=torch.float64,
                                device=torch.device("cuda", int(kw["device"])) if backend == "nccl" else "cpu")
            every = [torch.zeros_like(mine) for _ in range(nworld)]
            dist.all_gather(every, mine, group=group)
            seen = {}
            for r, t in enumerate(every):
                key = (int(t[0].item()), int(t[1].item()))
                if key in seen:
                    raise ValueError("ranks %d and %d of the group share GPU %d of one host: the RCCL transport "
                                     "needs one GPU per rank (pass device=LOCAL_RANK, or use transport="
                                     "'callbacks' over a gloo group for rehearsals)" % (seen[key], r, key[0]))
                seen[key] = r
        super().__init__(*args, **kw)
        self._torch, self._dist, self._group = torch, dist, group
        self._user_exchange, self._user_allreduce = exchange, allreduce
        self._rank = dist.get_rank(group) if have_pg else (rank or 0)
        self._world = nworld
        self._backend = backend
        L = _lib.lib()
        if grid is None:
            pr, pc = ctypes.c_int(0), ctypes.c_int(0)
            self._check(L.gogp_dist_grid(self._world, ctypes.byref(pr), ctypes.byref(pc)))
            grid = (pr.value, pc.value)
        self.grid = (int(grid[0]), int(grid[1]))
        self.transport = transport
        self._cb_error = None
        if transport == "rccl":
            uid = torch.zeros(_lib.GOGP_UNIQUE_ID_BYTES, dtype=torch.uint8)
            if self._rank == 0:
                buf = (ctypes.c_char * _lib.GOGP_UNIQUE_ID_BYTES)()
                self._check(L.gogp_dist_unique_id(buf))
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if self._world > 1:
                dev = "cpu"
                if self._backend == "nccl":  # on the handle's own GPU, not on device 0
                    dev = torch.device("cuda", self.device) if self.device >= 0 else torch.device("cuda")
                t = uid.to(dev)
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast(t, src=src, group=group)
                uid = t.cpu()
            raw = bytes(uid.numpy().tobytes())
            self._check(L.gogp_dist_init_rccl(self._h, self._rank, self._world, self.grid[0],
                                              self.grid[1], raw))
        elif transport == "callbacks":
            if self._world > 1 and self._backend == "nccl":
                raise ValueError("the callback transport exchanges host buffers: use a gloo group")
            self._exchange_cb = _lib.EXCHANGE_FN(self._exchange)   # keep the thunks alive
            self._allreduce_cb = _lib.ALLREDUCE_FN(self._allreduce)
            self._check(L.gogp_dist_init_callbacks(
                self._h, self._rank, self._world, self.grid[0], self.grid[1],
                ctypes.cast(self._exchange_cb, ctypes.c_void_p),
                ctypes.cast(self._allreduce_cb, ctypes.c_void_p), None))
        elif transport == "replay":
            # measurement only (tools/sharded_replay.py): this object is rank `rank` of the grid ALONE on its GPU; the
            # hook library installs a transport that sends nothing and zero-fills what would have been received
            hooks = _lib.hooks()
            self._check(hooks.gogp_test_dist_init_replay(self._h, self._rank, self._world, self.grid[0], self.grid[1]))
        else:
            raise ValueError("transport must be 'rccl', 'callbacks' or 'replay'")

    # ---- callbacks (host buffers; never let an exception cross the C boundary) ----------------
    def _global(self, r):
        return self._dist.get_global_rank(self._group, r) if self._group is not None else r

    def _exchange(self, user, ops, nops):
        try:
            torch, dist = self._torch, self._dist
            if self._user_exchange is not None:
                lst = []
                for i in range(nops):
                    o = ops[i]
                    lst.append((int(o.peer), bool(o.is_send),
                                memoryview((ctypes.c_char * o.bytes).from_address(o.buf)).cast("B")))
                self._user_exchange(self._rank, lst)
                return 0
            reqs = []
            keep = []
            for i in range(nops):
                o = ops[i]
                buf = (ctypes.c_char * o.bytes).from_address(o.buf)
                t = torch.frombuffer(buf, dtype=torch.uint8)
                keep.append((buf, t))
                if o.is_send:
                    reqs.append(dist.isend(t, dst=self._global(o.peer), group=self._group))
                else:
                    reqs.append(dist.irecv(t, src=self._global(o.peer), group=self._group))
            for r in reqs:
                r.wait()
            return 0
        except Exception as e:  # noqa: BLE001
            self._cb_error = e
            return 1

    def _allreduce(self, user, host_buf, count):
        try:
            if self._user_allreduce is not None:
                self._user_allreduce(self._rank, np.ctypeslib.as_array(host_buf, shape=(count,)))
            elif self._world > 1:
                arr = np.ctypeslib.as_array(host_buf, shape=(count,))
                t = self._torch.from_numpy(arr)
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
            return 0
        except Exception as e:  # noqa: BLE001
            self._cb_error = e
            return 1

    def _check(self, rc):
        if rc != _lib.GOGP_OK and self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise RuntimeError("communication callback failed") from e
        super()._check(rc)

    # ---- descriptions for reports -------------------------------------------------------------
    def grid_text(self) -> str:
        return "%dx%d" % self.grid

    def transport_text(self) -> str:
        return ("RCCL inside libgogp_hip (grouped ncclSend/ncclRecv + ncclAllReduce)"
                if self.transport == "rccl" else
                "host callbacks over torch.distributed (%s)" % self._backend)

    def comm_ranks(self):
        """(ranks of the communicator as the transport counts them, True for RCCL): ncclCommCount."""
        flag = ctypes.c_int(0)
        n = int(_lib.lib().gogp_dist_comm_ranks(self._h, ctypes.byref(flag)))
        return n, bool(flag.value)

    def selftest(self, phase: int, count: int = 1 << 16) -> None:
        """Pre-flight of the transport (collective): phase 0 = one grouped send/recv ring, phase 1 = one
        all-reduce; raises on a wrong payload.  A hang is for the caller's watchdog to catch."""
        self._check(_lib.lib().gogp_dist_selftest(self._h, int(phase), int(count)))

    def local_bytes(self) -> int:
        """Device bytes of this rank's shard (its tiles of K / L / Y, panel buffers)."""
        return int(_lib.lib().gogp_dist_local_bytes(self._h))
