"""One GP evaluation sharded over several GPUs (one process per GPU).

``ShardedGP`` is a ``gp.GP`` whose Observe / Gradient run the 1-D block-cyclic
sharded factorisation of libgogp_hip (include/gogp_hip.h, "one evaluation sharded
over several GPUs").  Every rank constructs it with the SAME kernel, data and
arguments and calls the SAME methods in the same order (collective semantics);
every rank gets the same LML, gradient, Alpha and L, so Produce works on any rank.

Communication: ``torch.distributed`` -- backend "nccl" (RCCL over xGMI) on a real
node, "gloo" for rehearsals (several ranks may then share one GPU).  The panel
broadcast goes through a staging tensor owned here; the library packs / unpacks it.
The reference has no counterpart (single process).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .gp import GP


class ShardedGP(GP):
    def __init__(self, *args, group=None, **kw):
        import torch
        import torch.distributed as dist
        super().__init__(*args, **kw)
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("ShardedGP needs an initialised torch.distributed process group")
        self._torch, self._dist, self._group = torch, dist, group
        self._rank = dist.get_rank(group)
        self._world = dist.get_world_size(group)
        self._backend = dist.get_backend(group)
        self._staging = None
        self._staging_n = -1

        def bcast(user, dev_buf, nbytes, root):
            try:
                st = self._staging
                assert dev_buf == st.data_ptr() and nbytes <= st.numel() * 8
                view = st[: (nbytes + 7) // 8]
                # group-relative root -> global rank
                src = dist.get_global_rank(group, root) if group is not None else root
                dist.broadcast(view, src=src, group=group)
                # Wait for the collective only: the library's own streams (non-blocking, so
                # the null stream does not join them) keep running this rank's bulk update
                # meanwhile; a device-wide synchronize here would serialise the two.
                torch.cuda.current_stream().synchronize()
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                self._cb_error = e
                return 1

        def allreduce(user, host_buf, count):
            try:
                arr = np.ctypeslib.as_array(host_buf, shape=(count,))
                dev = "cuda" if self._backend == "nccl" else "cpu"
                t = torch.from_numpy(arr.copy()).to(dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                arr[:] = t.cpu().numpy()
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        self._cb_error = None
        self._bcast_cb = _lib.BCAST_FN(bcast)          # keep the thunks alive
        self._allreduce_cb = _lib.ALLREDUCE_FN(allreduce)

    def _push_data(self):
        super()._push_data()
        n = len(self._Y)
        if n != self._staging_n:
            L = _lib.lib()
            nbytes = int(L.gogp_dist_staging_bytes(n))
            self._staging = self._torch.empty(nbytes // 8, dtype=self._torch.float64, device="cuda")
            self._check(L.gogp_dist_setup(
                self._h, self._rank, self._world,
                ctypes.cast(self._bcast_cb, ctypes.c_void_p), ctypes.cast(self._allreduce_cb, ctypes.c_void_p),
                None, ctypes.c_void_p(self._staging.data_ptr()), nbytes))
            self._staging_n = n

    def _check(self, rc):
        if rc != _lib.GOGP_OK and self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise RuntimeError("communication callback failed") from e
        super()._check(rc)
