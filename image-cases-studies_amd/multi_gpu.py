"""One image per GPU (BASELINE.json configs[4]).

Every `richardson_lucy_MM` problem is independent (the reference is single-process, lib/deconvolution.pyx:341), so the
multi-GPU story is job sharding: one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher -- torchrun or
`bench.py --gpus N` itself), each rank owns the jobs `rank, rank + world, ...`, runs them on its own device and stream,
and nothing crosses GPUs during the iterations.  The only collective is the gather of a small per-rank record (time,
iterations, checksum) at the end:

  * backend "rccl" (default on GPUs): `ics_group_*` of libics_hip.so -- RCCL over xGMI called directly from the library
    (include/ics_hip.h), the unique id travels through a rendezvous file; no PyTorch anywhere in this path;
  * backend "gloo": torch.distributed on CPU -- for the tests in the GPU-less container, and for world-size-2 tests on a
    single-GPU box where both ranks share device 0 (RCCL refuses two ranks on one device).
"""
from __future__ import annotations

import ctypes as C
import os


def world():
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when launched directly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard(n_jobs, rank, world_size):
    """Job indices owned by `rank`: round-robin, so that consecutive seeds spread over the GPUs."""
    return list(range(rank, n_jobs, world_size))


def rendezvous_path():
    """File through which rank 0 hands the RCCL unique id to the other ranks of the node.  All ranks of one launch share
    the parent process (the torchrun agent or bench.py's self-launcher) and the master port."""
    return os.environ.get("ICS_RDZV") or "/tmp/ics_rccl_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())


class Group:
    """barrier / max / gather over the ranks of one node."""

    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.size = world()
        self.backend = "none"
        self._h = None
        self.dist = None
        if self.size == 1:
            return
        backend = backend or os.environ.get("ICS_DIST_BACKEND") or "rccl"
        self.backend = backend
        if backend == "rccl":
            from lib import _native
            lib = _native.load()
            dev = int(os.environ.get("ICS_DEVICE", self.local_rank)) if device is None else int(device)
            h = C.c_void_p()
            rc = lib.ics_group_create(dev, self.rank, self.size, rendezvous_path().encode(), 180, C.byref(h))
            if rc == 0:
                self._h, self._lib, self._check = h, lib, _native._check
            elif os.environ.get("ICS_DIST_FALLBACK", "1") != "0" and os.environ.get("MASTER_ADDR"):
                # communicator creation is collective and fails on every rank alike (e.g. two ranks on one device): carry on with the
                # CPU group rather than lose the run; the record it gathers is 4 doubles per rank
                import sys
                sys.stderr.write("multi_gpu: RCCL group failed on rank %d (%s); falling back to gloo\n" % (self.rank, lib.ics_last_error().decode("utf-8", "replace")))
                backend = self.backend = "gloo (RCCL init failed)"
            else:
                _native._check(rc)
        if backend.startswith("gloo"):
            import torch
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo")
            self.dist, self.torch = dist, torch
        elif backend != "rccl":
            raise ValueError("unknown backend %r (rccl, gloo)" % backend)

    def barrier(self):
        if self._h is not None:
            self._check(self._lib.ics_group_barrier(self._h))
        elif self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        if self._h is not None:
            x = (C.c_double * 1)(float(value))
            self._check(self._lib.ics_group_allreduce_max(self._h, x, 1))
            return float(x[0])
        if self.dist is not None:
            t = self.torch.tensor([float(value)], dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return float(value)

    def gather(self, values):
        """All-gather a fixed-length list of floats (<= 64); returns one list per rank."""
        vals = [float(v) for v in values]
        if self._h is not None:
            n = len(vals)
            send = (C.c_double * n)(*vals)
            recv = (C.c_double * (n * self.size))()
            self._check(self._lib.ics_group_allgather(self._h, send, n, recv))
            return [[recv[r * n + i] for i in range(n)] for r in range(self.size)]
        if self.dist is not None:
            t = self.torch.tensor(vals, dtype=self.torch.float64)
            out = [self.torch.zeros_like(t) for _ in range(self.size)]
            self.dist.all_gather(out, t)
            return [o.tolist() for o in out]
        return [vals]

    def close(self):
        if self._h is not None:
            self._lib.ics_group_destroy(self._h)
            self._h = None
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self.dist = None
