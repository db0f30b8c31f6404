"""One image per GPU (BASELINE.json configs[4]).

Every `richardson_lucy_MM` problem is independent, so the multi-GPU story is job sharding: one
process per GPU (torchrun sets RANK / LOCAL_RANK / WORLD_SIZE), each rank owns the jobs
`rank, rank + world, ...`, runs them on its own device and stream, and nothing crosses GPUs during
the iterations.  The only collective is the trivial gather of per-job records (time, iterations,
checksum) at the end -- `torch.distributed` all_gather, i.e. RCCL over xGMI with the "nccl"
backend, gloo on CPU for the tests.  torch is imported lazily and only for this plumbing.
"""
from __future__ import annotations

import os


def world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched directly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard(n_jobs, rank, world_size):
    """Job indices owned by `rank`: round-robin, so that consecutive seeds spread over the GPUs."""
    return list(range(rank, n_jobs, world_size))


class Group:
    """Thin wrapper over torch.distributed for the three things the path needs: barrier, max, gather."""

    def __init__(self, backend=None):
        self.rank, self.local_rank, self.size = world()
        self.dist = None
        self.device = None
        if self.size > 1:
            import torch
            import torch.distributed as dist
            if backend is None:  # ICS_DIST_BACKEND=gloo lets several ranks share one GPU (testing only)
                backend = os.environ.get("ICS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
            else:
                self.device = torch.device("cpu")
            if not dist.is_initialized():
                dist.init_process_group(backend=backend)
            self.dist, self.torch = dist, torch

    def barrier(self):
        if self.dist is not None:
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, values):
        """All-gather a fixed-length list of floats; returns a list (one entry per rank) of lists."""
        if self.dist is None:
            return [list(map(float, values))]
        t = self.torch.tensor(list(map(float, values)), dtype=self.torch.float64, device=self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.size)]
        self.dist.all_gather(out, t)
        return [o.cpu().tolist() for o in out]

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
