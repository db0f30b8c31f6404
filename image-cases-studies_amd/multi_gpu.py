"""One image per GPU (BASELINE.json configs[4]).

Every `richardson_lucy_MM` problem is independent (the reference is single-process, lib/deconvolution.pyx:341), so the
multi-GPU story is job sharding: one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher -- torchrun or
`bench.py --gpus N` itself), each rank owns the jobs `rank, rank + world, ...`, runs them on its own device and stream,
and nothing crosses GPUs during the iterations.  The only collective is the gather of a small per-rank record (time,
iterations, checksum) at the end:

  * backend "rccl" (default on GPUs): `ics_group_*` of libics_hip.so -- RCCL over xGMI called directly from the library
    (include/ics_hip.h), the unique id travels through a rendezvous file; no PyTorch anywhere in this path;
  * backend "gloo" (ICS_DIST_BACKEND=gloo, explicit): torch.distributed on CPU -- for the tests in the GPU-less container, and
    for world-size-2 tests on a single-GPU box where both ranks share device 0 (RCCL refuses two ranks on one device).  There
    is no automatic fallback from one to the other.
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


def _process_start_ticks(pid):
    """start time of a process in clock ticks since boot (/proc/<pid>/stat field 22): distinguishes two launchers that happen
    to get the same pid"""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return 0


def rendezvous_path():
    """File through which rank 0 hands the RCCL unique id to the other ranks of the node.  All ranks of one launch share the
    parent process (the torchrun agent or bench.py's self-launcher) and the master port; the parent's pid AND start time make
    the name unique to the launch, so that the id a failed earlier launch may have left behind is never picked up."""
    if os.environ.get("ICS_RDZV"):
        return os.environ["ICS_RDZV"]
    ppid = os.getppid()
    return "/tmp/ics_rccl_%s_%d_%d" % (os.environ.get("MASTER_PORT", "0"), ppid, _process_start_ticks(ppid))


def launch_ranks(cmd, n, timeout_s=3600.0, logdir=None, extra_env=None):
    """Starts `cmd` n times as FRESH child processes, one rank per GPU (RANK = LOCAL_RANK = 0 .. n-1, a free MASTER_PORT, one
    rendezvous file name for the launch), relays nothing but rank 0's stdout, and supervises them: the first rank that exits
    non-zero -- or the time limit -- ends all the others (a rank that dies before ncclCommInitRank would otherwise leave its
    siblings inside a collective that has no time-out).  Per-rank stderr / stdout go to files in `logdir`; on failure their tails
    are copied to this process's stderr.  Returns (exit code, rank 0's stdout).  The calling process must not have touched HIP and
    is never replaced (no exec)."""
    import socket
    import subprocess
    import sys
    import threading
    import time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    rdzv = "/tmp/ics_rccl_%d_%d_%d" % (port, os.getpid(), int(time.time()))
    logdir = logdir or "/tmp/ics_ranks_%d_%d" % (os.getpid(), int(time.time()))
    os.makedirs(logdir, exist_ok=True)
    procs, files = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ICS_RDZV=rdzv, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        ferr = open(os.path.join(logdir, "rank%d.stderr" % r), "w")
        fout = subprocess.PIPE if r == 0 else open(os.path.join(logdir, "rank%d.stdout" % r), "w")
        files += [ferr] + ([] if r == 0 else [fout])
        procs.append(subprocess.Popen(list(cmd), env=env, stderr=ferr, stdout=fout, text=True))
    out0 = []
    th = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # (a full pipe would block rank 0)
    th.start()
    t0, rc, failed = time.time(), 0, None
    while True:
        codes = [pr.poll() for pr in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0], codes[bad[0]]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > timeout_s:
            failed, rc = -1, 124
            break
        time.sleep(0.1)
    if failed is not None:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
    th.join(timeout=10)
    for f in files:
        f.close()
    try:
        os.unlink(rdzv)
    except OSError:
        pass
    if failed is not None:
        sys.stderr.write("launch_ranks: %s (exit code %d); the other ranks were stopped.  Per-rank logs in %s\n" %
                         ("rank %d failed" % failed if failed >= 0 else "time-out after %.0f s" % timeout_s, rc, logdir))
        for r in range(n):
            try:
                tail = open(os.path.join(logdir, "rank%d.stderr" % r)).read()[-1500:]
            except OSError:
                tail = ""
            if tail.strip():
                sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, tail))
    return rc, (out0[0] if out0 else "")


class Group:
    """barrier / max / sum / gather over the ranks of one node.

    The backend is decided UP FRONT and identically on every rank -- "rccl" unless ICS_DIST_BACKEND (or the argument) says
    "gloo" -- and a failure to build the RCCL communicator is an error on the rank that sees it: ranks fail one by one
    (a wrong device index here, a rendezvous time-out there), so a per-rank fallback would leave some ranks in gloo's
    rendezvous and the others in ncclCommInitRank for ever.  "gloo" (torch.distributed on CPU) exists for the GPU-less test
    container and for two ranks sharing one GPU on a single-GPU box, which RCCL refuses."""

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
            _native._check(lib.ics_group_create(dev, self.rank, self.size, rendezvous_path().encode(), 180, C.byref(h)))
            self._h, self._lib, self._check = h, lib, _native._check
        elif backend == "gloo":
            import torch
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo")
            self.dist, self.torch = dist, torch
        else:
            raise ValueError("unknown backend %r (rccl, gloo)" % backend)

    def describe(self):
        """{"backend", "world", "ranks_in_communicator", "lib"}: what actually carries the collectives (for the bench JSON)."""
        if self._h is not None:
            be, n = C.c_int(0), C.c_int(0)
            name = C.create_string_buffer(256)
            self._check(self._lib.ics_group_describe(self._h, C.byref(be), C.byref(n), name, 256))
            return {"backend": "rccl" if be.value == 1 else "local", "world": self.size, "ranks_in_communicator": n.value, "lib": name.value.decode()}
        if self.dist is not None:
            return {"backend": "gloo (torch.distributed, CPU)", "world": self.size, "ranks_in_communicator": self.dist.get_world_size(), "lib": "torch"}
        return {"backend": "none", "world": 1, "ranks_in_communicator": 1, "lib": ""}

    def barrier(self):
        if self._h is not None:
            self._check(self._lib.ics_group_barrier(self._h))
        elif self.dist is not None:
            self.dist.barrier()

    def _allreduce(self, values, op):
        vals = [float(v) for v in values]
        if self._h is not None:
            fn = self._lib.ics_group_allreduce_max if op == "max" else self._lib.ics_group_allreduce_sum
            out = []
            for i in range(0, len(vals), 49152):                 # ICS_GROUP_MAX_COUNT doubles per call (3 x 127^2 fits in one)
                chunk = vals[i:i + 49152]
                x = (C.c_double * len(chunk))(*chunk)
                self._check(fn(self._h, x, len(chunk)))
                out.extend(x)
            return out
        if self.dist is not None:
            t = self.torch.tensor(vals, dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
            return t.tolist()
        return vals

    def max(self, value):
        return self._allreduce([value], "max")[0]

    def max_many(self, values):
        return self._allreduce(values, "max")

    def sum_many(self, values):
        """element-wise float64 sum over the ranks (the 3 K^2 PSF-gradient partial sums of the row-band split)"""
        return self._allreduce(values, "sum")

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

    def reduce_band_keys(self, job):
        """max over the ranks of the six step-size keys of a band job (ICS_BUF_RED [0..5]).  RCCL group (or one local rank): in place on
        the device buffer, on the job's stream, no host round trip (ics_rl_allreduce_keys); CPU stand-in: through the host."""
        if self._h is not None:
            self._check(self._lib.ics_rl_allreduce_keys(job._h, self._h))
            return
        if self.dist is None:
            return
        import numpy as np
        keys = self.max_many([float(k) for k in job.red_keys()[:6]])        # (order-preserving uint32 keys are exact in float64)
        job.set_red_keys(np.array(keys, np.float64).astype(np.uint32))

    def reduce_band_gradk(self, job):
        """sum over the ranks of a band job's PSF-gradient (ICS_BUF_GRADK), float64 across the ranks, rounded to float32 once
        (ics_rl_allreduce_gradk: one in-place RCCL call on the device); CPU stand-in: through the host."""
        if self._h is not None:
            self._check(self._lib.ics_rl_allreduce_gradk(job._h, self._h))
            return
        if self.dist is None:
            return
        import numpy as np
        from lib import _native
        gk = job.read(_native.BUF_GRADK)
        tot = np.array(self.sum_many(gk.astype(np.float64).ravel()), np.float64)
        job.write(_native.BUF_GRADK, tot.astype(np.float32).reshape(gk.shape))

    def exchange_rows(self, job, which, send=None, recv=None):
        """Point-to-point rows of frame buffer `which` between band jobs on different ranks (lib/banded.py rank mode):
        send = (row0, nrows, peer) leaves this rank's job, recv = (row0, nrows, peer) arrives in it; either may be None.  RCCL
        backend: ncclSend / ncclRecv between the device frames (ics_rl_exchange_rows); gloo backend (CPU stand-in, ranks sharing a
        GPU): through the host."""
        s0, sn, sp = send if send is not None else (0, 0, -1)
        r0, rn, rp = recv if recv is not None else (0, 0, -1)
        if self._h is not None:
            self._check(self._lib.ics_rl_exchange_rows(job._h, self._h, int(which), int(s0), int(sn), int(sp), int(r0), int(rn), int(rp)))
            return
        if self.dist is None:
            raise RuntimeError("exchange_rows needs a group of more than one rank")
        reqs, buf = [], None
        if sp >= 0:
            reqs.append(self.dist.isend(self.torch.from_numpy(job.read_rows(which, s0, sn).copy()), dst=sp))
        if rp >= 0:
            import numpy as np
            buf = np.empty((rn,) + job._shape(which)[1:], np.float32)
            reqs.append(self.dist.irecv(self.torch.from_numpy(buf), src=rp))
        for q in reqs:
            q.wait()
        if buf is not None:
            job.write_rows(which, r0, buf)

    def close(self):
        if self._h is not None:
            self._lib.ics_group_destroy(self._h)
            self._h = None
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self.dist = None
