"""One-image-per-GPU sharding: job assignment, the launcher of bench.py, and the trivial gather.

CPU (this container): world_size 2 over gloo with the oracle standing in for the GPU job; `bench.py --gpus N` must refuse
to run on fewer than N GPUs; the native group entry points with world = 1.
GPU (one MI355X): two ranks sharing device 0 drive the PRODUCT path (bench.py -> RLJob) with the gloo group (RCCL refuses
two ranks on one device), and the RCCL plumbing of ics_group_* is exercised with a one-rank communicator."""
import ctypes as C
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_partitions_jobs_exactly_once():
    sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
    import multi_gpu
    for n, w in [(8, 8), (8, 2), (5, 4), (0, 3), (17, 8)]:
        seen = sorted(j for r in range(w) for j in multi_gpu.shard(n, r, w))
        assert seen == list(range(n))
        sizes = [len(multi_gpu.shard(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    assert multi_gpu.world() == (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def test_two_ranks_gloo_barrier_max_gather(tmp_path):
    """torchrun-style launch of 2 CPU ranks: each 'deconvolves' its shard of 5 jobs with the oracle on
    tiny frames, then the group gathers (time, checksum) records exactly like bench.py does."""
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys, time
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np
        import multi_gpu
        import rl_mm_oracle as orc      # test-only checker standing in for the GPU job
        grp = multi_gpu.Group(backend="gloo")
        jobs = multi_gpu.shard(5, grp.rank, grp.size)
        t0 = time.perf_counter(); cks = 0.0
        for j in jobs:
            case = orc.synth_case(24, 20, 3, seed=j)
            u = case["u0"].copy()
            orc.richardson_lucy_MM(case["image"], u, case["psf0"].copy(), 2, 20, 2, 18, 1e9, 24, 20, 3, 3, 1, 1e-3, 1e4, blind=False, quiet=True)
            cks += float(u.sum())
        grp.barrier()
        dt = grp.max(time.perf_counter() - t0)
        rec = grp.gather([float(len(jobs)), cks, float(grp.rank)])
        if grp.rank == 0:
            print("RESULT " + json.dumps({"dt": dt, "rec": rec}))
        grp.close()
    """ % (os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle"))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    rec = res["rec"]
    assert len(rec) == 2 and sorted(int(r[2]) for r in rec) == [0, 1]
    assert sorted(int(r[0]) for r in rec) == [2, 3]      # 5 jobs over 2 ranks
    assert res["dt"] > 0 and all(r[1] > 0 for r in rec)


def test_launch_ranks_stops_the_siblings_when_one_rank_dies(tmp_path, capsys):
    """The supervisor behind `bench.py --gpus N` (round-2 verdict: siblings were not killed, rank 0 sat in the rendezvous):
    rank 1 exits with code 3 at once, ranks 0 and 2 would sleep for a minute -- the launch must end in seconds with code 3,
    the sleepers terminated and rank 1's stderr relayed."""
    import time
    sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
    import multi_gpu
    code = "import os, sys, time\nr = int(os.environ['RANK'])\nassert os.environ['WORLD_SIZE'] == '3' and os.environ['ICS_RDZV']\n" \
           "print('hello from', r, flush=True)\nif r == 1:\n    sys.stderr.write('rank one gives up\\n'); sys.exit(3)\ntime.sleep(60)\n"
    t0 = time.time()
    rc, out0 = multi_gpu.launch_ranks([sys.executable, "-c", code], 3, timeout_s=30, logdir=str(tmp_path))
    assert rc == 3 and time.time() - t0 < 20
    assert "hello from 0" in out0
    err = capsys.readouterr().err
    assert "rank 1 failed" in err and "rank one gives up" in err
    # and the good case: every rank exits 0, rank 0's stdout comes back, nothing on stderr
    rc, out0 = multi_gpu.launch_ranks([sys.executable, "-c", "import os; print('rank', os.environ['RANK'])"], 2, timeout_s=30, logdir=str(tmp_path))
    assert rc == 0 and out0.strip() == "rank 0"
    # time limit
    t0 = time.time()
    rc, _ = multi_gpu.launch_ranks([sys.executable, "-c", "import time; time.sleep(60)"], 2, timeout_s=1.0, logdir=str(tmp_path))
    assert rc == 124 and time.time() - t0 < 20


def test_rccl_group_failure_is_an_error_not_a_silent_gloo_run(monkeypatch):
    """multi_gpu.Group(backend rccl) with a communicator that cannot be built must raise (exit non-zero in bench.py): no
    per-rank fallback to gloo (round-2 advice: ranks fail one by one, a fallback deadlocks the others).  Here: world 2 on a box
    with no (or one) GPU and an unwritable rendezvous path."""
    sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
    import multi_gpu
    from lib import _native
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "5"); monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("ICS_RDZV", "/nonexistent-dir/rdzv"); monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.delenv("ICS_DIST_BACKEND", raising=False)
    with pytest.raises(_native.NativeError):
        multi_gpu.Group()
    assert multi_gpu.rendezvous_path() == "/nonexistent-dir/rdzv"
    monkeypatch.delenv("ICS_RDZV")
    assert str(os.getppid()) in multi_gpu.rendezvous_path() and multi_gpu._process_start_ticks(os.getpid()) > 0


def _ndev():
    sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
    from lib import _native
    return _native.device_count()


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` launches its own ranks; with fewer than N devices it must exit non-zero with a clear
    message, never fall back to fewer GPUs (on the GPU-less container: 0 devices)."""
    n = 9 if _ndev() >= 1 else 2
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode != 0
    assert "--gpus %d requested" % n in out.stderr and "no fallback" in out.stderr
    assert out.stdout.strip() == ""


def test_native_group_single_rank_needs_no_device_and_validates_arguments():
    sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
    from lib import _native
    lib = _native.load()
    h = C.c_void_p()
    assert lib.ics_group_create(0, 0, 1, None, 1, C.byref(h)) == 0          # world = 1: nothing to exchange, no HIP call
    send = (C.c_double * 3)(1.5, -2.0, 7.0)
    recv = (C.c_double * 3)()
    assert lib.ics_group_allgather(h, send, 3, recv) == 0 and list(recv) == [1.5, -2.0, 7.0]
    x = (C.c_double * 1)(4.25)
    assert lib.ics_group_allreduce_max(h, x, 1) == 0 and x[0] == 4.25
    assert lib.ics_group_allreduce_sum(h, x, 1) == 0 and x[0] == 4.25
    be, nr = C.c_int(-1), C.c_int(-1)
    name = C.create_string_buffer(64)
    assert lib.ics_group_describe(h, C.byref(be), C.byref(nr), name, 64) == 0 and (be.value, nr.value, name.value) == (0, 1, b"")
    assert lib.ics_group_barrier(h) == 0
    r, w = C.c_int(-1), C.c_int(-1)
    assert lib.ics_group_info(h, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
    assert lib.ics_group_allgather(h, send, 49153, recv) == _native.ICS_EINVAL      # (the staging buffer holds 49152 doubles: 3 x 127^2 fits)
    lib.ics_group_destroy(h)
    assert lib.ics_group_create(0, 2, 2, b"/tmp/x", 1, C.byref(h)) == _native.ICS_EINVAL       # rank out of range
    assert lib.ics_group_create(0, 1, 2, None, 1, C.byref(h)) == _native.ICS_EINVAL            # no rendezvous path
    assert b"rendezvous" in lib.ics_last_error()


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_through_the_product_path():
    """bench.py under torchrun with 2 ranks: both run the product RLJob on device 0 (ICS_DEVICE=0), group = gloo because
    RCCL refuses two ranks on one device.  Checks the launch contract: n_gpus, per_rank, weak scaling, value = all pixels
    of both ranks over the max-rank time."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ICS_DIST_BACKEND="gloo", ICS_DEVICE="0", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--size", "1024",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and len(d["per_rank"]) == 2 and d["scaling"] == "weak" and d["steps"] == 20
    assert all(r["outer_done"] == 4 for r in d["per_rank"])
    assert abs(d["value"] - 2 * 1024 * 1024 * 20 / (d["ms_per_step"] * 20 * 1e-3) / 1e6) < 0.01 * d["value"]


@pytest.mark.gpu
def test_rccl_plumbing_with_a_one_rank_communicator(tmp_path):
    """ics_group_* over RCCL itself (dlopen, unique id through the rendezvous file, ncclCommInitRank, all-gather,
    all-reduce) with a communicator of one rank -- all a single-GPU box allows.  Runs in a child process."""
    code = textwrap.dedent("""
        import ctypes as C, os, sys
        sys.path.insert(0, %r)
        from lib import _native
        lib = _native.load()
        h = C.c_void_p()
        rc = lib.ics_group_create(0, 0, 1, %r.encode(), 30, C.byref(h))
        assert rc == 0, lib.ics_last_error()
        send = (C.c_double * 4)(1.0, 2.5, -3.0, 4e10); recv = (C.c_double * 4)()
        assert lib.ics_group_allgather(h, send, 4, recv) == 0, lib.ics_last_error()
        assert list(recv) == [1.0, 2.5, -3.0, 4e10]
        x = (C.c_double * 2)(7.0, -1.0)
        assert lib.ics_group_allreduce_max(h, x, 2) == 0 and list(x) == [7.0, -1.0]
        assert lib.ics_group_allreduce_sum(h, x, 2) == 0 and list(x) == [7.0, -1.0]
        be, nr = C.c_int(-1), C.c_int(-1); name = C.create_string_buffer(256)
        assert lib.ics_group_describe(h, C.byref(be), C.byref(nr), name, 256) == 0
        assert (be.value, nr.value) == (1, 1) and b"rccl" in name.value.lower(), (be.value, nr.value, name.value)
        assert lib.ics_group_barrier(h) == 0
        lib.ics_group_destroy(h)
        print("RCCL-OK")
    """ % (os.path.join(ROOT, "image-cases-studies_amd"), str(tmp_path / "rdzv")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ICS_GROUP_FORCE_RCCL="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL-OK" in out.stdout, out.stderr[-3000:]
    assert not (tmp_path / "rdzv").exists()          # rank 0 removes the id file once the communicator exists


@pytest.mark.gpu
@pytest.mark.parametrize("bands", [1, 2])
def test_bench_bands_strong_scaling_line(bands):
    """`bench.py --bands N`: one frame over N ranks (lib.banded.BandRank).  On the single-GPU box: N = 1 in-process, N = 2 as two ranks
    sharing device 0 through the CPU stand-in of the group.  Checks the JSON contract of that line (strong scaling, one frame)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ICS_DIST_BACKEND="gloo", ICS_DEVICE="0", OMP_NUM_THREADS="1")
    cmd = [os.path.join(ROOT, "bench.py"), "--bands", str(bands), "--steps", "10", "--warmup", "5", "--size", "768"]
    if bands > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % bands, "--master-addr", "127.0.0.1", "--master-port", "29547"] + cmd
    else:
        cmd = [sys.executable] + cmd
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == bands and d["scaling"] == "strong" and d["steps"] == 10 and d["unit"] == "MPixels/s/iter"
    assert sum(d["config"]["band_rows"]) == 768 and len(d["config"]["band_rows"]) == bands
    assert abs(d["value"] - 768 * 768 * 10 / (d["ms_per_step"] * 10 * 1e-3) / 1e6) < 0.01 * d["value"]
    assert d["rccl"]["ranks_gathered"] == bands
