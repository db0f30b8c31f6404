"""One-image-per-GPU sharding: job assignment and the trivial gather, exercised with
world_size 2 over gloo on CPU (the GPU path uses the same Group over RCCL)."""
import json
import os
import subprocess
import sys
import textwrap

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
