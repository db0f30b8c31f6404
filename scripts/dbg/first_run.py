"""what the FIRST ics_rl_run of a job costs beyond its iterations (deblur_module runs every pyramid level once): python scripts/dbg/first_run.py SIZE PSF BLIND [OUTER]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench
from lib import _native
M, MK, blind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] != "0"
outer = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ctx = _native.Context.get(0)
image, u0, psf_true, psf_uniform = bench.synth_frame(M, M, MK, seed=0)
win = (MK // 2 + 1, min(MK // 2 + 255, M - 1), MK // 2 + 1, min(MK // 2 + 255, M - 1))
for rep in range(3):
    t0 = time.perf_counter()
    job = _native.RLJob(M, M, MK, ctx)
    job.upload(image, u0, psf_uniform if blind else psf_true); ctx.synchronize()
    t1 = time.perf_counter()
    p = job.params(*win, 1e9, outer, 1e-3, 10000.0, blind, 0, 3, stop_test=2)
    job.run(p); ctx.synchronize(); t2 = time.perf_counter()
    job.run(p); ctx.synchronize(); t3 = time.perf_counter()
    job.run(p); ctx.synchronize(); t4 = time.perf_counter()
    job.close()
    print("%d^2 %dx%d blind=%d, %d outer: create + upload %.2f ms | first run %.2f ms | second %.2f | third %.2f  (%.4f ms per inner iteration)" % (M, MK, MK, blind, outer, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t3) * 1e3 / (5 * outer)), flush=True)
