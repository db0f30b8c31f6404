"""ms per stage call (kernel time through HIP events is not exposed per stage: wall time of N back-to-back stage calls, each synchronous)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench
from lib import _native as nv
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
MK = int(sys.argv[2]) if len(sys.argv) > 2 else 15
blind = (sys.argv[3] != "0") if len(sys.argv) > 3 else False
tv = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ctx = nv.Context.get(0)
image, u0, psf_true, psf_uniform = bench.synth_frame(M, M, MK, seed=0)
job = nv.RLJob(M, M, MK, ctx)
job.upload(image, u0, psf_true)
pad = MK // 2
win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)
p = job.params(*win, 1e9, 1, 1e-3, 10000.0, blind, tv_mode=tv)
job.run(job.params(*win, 1e9, 2, 1e-3, 10000.0, blind, tv_mode=tv, stop_test=2))
stages = [("synth", nv.STAGE_SYNTH_RESIDUAL)] + ([("tvterm", nv.STAGE_TVTERM)] if tv else []) + [("backproject", nv.STAGE_BACKPROJECT), ("update", nv.STAGE_UPDATE)]
for name, st in stages:
    for _ in range(5): job.stage(st, p)
    t0 = time.perf_counter()
    n = 50
    for _ in range(n): job.stage(st, p)
    print("%-12s %.4f ms per call (wall, incl. ~0.01 ms of call + sync overhead)" % (name, (time.perf_counter() - t0) * 1e3 / n))
