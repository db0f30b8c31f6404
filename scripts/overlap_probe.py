"""Probe: how well does the memory-bound update kernel overlap with the VALU-bound convolution kernel when they
run on two streams (two jobs, two host threads)?  Upper bound for a band-pipelined update||synth schedule."""
import sys, time, threading
sys.path.insert(0, "/root/repo/image-cases-studies_amd"); sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from lib import _native as nv
M = N = 4096; MK = 15
image, u0, pt, pu = bench.synth_frame(M, N, MK, 0)
jobs = []
for i in range(2):
    ctx = nv.Context(0); job = nv.RLJob(M, N, MK, ctx); job.upload(image, u0, pt); job.write(nv.BUF_UT, u0); jobs.append((ctx, job))
p = jobs[0][1].params(8, 247, 8, 247, 1e9, 1, 1e-3, 1e4, False)
for ctx, job in jobs:
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p); job.stage(nv.STAGE_BACKPROJECT, p)
def loop(job, stage, n):
    for _ in range(n): job.stage(stage, p)
n = 100
def timed(pairs):
    th = [threading.Thread(target=loop, args=(j, s, n)) for j, s in pairs]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
    return (time.perf_counter() - t0) * 1e3 / n
a = timed([(jobs[0][1], nv.STAGE_SYNTH_RESIDUAL)])
b = timed([(jobs[1][1], nv.STAGE_UPDATE)])
c = timed([(jobs[0][1], nv.STAGE_SYNTH_RESIDUAL), (jobs[1][1], nv.STAGE_UPDATE)])
d = timed([(jobs[0][1], nv.STAGE_SYNTH_RESIDUAL), (jobs[1][1], nv.STAGE_SYNTH_RESIDUAL)])
print("per call (ms, incl. host sync): synth alone %.3f, update alone %.3f, synth||update %.3f (sum %.3f), synth||synth %.3f" % (a, b, c, a + b, d))
