"""driver-style short run (--steps 20 --warmup 5) repeated in one process: does the first timed region differ from later ones?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench
from lib import _native
ctx = _native.Context.get(0)
M, MK = 4096, 15
image, u0, psf_true, psf_uniform = bench.synth_frame(M, M, MK, seed=0)
job = _native.RLJob(M, M, MK, ctx)
job.upload(image, u0, psf_uniform)
pad = MK // 2
win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)
def run(n, profile=0):
    return job.run(job.params(*win, 1e9, n // 5, 1e-3, 10000.0, True, 0, 3, stop_test=2, profile=profile))
t_idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for rep in range(6):
    if t_idle: time.sleep(t_idle)
    run(5); ctx.synchronize()
    t0 = time.perf_counter(); st = run(20, 4); ctx.synchronize(); el = time.perf_counter() - t0
    names = _native.KERNEL_NAMES
    print("rep %d: %.4f ms/step (device %.4f)  " % (rep, el * 1e3 / 20, st.ms_total / 20), {names[k]: round(st.ms_kernel[k], 4) for k in range(12) if st.launches[k]})
t0 = time.perf_counter(); run(200); ctx.synchronize(); print("200 steps: %.4f ms/step" % ((time.perf_counter() - t0) * 1e3 / 200))
