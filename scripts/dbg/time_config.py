"""ms per inner iteration and per kernel class of one configuration: python scripts/dbg/time_config.py SIZE PSF BLIND [STEPS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench
from lib import _native
M, MK, blind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] != "0"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ctx = _native.Context.get(0)
image, u0, psf_true, psf_uniform = bench.synth_frame(M, M, MK, seed=0)
job = _native.RLJob(M, M, MK, ctx)
job.upload(image, u0, psf_uniform if blind else psf_true)
win = (MK // 2 + 1, min(MK // 2 + 255, M - 1), MK // 2 + 1, min(MK // 2 + 255, M - 1))
def run(n, profile=0):
    return job.run(job.params(*win, 1e9, n // 5, 1e-3, 10000.0, blind, 0, 3, stop_test=2, profile=profile, conv=int(os.environ.get('ICS_TC_CONV', '0'))))
run(5); ctx.synchronize()
prof = int(os.environ.get("ICS_TC_PROFILE", "1"))
t0 = time.perf_counter(); st = run(steps, prof); ctx.synchronize(); el = time.perf_counter() - t0
names = _native.KERNEL_NAMES
print("%d^2, %dx%d, blind=%d: %.3f ms/step  " % (M, MK, MK, blind, el * 1e3 / steps), {names[k]: round(st.ms_kernel[k], 3) for k in range(12) if st.launches[k]})
