"""A/B of a debug switch inside one process, alternating, sustained runs: python scripts/dbg/ab_switch.py planar_image 0 1 [size psf blind]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench
from lib import _native
name, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
MK = int(sys.argv[5]) if len(sys.argv) > 5 else 15
blind = (sys.argv[6] != "0") if len(sys.argv) > 6 else True
tv = int(sys.argv[7]) if len(sys.argv) > 7 else 0
ctx = _native.Context.get(0)
image, u0, psf_true, psf_uniform = bench.synth_frame(M, M, MK, seed=0)
pad = MK // 2
win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)
jobs = {}
for v in (va, vb):
    _native.debug_set(name, v)
    jobs[v] = _native.RLJob(M, M, MK, ctx)
    jobs[v].upload(image, u0, psf_uniform if blind else psf_true)
def run(v, n, profile=0):
    _native.debug_set(name, v)
    j = jobs[v]
    return j.run(j.params(*win, 1e9, n // 5, 1e-3, 10000.0, blind, 0, 3, stop_test=2, profile=profile, tv_mode=tv))
run(va, 50); run(vb, 50); ctx.synchronize()
names = _native.KERNEL_NAMES
for rep in range(3):
    for v in (va, vb):
        t0 = time.perf_counter(); run(v, 200); ctx.synchronize(); el = time.perf_counter() - t0
        st = run(v, 40, 4); ctx.synchronize()
        print("%s=%d rep %d: %.4f ms/step  " % (name, v, rep, el * 1e3 / 200), {names[k]: round(st.ms_kernel[k], 4) for k in range(12) if st.launches[k]})
