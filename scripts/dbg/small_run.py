"""a plain run of a small frame, for rocprofv3: python scripts/dbg/small_run.py M MK blind outer"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench
from lib import _native
M, MK, blind, outer = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] != "0", int(sys.argv[4])
ctx = _native.Context.get(0)
image, u0, psf_true, psf_uniform = bench.synth_frame(M, M, MK, seed=3)
pad = MK // 2
win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)
j = _native.RLJob(M, M, MK, ctx)
j.upload(image, u0, psf_uniform if blind else psf_true)
p = j.params(*win, 1e9, outer, 1e-3, 10000.0, blind, 0, 3, stop_test=2)
j.run(p); ctx.synchronize()
t0 = time.perf_counter(); j.run(p); ctx.synchronize()
print("%.4f ms per inner iteration" % ((time.perf_counter() - t0) * 1e3 / (5 * outer)))
