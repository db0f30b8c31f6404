"""Wall time of a whole drop-in call lib.deconvolution.richardson_lucy_MM (upload of image / u / psf from numpy arrays, the run, download into the caller's
strided u) against the device time of the run itself:  python scripts/call_overhead.py [size] [psf] [outer iterations]"""
import contextlib, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd")); sys.path.insert(0, ROOT)
import bench
from lib import deconvolution as dc
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 15
it = int(sys.argv[3]) if len(sys.argv) > 3 else 10
image, u0, psf_true, psf_uniform = bench.synth_frame(S, S, K, 0)
pad = K // 2
win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)
for blind in (False, True):
    for rep in range(3):
        u, psf = u0.copy(), (psf_uniform if blind else psf_true).copy()
        t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            dc.richardson_lucy_MM(image, u, psf, *win, 1e9, S, S, 3, K, it, 1e-3, 1e4, blind=blind)
        wall = time.perf_counter() - t
        st = dc.richardson_lucy_MM.last
        print("%dx%d k%d blind=%d, %d outer: call %.1f ms, device run %.1f ms, outside the run %.1f ms (%.0f %%)" % (S, S, K, blind, it, wall * 1e3, st.ms_total, wall * 1e3 - st.ms_total, 100 * (1 - st.ms_total / (wall * 1e3))))
