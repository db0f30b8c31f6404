"""Timing of the lib/utils.py filters on one 4096 x 4096 float64 channel (GPU box): device time of the kernels (HIP events,
ics_ctx_last_kernel_ms), wall time of the call including the two PCIe transfers of the 134 MB channel, and scipy's own
convolve2d on the host for the blur."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
from lib import _native, utils
ctx = _native.Context.get()
rng = np.random.default_rng(0)
src = rng.random((4096, 4096))
for name, fn in (("gaussian_blur(r=15, 2.5)", lambda: utils.gaussian_blur(src, 15, 2.5)), ("bessel_blur(r=31, 4.0)", lambda: utils.bessel_blur(src, 31, 4.0)),
                 ("USM(r=15, bessel)", lambda: utils.USM(src, 15, 3.0, 0.7)), ("bilateral(r=5)", lambda: utils.bilateral_filter(src, 5, 0.1, 2.0)),
                 ("bilateral(r=10)", lambda: utils.bilateral_filter(src, 10, 0.1, 4.0))):
    fn()
    t0 = time.perf_counter(); fn(); wall = time.perf_counter() - t0
    print("%-28s device %.3f ms   call incl. transfers %.1f ms" % (name, ctx.last_kernel_ms(), wall * 1e3))
from scipy.signal import convolve2d
t0 = time.perf_counter(); convolve2d(src, utils.gaussian_kernel(15, 2.5), mode="same", boundary="symm"); print("scipy convolve2d 15x15 on the host: %.0f ms" % ((time.perf_counter() - t0) * 1e3))
