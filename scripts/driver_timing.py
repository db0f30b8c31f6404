"""End-to-end time of deconvolve.deblur_module on a synthetic picture: frames on the host between the solver calls vs frames
resident in HBM (device_resident=True).    python scripts/driver_timing.py [size] [blur_width] [iterations]"""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import deconvolve as dv  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 15
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(0)
coarse = rng.random((size // 8 + 2, size // 8 + 2, 3))
pic = (np.repeat(np.repeat(coarse, 8, 0), 8, 1)[:size, :size] * 200 + 20).astype(np.uint8)
kw = dict(mask=[size // 2, size // 2], mask_size=255, display=False, iterations=iters, save=False)
for dev in ((True, True) if os.environ.get("ICS_DRIVER_ONLY_RESIDENT") else (False, True, False, True)):
    with contextlib.redirect_stdout(io.StringIO()):
        t = time.perf_counter()
        out, psf = dv.deblur_module(pic, "t", ".", bw, device_resident=dev, **kw)
        dt = time.perf_counter() - t
    print("%dx%d, blur %d, %d outer iterations, pyramid: frames %s: %.2f s" % (size, size, bw, iters, "in HBM " if dev else "on host", dt))
