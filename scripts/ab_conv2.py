"""A/B of mode 2 of the transform tiles (A1 + A3 as one unit per tile pair, k_conv_fft<2>) against the two kernels, per (frame size, PSF size):
ms per inner iteration with conv = ICS_CONV_FFT, debug switch fft_conv2 = 0 / 2 -- what ICS_CONV2_MAX_K (csrc/ics_api.hip) is set from.
Run on the GPU box:    python scripts/ab_conv2.py [size,psf ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench  # noqa: E402
from lib import _native  # noqa: E402

cases = [tuple(int(t) for t in a.split(",")) for a in sys.argv[1:]] or [(S, K) for S in (1024, 2048, 4096) for K in (9, 13, 15, 17, 19, 21, 23, 25, 31)]
ctx = _native.Context.get(0)
for M, K in cases:
    row = []
    for blind in (False, True):
        for sw in (0, 2):
            _native.debug_set("fft_conv2", sw)
            steps = 50 if M <= 2048 else 20
            try:
                r = bench.timed_run(ctx, M, K, blind, 0, 3, steps, 10)
                row.append("%.4f" % r["ms_per_step"])
            except Exception as exc:
                row.append("n/a")
    print("%5d^2 K=%2d   non-blind two kernels %s one unit %s   blind two kernels %s one unit %s" % (M, K, *row), flush=True)
