"""A/B of the convolution paths per (frame size, PSF size): ms per inner iteration with the matrix-core kernels (conv = 2) and with the
transform tiles (conv = 3), blind and non-blind -- what `fft_preferred` (csrc/ics_api.hip) is set from.  Run on the GPU box:
    python scripts/ab_fft.py [size,psf ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import bench  # noqa: E402
from lib import _native  # noqa: E402

cases = [tuple(int(t) for t in a.split(",")) for a in sys.argv[1:]] or [(1024, 17), (1024, 31), (1448, 17), (1448, 31), (2048, 15), (2048, 17), (2048, 21), (2048, 31),
                                                                         (4096, 15), (4096, 17), (4096, 21), (4096, 31), (4096, 45), (4096, 63), (6144, 31)]
ctx = _native.Context.get(0)
for M, K in cases:
    row = []
    for blind in (False, True):
        for conv in (2, 3):
            steps = 50 if M <= 2048 else 20
            try:
                r = bench.timed_run(ctx, M, K, blind, 0, conv, steps, 10)
                row.append("%.4f" % r["ms_per_step"])
            except Exception as exc:   # (conv = 2 above 49 runs as tap blocks; anything refused is shown)
                row.append("n/a")
    print("%5d^2 K=%2d   non-blind matrix %s fft %s   blind matrix %s fft %s" % (M, K, *row), flush=True)
