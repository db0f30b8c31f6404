"""tv_mode 2 with conv = 1 (fp32 HWC kernels) against conv = 3 (planar mirrors + transform tiles): where and how far they differ (the frame-wide 1e-3 of the one-tile-column decode bug showed up here)."""
import contextlib, io, os, sys
import numpy as np
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
from lib import deconvolution as dc
for (M, N, MK, blind, lambd) in [(160, 109, 17, True, 200.0), (151, 115, 3, True, 50.0), (147, 172, 17, True, 50.0), (144, 99, 9, False, 1e4)]:
    case = orc.synth_case(M, N, MK, seed=7, blind=blind)
    for iters in (1, 2):
        args = (*orc.default_window(M, N, MK), 1e9, M, N, 3, MK, iters, 1e-3, lambd)
        res = {}
        for conv in (1, 3):
            img, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
            with contextlib.redirect_stdout(io.StringIO()):
                dc.richardson_lucy_MM(img, u, psf, *args, blind=blind, tv_mode=2, conv=conv)
            res[conv] = u
        d = np.abs(res[3] - res[1]).max(axis=2)
        y, x = np.unravel_index(np.argmax(d), d.shape)
        rows = np.where(d.max(axis=1) > 1e-5)[0]; cols = np.where(d.max(axis=0) > 1e-5)[0]
        print(M, N, MK, blind, "iters", iters, "max", d.max(), "at", (y, x), "of", d.shape, "rows", (rows.min(), rows.max(), len(rows)) if len(rows) else None, "cols", (cols.min(), cols.max(), len(cols)) if len(cols) else None)
