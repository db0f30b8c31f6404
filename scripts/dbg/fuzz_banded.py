"""One-off: row-band runs (lib.banded, one process) against the single job over the PSF-size ranges."""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
from lib import deconvolution as dc, banded
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for it in range(n):
    MK = int(rng.choice([9, 15, 23, 31, 33, 45, 51, 63, 65, 89]))
    bands = int(rng.integers(2, 5))
    lo = max(bands * (MK + 8), 60)
    M, N = int(rng.integers(lo, lo + 120)), int(rng.integers(max(24, MK), max(24, MK) + 150))
    blind = bool(rng.integers(0, 2))
    case = orc.synth_case(M, N, MK, seed=int(rng.integers(0, 1 << 30)), blind=blind)
    args = (*orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 2, 1e-3, 10000.0)
    u1, p1 = case["u0"].copy(), case["psf0"].copy()
    u2, p2 = case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u1, p1, *args, blind=blind)
        try:
            banded.richardson_lucy_MM_banded(case["image"].copy(), u2, p2, *args, blind=blind, bands=bands)
        except Exception as ex:
            print("EXC", ex, file=sys.stderr); u2[:] = np.nan
    eu = float(np.max(np.abs(u1 - u2)) / np.max(np.abs(u1))); ep = float(np.max(np.abs(p1 - p2)) / np.max(np.abs(p1)))
    print("MK %3d %3dx%3d bands=%d blind=%d: u %.2e psf %.2e  identical=%s%s" % (MK, M, N, bands, blind, eu, ep, np.array_equal(u1, u2), "" if eu < 1e-5 and ep < 1e-5 else "   <-- FAIL"))
