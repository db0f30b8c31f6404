"""One-off differential fuzz: whole richardson_lucy_MM calls on random small problems over every PSF-size range against the pinned oracle."""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
from lib import deconvolution as dc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
for it in range(n):
    MK = int(rng.choice([3, 9, 15, 17, 21, 23, 31, 33, 37, 39, 41, 45, 49, 51, 57, 63, 65, 71, 89, 127]))
    M, N = int(rng.integers(max(8, MK // 3), 200)), int(rng.integers(max(8, MK // 3), 200))
    blind = bool(rng.integers(0, 2))
    case = orc.synth_case(M, N, MK, seed=int(rng.integers(0, 1 << 30)), blind=blind)
    t, l = int(rng.integers(0, max(1, M // 3))), int(rng.integers(0, max(1, N // 3)))
    b, r = int(rng.integers(t + 1, M + 1)), int(rng.integers(l + 1, N + 1))
    args = (t, b, l, r, 1e9, M, N, 3, MK, int(rng.integers(1, 3)), 1e-3, 10000.0)
    u_ref, psf_ref = case["u0"].copy(), case["psf0"].copy()
    with np.errstate(all="ignore"):
        orc.richardson_lucy_MM(case["image"].copy(), u_ref, psf_ref, *args, blind=blind, quiet=True)
    u, psf = case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *args, blind=blind)
    eu = float(np.max(np.abs(u - u_ref)) / np.max(np.abs(u_ref)))
    ep = float(np.max(np.abs(psf - psf_ref)) / np.max(np.abs(psf_ref)))
    worst = max(worst, eu, ep)
    flag = "" if (eu < 1e-4 and ep < 1e-4) else "   <-- FAIL (nan in ref u/psf: %d/%d, in ours: %d/%d; case seed in order)" % (np.isnan(u_ref).sum(), np.isnan(psf_ref).sum(), np.isnan(u).sum(), np.isnan(psf).sum())
    print("MK %3d  %3dx%3d blind=%d win=(%d,%d,%d,%d) it=%d: u %.2e psf %.2e%s" % (MK, M, N, blind, t, b, l, r, args[9], eu, ep, flag))
print("worst", worst)
