"""One-off differential fuzz of the build-defined TV modes (1 active MM-TV, 2 PAM isotropic) against oracle/rl_ext_oracle.py."""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
import rl_ext_oracle as ext
from lib import deconvolution as dc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
conv = int(os.environ.get("FUZZ_CONV", "0"))      # 3: the PAM kind with its convolutions on the transform tiles (mode 2 only, PSF sizes <= 65)
worst = 0.0
for it in range(n):
    MK = int(rng.choice([3, 5, 9, 15, 17, 21, 23, 31, 33, 37, 39, 45, 49, 51, 63]))
    M, N = int(rng.integers(2 * MK + 8, 2 * MK + 180)), int(rng.integers(2 * MK + 8, 2 * MK + 180))   # (default_window needs 2 pad + 3 rows)
    blind = bool(rng.integers(0, 2))
    mode = int(rng.choice([1, 2]))
    if conv == 3:
        mode = 2
    lambd = float(rng.choice([50.0, 200.0, 1e4]))
    case = orc.synth_case(M, N, MK, seed=int(rng.integers(0, 1 << 30)), blind=blind)
    args = (*orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 2, 1e-3, lambd)
    img_r, u_r, psf_r = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    with np.errstate(all="ignore"):
        if mode == 1: ext.richardson_lucy_MM_tv(img_r, u_r, psf_r, *args, blind=blind)
        else: ext.richardson_lucy_PAM(img_r, u_r, psf_r, *args, blind=blind, collaborative=False)
    img, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(img, u, psf, *args, blind=blind, tv_mode=mode, conv=conv)
    eu = float(np.max(np.abs(u - u_r)) / np.max(np.abs(u_r))); ep = float(np.max(np.abs(psf - psf_r)) / np.max(np.abs(psf_r)))
    worst = max(worst, eu, ep)
    # non-blind (epsilon = 1e-6) at wide PSFs: the TV term of nearly flat pixels turns on ANY convolution rounding -- the oracle's own FFT and
    # direct forms deviate from each other by the same 1e-4 ... 1e-3 (tests/test_tv_mode.py, last test): reported, not counted
    ill = (not blind) and MK >= 33
    ok = (eu < 1e-4 and ep < 1e-4) or (ill and eu < 5e-3)
    print("mode %d MK %3d %3dx%3d blind=%d lambd=%g: u %.2e psf %.2e%s" % (mode, MK, M, N, blind, lambd, eu, ep, "" if ok and eu < 1e-4 else ("   (ill-conditioned mode at this PSF size)" if ok else "   <-- FAIL")))
print("worst", worst)
