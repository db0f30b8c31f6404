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
big = len(sys.argv) > 3 and sys.argv[3] == "big"      # PSF sizes 129 ... 255 (tap blocks only)
only = [int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x]      # re-run these problems of the sequence only ...
f64 = os.environ.get("FUZZ_F64") == "1"                                          # ... and place both results against float64 convolutions
conv = int(os.environ.get("FUZZ_CONV", "0"))                                      # ics_rl_params.conv of the device runs (3: transform tiles; FUZZ_WIDE=1: PSF sizes 67 ... 255 there)
smax = int(os.environ.get("FUZZ_MAX", "200"))                                     # largest frame side
worst = 0.0
nfail = nrefnan = 0
for it in range(n):
    MK = int(rng.choice([3, 9, 15, 17, 21, 23, 31, 33, 37, 39, 41, 45, 49, 51, 57, 63, 65, 71, 89, 127]))
    if big:
        MK = int(rng.choice([129, 131, 133, 145, 165, 167, 199, 231, 253, 255]))
    if conv == 3:
        MK = int(rng.choice([3, 5, 9, 15, 17, 19, 21, 23, 27, 31, 33, 37, 45, 49, 51, 57, 63, 65]))
        if os.environ.get("FUZZ_WIDE") == "1":      # round 6: one tile to 85, tap blocks on the tiles to 255
            MK = int(rng.choice([67, 71, 79, 85, 87, 89, 97, 101, 127, 129, 161, 193, 201, 255]))
    if os.environ.get("FUZZ_SMALL") == "1":        # round 6: the cooperative small-frame kernel (ics_small.hip): PSF sizes 3 ... 31, frames to 288 px a side
        MK = int(rng.choice(np.arange(3, 33, 2)))
    M, N = int(rng.integers(max(8, MK // 3), smax)), int(rng.integers(max(8, MK // 3), smax))
    blind = bool(rng.integers(0, 2))
    case = orc.synth_case(M, N, MK, seed=int(rng.integers(0, 1 << 30)), blind=blind)
    t, l = int(rng.integers(0, max(1, M // 3))), int(rng.integers(0, max(1, N // 3)))
    b, r = int(rng.integers(t + 1, M + 1)), int(rng.integers(l + 1, N + 1))
    args = (t, b, l, r, 1e9, M, N, 3, MK, int(rng.integers(1, 3)), 1e-3, 10000.0)
    # data variants: black bands (all-zero tiles), small / large values, a flat image
    var = int(rng.integers(0, 6))
    pad = MK // 2
    if var == 1:
        k = int(rng.integers(1, max(2, M // 2))); case["image"][:k] = 0; case["u0"][:k + pad] = 0
    elif var == 2:
        k = int(rng.integers(1, max(2, N // 2))); case["image"][:, :k] = 0; case["u0"][:, :k + pad] = 0
    elif var == 3:
        sc = np.float32(10.0 ** float(rng.integers(-6, 5))); case["image"] *= sc; case["u0"] *= sc
    elif var == 4:
        case["image"][:] = np.float32(0.37); case["u0"][:] = np.float32(0.37)
    elif var == 5:
        k = int(rng.integers(1, max(2, M // 2))); case["image"][-k:] = 0; case["u0"][-(k + pad):] = 0; case["image"] *= np.float32(0.3); case["u0"] *= np.float32(0.3)
    if only and it not in only:
        continue
    u_ref, psf_ref = case["u0"].copy(), case["psf0"].copy()
    with np.errstate(all="ignore"):
        orc.richardson_lucy_MM(case["image"].copy(), u_ref, psf_ref, *args, blind=blind, quiet=True)
    if f64:
        # the same loop with float64 convolutions (a float64 FFT: exact to ~1e-15 of the largest value, seconds at any PSF size)
        from scipy.signal import fftconvolve
        keep = orc._conv_direct
        orc._conv_direct = lambda a, b, mode: fftconvolve(np.asarray(a, np.float64), np.asarray(b, np.float64), mode=mode)
        u64, psf64 = case["u0"].copy(), case["psf0"].copy()
        with np.errstate(all="ignore"):
            orc.richardson_lucy_MM(case["image"].copy(), u64, psf64, *args, blind=blind, quiet=True, conv="direct")
        orc._conv_direct = keep
    u, psf = case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *args, blind=blind, conv=conv)
    if np.isnan(u).any() or np.isnan(psf).any():              # finite inputs: this library never returns NaN (black regions: the 0/0 rule, ics_hip.h)
        print("MK %3d  %3dx%3d blind=%d var=%d: NaN in the device result (%d / %d)   <-- FAIL" % (MK, M, N, blind, var, np.isnan(u).sum(), np.isnan(psf).sum()))
        nfail += 1
        continue
    if np.isnan(u_ref).any() or np.isnan(psf_ref).any():
        # the reference met an exact zero in its FFT noise inside a black region (common for black columns, rare for black rows) and lost
        # the frame; compare with the float64-direct oracle, which carries the rule, instead
        nrefnan += 1
        if big or M * N * MK * MK > 4e8:      # (float64 direct sums: 65 025 taps per value at the big PSF sizes, or a megapixel frame -- minutes per case in numpy)
            print("   (reference = NaN by an exact zero of its FFT noise; not compared at this PSF size)")
            continue
        u_ref, psf_ref = case["u0"].copy(), case["psf0"].copy()
        orc.richardson_lucy_MM(case["image"].copy(), u_ref, psf_ref, *args, blind=blind, quiet=True, conv="direct")
        print("   (reference = NaN by an exact zero of its FFT noise; compared with the float64-direct oracle)")
    eu = float(np.max(np.abs(u - u_ref)) / max(np.max(np.abs(u_ref)), 1e-30))
    ep = float(np.max(np.abs(psf - psf_ref)) / max(np.max(np.abs(psf_ref)), 1e-30))
    if var != 4: worst = max(worst, eu, ep)
    gate = 5e-3 if var == 4 else 1e-4      # (a flat image: the residual is rounding noise, amplified by lambd = 1e4 on both sides)
    nfail += not (eu < gate and ep < gate)
    flag = "" if (eu < gate and ep < gate) else "   <-- FAIL (nan in ref u/psf: %d/%d, in ours: %d/%d; case seed in order)" % (np.isnan(u_ref).sum(), np.isnan(psf_ref).sum(), np.isnan(u).sum(), np.isnan(psf).sum())
    print("MK %3d  %3dx%3d blind=%d win=(%d,%d,%d,%d) it=%d var=%d: u %.2e psf %.2e%s" % (MK, M, N, blind, t, b, l, r, args[9], var, eu, ep, flag))
    if f64:
        d = lambda x, y: float(np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-30))
        print("      against float64 convolutions: device u %.2e psf %.2e | reference (complex64 FFT) u %.2e psf %.2e" % (d(u, u64), d(psf, psf64), d(u_ref, u64), d(psf_ref, psf64)))
print("worst", worst, "failures", nfail, "reference-NaN cases", nrefnan)
