"""One-off: lib.utils blurs / USM / bilateral on random shapes and windows against oracle/utils_oracle.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import utils_oracle as uo
from lib import utils
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    H, W = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    src = rng.random((H, W))
    kind = int(rng.integers(0, 4))
    try:
        if kind == 0:
            size, sig = int(rng.integers(1, 40)), float(rng.uniform(0.3, 6)); a, b, name = utils.gaussian_blur(src, size, sig), uo.gaussian_blur(src, size, sig), "gauss %d %.2f" % (size, sig)
        elif kind == 1:
            size, al = int(rng.integers(1, 40)), float(rng.uniform(0.5, 8)); a, b, name = utils.bessel_blur(src, size, al), uo.bessel_blur(src, size, al), "bessel %d %.2f" % (size, al)
        elif kind == 2:
            size, sig, amt = int(rng.integers(1, 30)), float(rng.uniform(0.3, 5)), float(rng.uniform(0, 2)); a, b, name = utils.USM(src, size, sig, amt), uo.USM(src, size, sig, amt), "usm %d" % size
        else:
            r, si, ss = int(rng.integers(0, 7)), float(rng.uniform(0.05, 1)), float(rng.uniform(0.5, 4)); a, b, name = utils.bilateral_filter(src, r, si, ss), uo.bilateral_filter(src, r, si, ss), "bilateral %d" % r
    except Exception as ex:
        print("EXC %dx%d kind %d: %s" % (H, W, kind, ex)); continue
    err = float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) if a.shape == b.shape else float("inf")
    worst = max(worst, err)
    print("%3dx%3d %-18s err %.2e%s" % (H, W, name, err, "" if err < 1e-10 else "   <-- FAIL"))
print("worst", worst)
