"""re-run ONE case of fuzz_runs.py (same generator): python scripts/dbg/fuzz_one.py SEED INDEX [switch=value ...]"""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
from lib import deconvolution as dc
from lib import _native as nv
rng = np.random.default_rng(int(sys.argv[1]))
idx = int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("="); nv.debug_set(k, int(v))
for it in range(idx + 1):
    MK = int(rng.choice([3, 9, 15, 17, 21, 23, 31, 33, 37, 39, 41, 45, 49, 51, 57, 63, 65, 71, 89, 127]))
    M, N = int(rng.integers(max(8, MK // 3), 200)), int(rng.integers(max(8, MK // 3), 200))
    blind = bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1 << 30))
    t, l = int(rng.integers(0, max(1, M // 3))), int(rng.integers(0, max(1, N // 3)))
    b, r = int(rng.integers(t + 1, M + 1)), int(rng.integers(l + 1, N + 1))
    its = int(rng.integers(1, 3))
case = orc.synth_case(M, N, MK, seed=seed, blind=blind)
print("case", MK, M, N, blind, seed, (t, b, l, r), its)
args = (t, b, l, r, 1e9, M, N, 3, MK, its, 1e-3, 10000.0)
u, psf = case["u0"].copy(), case["psf0"].copy()
dc.richardson_lucy_MM(case["image"].copy(), u, psf, *args, blind=blind)
print("nan u/psf", np.isnan(u).sum(), np.isnan(psf).sum())
