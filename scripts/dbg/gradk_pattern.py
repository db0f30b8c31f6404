import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "image-cases-studies_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import rl_mm_oracle as orc
from helpers import gradk64
from lib import _native as nv
M, N, MK = (int(a) for a in (sys.argv[1:4] or (64, 64, 15)))
case = orc.synth_case(M, N, MK, seed=3, blind=True)
job = nv.RLJob(M, N, MK)
job.upload(case["image"], case["u0"], case["psf0"])
rng = np.random.default_rng(5)
u = (case["u0"] + 0.02 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
job.write(nv.BUF_U, u)
p = job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=True)
job.stage(nv.STAGE_SYNTH_GRADK, p)
e, gk = job.read(nv.BUF_ERROR), job.read(nv.BUF_GRADK)
ref = gradk64(u.astype(np.float64), e.astype(np.float64))
d = np.abs(gk - ref) / np.abs(ref).max()
np.set_printoptions(precision=1, linewidth=250)
print("rel err max", d.max())
print("by tap row a:", d.max(axis=(1, 2)))
print("by tap col b:", d.max(axis=(0, 2)))
print("by channel:", d.max(axis=(0, 1)))
# hypothesis checks: is gk[a] == ref[a+1] or ref[a-1]?
for sh in (-2, -1, 1, 2):
    r2 = np.roll(ref, sh, axis=0)
    print("shift", sh, np.abs(gk - r2).max() / np.abs(ref).max())
