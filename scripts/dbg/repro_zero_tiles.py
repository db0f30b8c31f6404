import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
from helpers import gradk64, rel_err
from lib import _native as nv
MK, M, N = int(sys.argv[1]), 200, 150
case = orc.synth_case(M, N, MK, seed=5, blind=True)
job = nv.RLJob(M, N, MK); job.upload(case["image"], case["u0"], case["psf0"])
p = job.params(4, M - 4, 4, N - 4, 1e9, 1, 1e-3, 10000.0, blind=True)
u = case["u0"].copy(); u[:int(sys.argv[2])] = 0.0          # the top rows of u all zero: all-zero U tiles at the head of every strip
job.write(nv.BUF_U, u)
rng = np.random.default_rng(1)
e = np.zeros((M, N, 3), np.float32); e[:] = (0.01 * rng.standard_normal((M, N, 3))).astype(np.float32)
job.write(nv.BUF_ERROR, e)
job.stage(nv.STAGE_PSF_GRADIENT, p); gk = job.read(nv.BUF_GRADK)
print("MK", MK, "nan taps", int(np.isnan(gk).any(axis=2).sum()), "rel err", rel_err(np.nan_to_num(gk), gradk64(u.astype(np.float64), e.astype(np.float64))))
