import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
from lib import _native as nv
MK, M, N, seed = 89, 157, 156, 676839450
case = orc.synth_case(M, N, MK, seed=seed, blind=True)
job = nv.RLJob(M, N, MK); job.upload(case["image"], case["u0"], case["psf0"])
p = job.params(4, M - 4, 4, N - 4, 1e9, 1, 1e-3, 10000.0, blind=True)
job.stage(nv.STAGE_SYNTH_RESIDUAL, p); job.stage(nv.STAGE_BACKPROJECT, p); job.stage(nv.STAGE_UPDATE, p); job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
job.stage(nv.STAGE_PSF_GRADIENT, p); gk = job.read(nv.BUF_GRADK)
nanmap = np.isnan(gk).any(axis=2)
print("nan taps", nanmap.sum())
rows = np.where(nanmap.any(axis=1))[0]; cols = np.where(nanmap.any(axis=0))[0]
print("rows", rows.min() if len(rows) else None, rows.max() if len(rows) else None, "cols", cols.min() if len(cols) else None, cols.max() if len(cols) else None)
for a0 in (0, 31, 62):
    for b0 in (0, 31, 62):
        print("block", a0, b0, int(nanmap[a0:a0+31, b0:b0+31].sum()), "of", nanmap[a0:a0+31, b0:b0+31].size)
u = job.read(nv.BUF_U); e = job.read(nv.BUF_ERROR)
print("u max", np.abs(u).max(), "e max", np.abs(e).max(), "e nan", np.isnan(e).sum())
