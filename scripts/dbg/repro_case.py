"""stage-by-stage check of one problem against float64 sums: python scripts/dbg/repro_case.py MK M N BLIND SEED"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-cases-studies_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import rl_mm_oracle as orc
from helpers import conv_valid64, corr_full64, gradk64, rel_err
from lib import _native as nv
MK, M, N, blind, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] != "0", int(sys.argv[5])
case = orc.synth_case(M, N, MK, seed=seed, blind=blind)
job = nv.RLJob(M, N, MK); job.upload(case["image"], case["u0"], case["psf0"])
p = job.params(4, M - 4, 4, N - 4, 1e9, 1, 1e-3, 10000.0, blind=blind)
u = case["u0"]; psf = case["psf0"]
for it in range(3):
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p); e = job.read(nv.BUF_ERROR)
    synth = conv_valid64(job.read(nv.BUF_U), job.read(nv.BUF_PSF))
    print(it, "synth", np.max(np.abs(e - (synth - case["image"]))) / np.max(np.abs(synth)), "nan", np.isnan(e).sum())
    job.stage(nv.STAGE_BACKPROJECT, p); g = job.read(nv.BUF_GRADU)
    print(it, "backproject", rel_err(g, corr_full64(e.astype(np.float64), job.read(nv.BUF_PSF))), "nan", np.isnan(g).sum())
    job.stage(nv.STAGE_UPDATE, p)
    print(it, "update nan", np.isnan(job.read(nv.BUF_U)).sum(), job.scalars())
    if blind:
        job.stage(nv.STAGE_SYNTH_RESIDUAL, p); e = job.read(nv.BUF_ERROR)
        job.stage(nv.STAGE_PSF_GRADIENT, p); gk = job.read(nv.BUF_GRADK)
        print(it, "gradk", rel_err(gk, gradk64(job.read(nv.BUF_U).astype(np.float64), e.astype(np.float64))), "nan", np.isnan(gk).sum())
        job.stage(nv.STAGE_PSF_UPDATE, p)
        print(it, "psf nan", np.isnan(job.read(nv.BUF_PSF)).sum())
