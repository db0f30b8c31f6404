#!/usr/bin/env python3
"""Blind golden at 576 x 520 x 3, 15 x 15 PSF, 2 outer iterations, from the COMPILED REFERENCE (build container only).

TEST INFRASTRUCTURE ONLY (data, no reference source).  Purpose: a reference trajectory on a frame of 9 x 9 tiles, run on
the GPU with the test hook ICS_TEST_MAX_WGS=8 so that every persistent workgroup walks 8-11 tiles (next-tile prefetch,
band split, interior-origin grid of the matrix-core kernels) -- the 129^2 goldens have at most 3 x 3 tiles.  The inputs
come from orc.synth_case(seed); the outputs are stored as crops, every 48th row, float64 moments, the full PSF and the
per-outer-iteration scalars (taken from the numpy oracle after asserting it equals the reference bit for bit)."""
import json
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_reference  # noqa: E402
import make_golden as mg  # noqa: E402
import rl_mm_oracle as orc  # noqa: E402


def main():
    ref = build_reference.load()
    M, N, MK = 576, 520, 15
    c = dict(name="bl_576x520_k15", M=M, N=N, MK=MK, blind=1, corr=0, step=1e-3, lambd=10000.0, tau=1e9, seed=4242)
    case = orc.synth_case(M, N, MK, seed=c["seed"], blind=True)
    c["window"] = orc.default_window(M, N, MK)
    out = {}
    logs = {}
    for n in (1, 2):
        img_r, u_r, psf_r, log_r = mg.run_ref(ref, case, c, n)
        _, u_o, psf_o, tr = mg.run_orc(case, c, n)
        assert np.array_equal(u_r, u_o) and np.array_equal(psf_r, psf_o) and log_r == tr.log.getvalue()
        assert np.array_equal(img_r, case["image"])
        uf = u_r.astype(np.float64)
        out["u_rows_%d" % n] = u_r[::48].copy()
        out["u_crop_%d" % n] = u_r[240:336, 212:308].copy()
        out["u_corner_%d" % n] = u_r[-40:, -40:].copy()
        out["moments_%d" % n] = np.array([uf.sum(), (uf ** 2).sum(), uf.min(), uf.max()])
        out["psf_%d" % n] = psf_r
        logs[str(n)] = log_r
        _, u_d, psf_d, _ = mg.run_orc(case, c, n, conv="direct")
        out["noise_%d" % n] = np.array([mg.rel(u_d, u_r), mg.rel(psf_d, psf_r)])
    out["M_r"] = np.array(tr.M_r, np.float32); out["Hu"] = np.array(tr.Hu, np.float32); out["varu"] = np.array(tr.varu, np.float32)
    meta = dict(c, logs=logs, crop=[240, 336, 212, 308], row_step=48, corner=40,
                versions=dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                              reference="aurelienpierre/Image-Cases-Studies lib/deconvolution.pyx (cython language_level=2, -O3 -fopenmp)"))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(mg.OUT, "rl_bl_576x520_k15.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), "noise floors", out["noise_1"], out["noise_2"])


if __name__ == "__main__":
    main()
