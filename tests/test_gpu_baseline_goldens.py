"""Reference trajectories at the frame sizes BASELINE.json's metric is quoted on (oracle/make_golden_baseline.py): the COMPILED
REFERENCE (lib/deconvolution.pyx:460-659 itself, not a float64 stage pass) ran configs[1] -- non-blind 2048^2, 15 x 15, two outer
iterations --, the blind loop at 2048^2 (two outer iterations), one outer iteration of configs[2] -- blind 4096^2, 15 x 15, the
headline workload -- and one of configs[3] -- blind 6144^2, 31 x 31 (two-window 8-wave convolutions, 2 x 2 tap-block gradient); the fixtures keep crops (centre, a corner shared by four 64 x 64 tiles, frame corner, frame origin), every n-th
row and column, float64 moments and quadrant sums of the whole frame, the PSF and the reference's stdout.  Here the product path
runs the same calls on the full grid with the default kernels, and with fp32 products (`conv=1`); gate = 1e-5 on u and on the PSF
(the north-star bar is 1e-4), the whole-frame sums to 1e-6."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import rel_err

pytestmark = pytest.mark.gpu

_cases = {}


def case_of(meta):
    key = (meta["M"], meta["N"], meta["MK"], meta["seed"], meta["blind"])
    if key not in _cases:
        _cases.clear()                      # one full-size problem in host memory at a time
        _cases[key] = orc.synth_case_large(meta["M"], meta["N"], meta["MK"], seed=meta["seed"], blind=bool(meta["blind"]))
    return _cases[key]


@pytest.mark.parametrize("conv", [0, 1], ids=["default-kernels", "fp32-products"])
@pytest.mark.parametrize("name", ["nb_2048_k15", "bl_2048_k15", "bl_4096_k15", "bl_6144_k31"])
def test_reference_trajectory_at_baseline_size(golden_dir, name, conv):
    from lib import deconvolution as dc
    z = np.load(os.path.join(golden_dir, "rl_%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    M, N, MK = meta["M"], meta["N"], meta["MK"]
    case = case_of(meta)
    dc._drop_jobs()
    for n in meta["iters"]:
        u, psf = case["u0"].copy(), case["psf0"].copy()
        image = case["image"].copy()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = dc.richardson_lucy_MM(image, u, psf, *meta["window"], meta["tau"], M, N, 3, MK, n, meta["step"], meta["lambd"],
                                        blind=bool(meta["blind"]), conv=conv)
        st = dc.richardson_lucy_MM.last
        assert np.shares_memory(out, u) and st.iterations_done == n and not st.has_nan
        assert np.array_equal(image, case["image"])                       # pyx:545-549 subtract exactly 0
        w = meta["where"]
        c, s = w["centre"], w["seam"]
        got = dict(centre=u[c[0]:c[1], c[2]:c[3]], seam=u[s[0]:s[1], s[2]:s[3]], corner=u[-w["corner"]:, -w["corner"]:],
                   origin=u[:w["origin"], :w["origin"]], rows=u[::meta["row_step"]], cols=u[:, ::meta["row_step"]])
        den = float(z["moments_%d" % n][3])                               # max of the reference's u
        errs = {k: float(np.max(np.abs(v.astype(np.float64) - z["u_%s_%d" % (k, n)]))) / den for k, v in got.items()}
        ep = rel_err(psf, z["psf_%d" % n])
        uf = u.astype(np.float64)
        mom = np.array([uf.sum(), (uf ** 2).sum(), uf.min(), uf.max()])
        h2, w2 = uf.shape[0] // 2, uf.shape[1] // 2
        quad = np.array([[uf[a:a + h2, b:b + w2, ch].sum() for ch in range(3)] for a in (0, h2) for b in (0, w2)])
        print("%s conv=%d, %d outer: u %s psf %.2e" % (name, conv, n, " ".join("%s %.1e" % kv for kv in errs.items()), ep))
        assert max(errs.values()) < 1e-5, errs
        assert ep < 1e-5
        assert np.all(np.abs(mom - z["moments_%d" % n]) <= 1e-6 * np.abs(z["moments_%d" % n]))
        assert np.all(np.abs(quad - z["quadrants_%d" % n]) <= 1e-6 * np.abs(z["quadrants_%d" % n]))
        # the reference's own progress lines (DoF extrema printed with six decimals may differ in the last digit)
        lines, ref = buf.getvalue().splitlines(), meta["logs"][str(n)].splitlines()
        assert len(lines) == len(ref)
        for lg, lr in zip(lines, ref):
            if lg != lr:
                vg = [float(t) for t in lg.replace("|", " ").replace("=", " ").split() if t.replace(".", "").replace("-", "").isdigit()]
                vr = [float(t) for t in lr.replace("|", " ").replace("=", " ").split() if t.replace(".", "").replace("-", "").isdigit()]
                assert len(vg) == len(vr) and np.allclose(vg, vr, rtol=2e-4, atol=2e-6), (lg, lr)
        if "M_r" in z.files:
            k = st.trace_len
            np.testing.assert_allclose(np.array(st.trace_M_r[:k]), z["M_r"][:k], rtol=5e-3)
            np.testing.assert_allclose(np.array(st.trace_Hu[:k]), z["Hu"][:k], rtol=5e-3)
            np.testing.assert_allclose(np.array(st.trace_varu[:k]), z["varu"][:k], rtol=1e-3)
    dc._drop_jobs()
