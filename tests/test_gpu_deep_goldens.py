"""north_star's depth at BASELINE.json's sizes: "<= 1e-4 max relative error vs reference after 50 iterations" (oracle/make_golden_deep.py).

The COMPILED REFERENCE (lib/deconvolution.pyx:460-591 itself) ran
  * bl_4096_k15_deep   configs[2], blind 4096^2 x 3, 15 x 15: a chain of five calls with iterations=2 -- 50 inner iterations free of the
                       rounding-fragile stop decision (`it > 1`, pyx:643, never holds inside a 2-outer call) -- snapshots every 10,
  * bl_4096_k15_stop   the same problem as one call with iterations=10: wherever the reference's own stop test (pyx:643-654) ends it,
  * nb_2048_k15_deep   configs[1], non-blind 2048^2, 15 x 15, step 1e-3, one call of 10 outer = 50 inner iterations,
  * nb_2048_k15_s1e-4  the same at step 1e-4, 50 outer = 250 inner iterations (SURVEY.md 8c's long-run regime),
  * bl_6144_k31_deep   configs[3], blind 6144^2, 31 x 31, one call of 2 outer = 10 inner iterations,
and the fixtures keep crops / rows / columns / moments of u, the PSF and the log (see test_gpu_baseline_goldens.py).  Here the product
path makes the same calls from the same inputs on every convolution path (default kernels, fp32 products, FFT tiles where they exist);
gate = 1e-4 of the reference's maximum (north_star), or twice the fixture's recorded noise floor (float64-direct oracle vs reference)
where that is larger, printed beside the measured deviation."""
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


def compare(z, meta, tag, u, psf, gate):
    w = meta["where"]
    c, s = w["centre"], w["seam"]
    got = dict(centre=u[c[0]:c[1], c[2]:c[3]], seam=u[s[0]:s[1], s[2]:s[3]], corner=u[-w["corner"]:, -w["corner"]:], origin=u[:w["origin"], :w["origin"]])
    if "u_rows_%s" % tag in z.files:
        got.update(rows=u[::meta["row_step"]], cols=u[:, ::meta["row_step"]])
    den = float(z["moments_%s" % tag][3])                               # max of the reference's u
    errs = {k: float(np.max(np.abs(v.astype(np.float64) - z["u_%s_%s" % (k, tag)]))) / den for k, v in got.items()}
    ep = rel_err(psf, z["psf_%s" % tag])
    uf = u.astype(np.float64)
    mom = np.array([uf.sum(), (uf ** 2).sum(), uf.min(), uf.max()])
    h2, w2 = uf.shape[0] // 2, uf.shape[1] // 2
    quad = np.array([[uf[a:a + h2, b:b + w2, ch].sum() for ch in range(3)] for a in (0, h2) for b in (0, w2)])
    assert max(errs.values()) < gate, (tag, errs)
    assert ep < gate, (tag, ep)
    assert np.all(np.abs(mom - z["moments_%s" % tag]) <= 1e-5 * np.abs(z["moments_%s" % tag]))
    assert np.all(np.abs(quad - z["quadrants_%s" % tag]) <= 1e-5 * np.abs(z["quadrants_%s" % tag]))
    return max(errs.values()), ep


def log_numbers(line):
    return [float(t) for t in line.replace("|", " ").replace("=", " ").split() if t.replace(".", "").replace("-", "").isdigit()]


CONV = [pytest.param(0, id="default-kernels"), pytest.param(1, id="fp32-products"), pytest.param(3, id="fft-tiles")]


@pytest.mark.parametrize("conv", CONV)
@pytest.mark.parametrize("name", ["nb_2048_k15_deep", "nb_2048_k15_s1e-4", "bl_4096_k15_deep", "bl_4096_k15_stop", "bl_6144_k31_deep"])
def test_reference_trajectory_at_depth(golden_dir, name, conv):
    from lib import deconvolution as dc
    path = os.path.join(golden_dir, "rl_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated (oracle/make_golden_deep.py)" % name)
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    M, N, MK = meta["M"], meta["N"], meta["MK"]
    case = case_of(meta)
    dc._drop_jobs()
    floor = meta.get("noise_floor")
    gate = 1e-4 if not floor else max(1e-4, 2 * max(floor[0], floor[1]))
    u, psf = case["u0"].copy(), case["psf0"].copy()
    image = case["image"].copy()
    for k, tag in enumerate(meta["tags"]):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = dc.richardson_lucy_MM(image, u, psf, *meta["window"], meta["tau"], M, N, 3, MK, meta["iters"], meta["step"], meta["lambd"],
                                        blind=bool(meta["blind"]), conv=conv)
        st = dc.richardson_lucy_MM.last
        assert np.shares_memory(out, u) and not st.has_nan
        assert np.array_equal(image, case["image"])                       # pyx:545-549 subtract exactly 0
        ref_lines = meta["logs"][tag].splitlines()
        done_ref = sum(1 for l in ref_lines if l.startswith("DoF"))
        # the stop decision (pyx:643-654) must be the reference's: same number of outer iterations, same closing lines
        assert st.iterations_done == done_ref, (tag, st.iterations_done, done_ref)
        eu, ep = compare(z, meta, tag, u, psf, gate)
        print("%s conv=%d after %s outer iterations (%d inner): u %.2e psf %.2e (gate %.1e%s)"
              % (name, conv, tag, 5 * int(tag), eu, ep, gate, ", noise floor %.1e / %.1e" % (floor[0], floor[1]) if floor else ""))
        lines = buf.getvalue().splitlines()
        assert len(lines) == len(ref_lines)
        for lg, lr in zip(lines, ref_lines):
            if lg != lr:
                vg, vr = log_numbers(lg), log_numbers(lr)
                assert len(vg) == len(vr) and np.allclose(vg, vr, rtol=5e-4, atol=2e-6), (lg, lr)
    dc._drop_jobs()
