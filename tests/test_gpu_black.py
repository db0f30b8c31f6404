"""Frames with exactly-black regions (clipped shadows, letterbox bars, zero borders) -- lib/deconvolution.pyx:499-502,552.

The DoF mask is ((gradu - image)/(gradu + image))^2.  In a region where image and u are exactly 0 the reference's gradu is the
rounding noise of its complex64 FFT and the ratio is 1 for every non-zero noise value: the compiled reference returns a finite
picture there unless one noise value happens to be exactly 0 (then its whole frame is NaN).  The device convolutions are
exact, gradu = 0, and IEEE 0/0 would make every such frame NaN; the library defines the ratio as 1 where gradu == image == 0
(include/ics_hip.h "DoF ratio", csrc/ics_common.h ics_dof_ratio).  tests/golden/rl_black.npz (oracle/make_golden_black.py)
holds the COMPILED REFERENCE's results on 36 such frames: where it returned a finite picture (22 of the 24 row-band cases)
the device result must match it at the north-star gate; in every case it must be finite and match the float64-direct oracle,
which carries the same rule (oracle/rl_mm_oracle.py dof_ratio).
"""
import contextlib
import io
import json
import os

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import rel_err

TOL = 1e-4


def _load(golden_dir):
    z = np.load(os.path.join(golden_dir, "rl_black.npz"))
    return z, json.loads(str(z["meta"]))


def _names():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    _, meta = _load(here)
    return sorted(meta["cases"])


def _inputs(c):
    case = orc.black_case(c["M"], c["N"], c["MK"], c["kind"], seed=c["seed"], blind=bool(c["blind"]))
    sums = [float(case["image"].astype(np.float64).sum()), float(case["u0"].astype(np.float64).sum())]
    assert np.allclose(sums, c["input_sums"], rtol=1e-12), "the seeded inputs are not the generator's"
    return case


def _run_direct_oracle(case, c, iters):
    image, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    orc.richardson_lucy_MM(image, u, psf, *c["window"], c["tau"], c["M"], c["N"], 3, c["MK"], iters, c["step"], c["lambd"],
                           blind=c["blind"], correlation=0, quiet=True, conv="direct")
    return u, psf


def _run_gpu(case, c, iters, conv, fn=None, **kw):
    from lib import deconvolution as dc
    fn = fn or dc.richardson_lucy_MM
    image, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(image, u, psf, *c["window"], c["tau"], c["M"], c["N"], 3, c["MK"], iters, c["step"], c["lambd"],
           blind=c["blind"], correlation=0, conv=conv, **kw)
    return u, psf, buf.getvalue()


def test_direct_oracle_with_the_rule_matches_the_reference_goldens(golden_dir):
    """CPU: the float64-direct oracle, ratio 1 at exact 0/0, against the compiled reference on black row bands (K = 9 cases;
    the generator asserted all of them when it wrote the fixture)."""
    z, meta = _load(golden_dir)
    n = 0
    for name, c in sorted(meta["cases"].items()):
        if c["MK"] != 9 or name + "/u" not in z.files:
            continue
        u, psf = _run_direct_oracle(_inputs(c), c, meta["iters"])
        assert not np.isnan(u).any()
        assert rel_err(u, z[name + "/u"]) < 5e-5 and rel_err(psf, z[name + "/psf"]) < 1e-6, name
        n += 1
    assert n == 8


def test_ieee_ratio_in_exact_arithmetic_loses_the_frame():
    """what the rule is for: the same oracle with the plain IEEE ratio returns an all-NaN frame on a black band"""
    c = dict(M=65, N=57, MK=9, kind="band_mid", seed=900, blind=0)
    case = orc.black_case(c["M"], c["N"], c["MK"], c["kind"], seed=c["seed"])
    image, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    orc.richardson_lucy_MM(image, u, psf, *orc.default_window(65, 57, 9), 1e9, 65, 57, 3, 9, 1, 1e-3, 1e4, blind=False, quiet=True,
                           conv="direct", dof_zero_rule=float("nan"))
    assert np.isnan(u).all()
    g = np.zeros((4, 4, 3), np.float32); f = np.zeros((4, 4, 3), np.float32); g[0, 0] = 1e-10; g[1, 1] = -1e-10; f[2, 2] = 0.5; g[3, 3] = 1.0; f[3, 3] = -1.0
    r = orc.dof_ratio(g, f)
    assert r[0, 0, 0] == 1 and r[1, 1, 0] == 1 and r[0, 1, 0] == 1 and r[2, 2, 0] == -1 and np.isinf(r[3, 3, 0])   # only exact 0/0 is defined away


@pytest.mark.gpu
@pytest.mark.parametrize("conv", [0, 1], ids=["auto", "fp32"])
@pytest.mark.parametrize("name", _names())
def test_black_region_frames(golden_dir, name, conv):
    z, meta = _load(golden_dir)
    c = meta["cases"][name]
    case = _inputs(c)
    u, psf, log = _run_gpu(case, c, meta["iters"], conv)
    assert not np.isnan(u).any() and not np.isnan(psf).any(), name
    assert "has NaN" not in log
    ud, pd = _run_direct_oracle(case, c, meta["iters"])
    eu, ep = rel_err(u, ud), rel_err(psf, pd)
    msg = "%s conv=%d: vs float64-direct oracle u %.2e psf %.2e" % (name, conv, eu, ep)
    assert eu < TOL and ep < TOL, msg
    # pixels whose whole (2K-1)^2 dependency window is black have an exactly-zero back-projection in the first inner iteration
    from scipy.ndimage import minimum_filter
    deep = minimum_filter(case["black"].astype(np.uint8), size=2 * c["MK"] - 1, mode="nearest").astype(bool)
    assert deep.any(), "the case does not hold a pixel with an exactly-zero back-projection"
    if c["blind"]:   # D = 1 on every black pixel (g != 0: (g - 0)/(g + 0); g == 0: the rule) -> u = image = 0 there, as in the reference
        pad = c["MK"] // 2
        assert np.all(u[pad:-pad, pad:-pad][case["black"]] == 0.0), name
    if name + "/u" in z.files:
        er, epr = rel_err(u, z[name + "/u"]), rel_err(psf, z[name + "/psf"])
        msg += " | vs COMPILED REFERENCE u %.2e psf %.2e" % (er, epr)
        assert er < TOL and epr < TOL, msg
        ref_lines = [l for l in c["log"].splitlines() if "iterations" in l]
        assert all(l in log for l in ref_lines)
    else:
        msg += " | the reference returned NaN in %.0f %% of this frame (an exact zero in its FFT noise)" % (100 * c["ref_nan"])
    print(msg)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["band_mid_nb_k15", "band_top_bl_k9", "band_bot_bl_k15"])
def test_black_band_through_the_row_band_path(golden_dir, name):
    """the same frames as two row bands (lib/banded.py): the band seam crosses or touches the black region"""
    from lib import banded
    z, meta = _load(golden_dir)
    c = meta["cases"][name]
    case = _inputs(c)
    u, psf, _ = _run_gpu(case, c, meta["iters"], 0, fn=banded.richardson_lucy_MM_banded, bands=2)
    assert not np.isnan(u).any()
    er, epr = rel_err(u, z[name + "/u"]), rel_err(psf, z[name + "/psf"])
    print("%s as 2 bands: vs COMPILED REFERENCE u %.2e psf %.2e" % (name, er, epr))
    assert er < TOL and epr < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["rows", "pixel_groups", "fused_update_synth"])
@pytest.mark.parametrize("blind", [False, True])
def test_black_band_update_stage_is_bit_exact(blind, kernel, debug_switch):
    """A5-A10 on a frame whose back-projection and image hold exact zeros: every update kernel (k_update_rows, k_update, the
    opt-in fused update + convolution) against the numpy float32 restatement, rule included, bit for bit."""
    from lib import _native as nv
    from helpers import update_f32
    M, N, MK = 96, 80, 9
    pad = MK // 2
    case = orc.black_case(M, N, MK, "band_mid", seed=5, blind=blind)
    rng = np.random.default_rng(1)
    ut = case["u0"]
    u = (ut + 0.02 * rng.standard_normal(ut.shape)).astype(np.float32)
    u[ut == 0] = 0.0
    u[pad + M // 2, pad + 7, 1] = -0.0                       # a signed zero is a zero
    if kernel == "pixel_groups":
        debug_switch("update_kernel", 0)
    job = nv.RLJob(M, N, MK)
    try:
        job.upload(case["image"], case["u0"], case["psf0"])
        job.write(nv.BUF_U, u)
        job.write(nv.BUF_UT, ut)
        p = job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=blind, conv=nv.CONV_VECTOR)
        job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
        job.stage(nv.STAGE_BACKPROJECT, p)
        g_raw = job.read(nv.BUF_GRADU)
        gi = g_raw[pad:-pad, pad:-pad]
        both = (gi == 0) & (case["image"] == 0)
        assert both.sum() > 1000, "no exact 0/0 pixels: the case does not exercise the rule"
        job.stage(nv.STAGE_UPDATE_SYNTH if kernel == "fused_update_synth" else nv.STAGE_UPDATE, p)
        got = job.read(nv.BUF_U)
        sc = job.scalars()
    finally:
        job.close()
    want, dt, DoF = update_f32(u, ut, g_raw, case["image"], 1e-3, 10000.0, blind, pad)
    assert not np.isnan(want).any() and not np.isnan(got).any()
    assert np.array_equal(got, want)
    assert DoF[both].min() == DoF[both].max() == (np.float32(1.0) if blind else np.float32(1.0) / np.float32(10000.0))
