"""CPU: the numpy oracle (oracle/rl_mm_oracle.py) against the golden vectors produced by the compiled
reference.  In the container that generated them the match is bit for bit; 1e-6 leaves room for a
different CPU's FFT code path."""
import json
import os

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import load_golden, rel_err

CASES = ["nb_33x37_k3", "nb_65x65_k7", "nb_65x81_k9_pcpsf", "nb_129x129_k15", "nb_97x97_k5_tau",
         "bl_65x49_k9", "bl_129x129_k15", "bl_65x65_k7_corr", "bl_101x101_k11_s1e-4"]


def run_oracle(z, meta, iters, conv="scipy"):
    image, u, psf = z["image"].copy(), z["u0"].copy(), z["psf0"].copy()
    tr = orc.Trace()
    out = orc.richardson_lucy_MM(image, u, psf, *meta["window"], meta["tau"], meta["M"], meta["N"], 3, meta["MK"], iters,
                                 meta["step"], meta["lambd"], blind=meta["blind"], correlation=meta["corr"], conv=conv,
                                 trace=tr, quiet=True)
    assert np.shares_memory(out, u)
    return image, u, psf, tr


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_golden(golden_dir, name):
    z, meta = load_golden(golden_dir, name)
    for n in meta["snaps"]:
        image, u, psf, tr = run_oracle(z, meta, n)
        assert rel_err(u, z["u_%d" % n]) < 1e-6
        assert rel_err(psf, z["psf_%d" % n]) < 1e-6
        assert np.array_equal(image, z["image"])            # pyx:549 subtracts exactly zero
        assert tr.log.getvalue().splitlines()[-2:] == meta["logs"][str(n)].splitlines()[-2:] or \
            rel_err(u, z["u_%d" % n]) > 0                  # identical log when arrays are bit-identical
    assert tr.iterations == meta["iterations_done"] and tr.stopped == meta["stopped"]
    np.testing.assert_allclose(np.array(tr.M_r), z["M_r"], rtol=1e-5)


@pytest.mark.parametrize("name", ["nb_65x65_k7", "bl_65x49_k9", "nb_129x129_k15"])
def test_direct_float64_convolution_noise_floor(golden_dir, name):
    """The oracle with float64 direct sums instead of scipy's complex64 FFT: the 'noise floor' any
    non-FFT implementation sits at relative to the reference."""
    z, meta = load_golden(golden_dir, name)
    n = meta["snaps"][1]
    _, u, psf, _ = run_oracle(z, meta, n, conv="direct")
    assert rel_err(u, z["u_%d" % n]) < 1e-5
    assert rel_err(psf, z["psf_%d" % n]) < 1e-5


def test_noise_floor_recorded_in_fixtures(golden_dir):
    """Every fixture carries the float64-direct noise floor; short runs sit far below the 1e-4 bar."""
    for name in CASES:
        z, meta = load_golden(golden_dir, name)
        first = meta["snaps"][0]
        if first <= 2:
            assert meta["noise_floor"][str(first)][0] < 1e-5, name


def test_oracle_long_run_small_step(golden_dir):
    z, meta = load_golden(golden_dir, "nb_129x129_k15_s1e-4")
    _, u, _, tr = run_oracle(z, meta, 50)
    assert rel_err(u, z["u_50"]) < 1e-6 and tr.iterations == 50


def test_normalize_kernel(golden_dir):
    z = np.load(os.path.join(golden_dir, "normalize_kernel.npz"))
    for MK in (3, 7, 15, 31):
        k = z["in_%d" % MK].copy()
        orc.normalize_kernel(k, MK)
        assert np.array_equal(k, z["out_%d" % MK])
        assert np.all(k >= 0) and np.allclose(k.sum(axis=(0, 1)), 1, atol=1e-5)


def test_config1_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "rl_config1_512_k9_20.npz"))
    meta = json.loads(str(z["meta"]))
    case = orc.synth_case(512, 512, 9, seed=0)
    u = case["u0"].copy()
    orc.richardson_lucy_MM(case["image"].copy(), u, case["psf0"].copy(), *meta["window"], meta["tau"], 512, 512, 3, 9, 20,
                           meta["step"], meta["lambd"], blind=False, quiet=True)
    c = meta["crop"]
    assert rel_err(u[c[0]:c[1], c[2]:c[3]], z["u_crop"]) < 1e-6


def test_index_forms_of_the_three_convolutions():
    """SURVEY.md 8a 'exact index forms': A1 valid convolution, A3 its adjoint, A13 the PSF gradient."""
    rng = np.random.default_rng(0)
    K, M, N = 5, 9, 11
    u = rng.standard_normal((M + K - 1, N + K - 1))
    psf = rng.standard_normal((K, K))
    e = rng.standard_normal((M, N))
    synth = orc._conv_direct(u, psf, "valid")
    ref = np.zeros((M, N))
    for i in range(M):
        for j in range(N):
            ref[i, j] = sum(psf[p, q] * u[i + K - 1 - p, j + K - 1 - q] for p in range(K) for q in range(K))
    assert np.allclose(synth, ref)
    # <conv(u), e> == <u, corr_full(e)>  and  gradk = d/dpsf 1/2||conv(u) - f||^2 direction
    g = orc._conv_direct(e, psf[::-1, ::-1], "full")
    assert np.isclose(np.sum(synth * e), np.sum(u * g))
    gk = orc._conv_direct(u[::-1, ::-1], e, "valid")
    ref_gk = np.array([[np.sum(e * u[K - 1 - a:K - 1 - a + M, K - 1 - b:K - 1 - b + N]) for b in range(K)] for a in range(K)])
    assert np.allclose(gk, ref_gk)
    from scipy.signal import convolve
    assert np.allclose(convolve(u, psf, mode="valid"), synth) and np.allclose(convolve(e, psf[::-1, ::-1], mode="full"), g)
