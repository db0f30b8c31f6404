"""one image split into row bands over several device jobs (lib/banded.py).  On the single-GPU test box all
bands live on device 0; the band arithmetic (owned rows, halos, the three cross-band steps) is the same with one band per GPU."""
import contextlib
import io

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import rel_err


def test_split_rows_and_ownership():
    from lib import banded
    assert banded.split_rows(100, 3, 4) == [(0, 33), (33, 66), (66, 100)]
    with pytest.raises(ValueError):
        banded.split_rows(20, 3, 7)
    M, pad = 100, 4
    bands = [banded._Band(k, y0, y1, M, pad, 0) for k, (y0, y1) in enumerate(banded.split_rows(M, 3, pad))]
    owned = sorted(r for b in bands for r in range(b.u0, b.u1))
    assert owned == list(range(M + 2 * pad))                               # every u row is owned exactly once
    for b in bands:
        assert b.lu0 == (0 if b.first else 2 * pad)                        # the halo above an interior band is 2 pad rows
        assert (b.b - b.a) + 2 * pad - b.lu1 == (0 if b.last else 2 * pad)


def run_single(case, M, N, MK, win, tau, iters, blind, conv, flags=0):
    from lib import deconvolution as dc
    u, psf = case["u0"].copy(), case["psf0"].copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *win, tau, M, N, 3, MK, iters, 1e-3, 1e4, blind=blind, conv=conv, flags=flags)
    return u, psf, buf.getvalue(), dc.richardson_lucy_MM.last


def run_banded(case, M, N, MK, win, tau, iters, blind, conv, bands):
    from lib import banded
    u, psf = case["u0"].copy(), case["psf0"].copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = banded.richardson_lucy_MM_banded(case["image"].copy(), u, psf, *win, tau, M, N, 3, MK, iters, 1e-3, 1e4, blind=blind, conv=conv, bands=bands)
    assert np.shares_memory(out, u)
    return u, psf, buf.getvalue(), banded.richardson_lucy_MM_banded.last


@pytest.mark.gpu
@pytest.mark.parametrize("bands", [2, 3, 5])
def test_nonblind_fp32_bands_are_bit_identical_to_one_job(bands):
    """fp32 convolution kernels accumulate every pixel in the same tap order whatever the tiling: a banded non-blind run
    must reproduce the single-job run bit for bit (halo width, ownership and the max-combine are exact or wrong)."""
    M, N, MK = 230, 150, 9
    case = orc.synth_case(M, N, MK, seed=4)
    win = (60, 121, 20, 101)
    u1, _, log1, st1 = run_single(case, M, N, MK, win, 1e9, 3, False, 1)
    u2, _, log2, st2 = run_banded(case, M, N, MK, win, 1e9, 3, False, 1, bands)
    assert st2.iterations_done == st1.iterations_done == 3
    assert np.array_equal(u1, u2)
    assert [l for l in log1.splitlines() if "DoF" not in l] == [l for l in log2.splitlines() if "DoF" not in l]


@pytest.mark.gpu
@pytest.mark.parametrize("bands,conv", [(2, 0), (3, 0), (3, 1)])
def test_blind_bands_match_one_job(bands, conv):
    M, N, MK = 300, 260, 15
    case = orc.synth_case(M, N, MK, seed=9, blind=True)
    win = (110, 191, 60, 201)
    u1, p1, _, st1 = run_single(case, M, N, MK, win, 0.0, 2, True, conv, flags=1)
    u2, p2, _, st2 = run_banded(case, M, N, MK, win, 0.0, 2, True, conv, bands)
    eu, ep = rel_err(u2, u1), rel_err(p2, p1)
    print("blind %d bands conv=%d: rel err u %.2e psf %.2e" % (bands, conv, eu, ep))
    assert st1.iterations_done == st2.iterations_done
    assert eu < 2e-6 and ep < 2e-6
    assert abs(st2.M_r - st1.M_r) <= 2e-3 * abs(st1.M_r) and abs(st2.Hu - st1.Hu) <= 1e-4 * abs(st1.Hu)


@pytest.mark.gpu
def test_banded_run_against_the_reference_golden(golden_dir):
    """the 129 x 129 blind golden from the compiled reference, deconvolved as 3 bands"""
    from helpers import load_golden
    z, meta = load_golden(golden_dir, "bl_129x129_k15")
    M, N, MK = meta["M"], meta["N"], meta["MK"]
    case = dict(image=z["image"], u0=z["u0"], psf0=z["psf0"])
    u, psf, _, st = run_banded(case, M, N, MK, tuple(meta["window"]), meta["tau"], 2, True, 0, 3)
    assert st.iterations_done == 2
    assert rel_err(u, z["u_2"]) < 1e-5 and rel_err(psf, z["psf_2"]) < 1e-5


@pytest.mark.gpu
def test_stop_test_works_across_bands():
    M, N, MK = 200, 140, 5
    case = orc.synth_case(M, N, MK, seed=13)
    win = (70, 131, 30, 111)
    u1, _, log1, st1 = run_single(case, M, N, MK, win, 0.0, 40, False, 1)
    u2, _, log2, st2 = run_banded(case, M, N, MK, win, 0.0, 40, False, 1, 2)
    assert st1.stopped and st2.stopped and st1.iterations_done == st2.iterations_done
    assert np.array_equal(u1, u2)
