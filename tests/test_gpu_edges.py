"""Edge cases of the drop-in surface on the GPU: tiny and ragged frames, zero iterations, NaN input,
degenerate stats windows, the largest PSF the reference's own examples use, repeated calls on a cached job."""
import contextlib
import io

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import rel_err

pytestmark = pytest.mark.gpu


def run_both(case, M, N, MK, window, iters, blind, step=1e-3, lambd=1e4, tau=1e9):
    from lib import deconvolution as dc
    args = (*window, tau, M, N, 3, MK, iters, step, lambd)
    u_r, psf_r = case["u0"].copy(), case["psf0"].copy()
    tr = orc.Trace()
    orc.richardson_lucy_MM(case["image"].copy(), u_r, psf_r, *args, blind=blind, quiet=True, trace=tr)
    u, psf = case["u0"].copy(), case["psf0"].copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *args, blind=blind)
    return (u, psf, buf.getvalue(), dc.richardson_lucy_MM.last), (u_r, psf_r, tr)


@pytest.mark.parametrize("M,N,MK,blind", [(5, 7, 3, False), (7, 5, 3, True), (9, 64, 5, True), (64, 9, 5, False), (1, 9, 3, False), (65, 1, 3, True)])
def test_tiny_and_ragged_frames(M, N, MK, blind):
    case = orc.synth_case(M, N, MK, seed=M * 11 + N, blind=blind)
    win = (0, M, 0, N)   # whole image as the stats window (the u window may be empty -> varu = nan, like numpy)
    (u, psf, log, st), (u_r, psf_r, tr) = run_both(case, M, N, MK, win, 2, blind)
    assert rel_err(u, u_r) < 1e-4 and rel_err(psf, psf_r) < 1e-4
    assert st.iterations_done == tr.iterations == 2


def test_zero_iterations_leaves_everything_untouched():
    case = orc.synth_case(33, 37, 3, seed=1)
    (u, psf, log, st), _ = run_both(case, 33, 37, 3, orc.default_window(33, 37, 3), 0, True)
    assert np.array_equal(u, case["u0"]) and np.array_equal(psf, case["psf0"]) and st.iterations_done == 0
    assert "Did not converge after 0 iterations" in log


def test_nan_in_the_image_is_reported_not_raised():
    """pyx:671-672: NaN is printed, never raised; it spreads exactly as far as in the reference."""
    case = orc.synth_case(40, 44, 5, seed=2)
    case["image"][20, 21, 1] = np.nan
    case["u0"] = np.ascontiguousarray(np.pad(case["image"], ((2, 2), (2, 2), (0, 0)), mode="edge"))
    (u, psf, log, st), (u_r, _, tr) = run_both(case, 40, 44, 5, orc.default_window(40, 44, 5), 1, False)
    assert st.has_nan and "has NaN after DoF correction" in log
    assert np.array_equal(np.isnan(u), np.isnan(u_r))


def test_single_pixel_window_and_window_at_the_border():
    case = orc.synth_case(48, 40, 7, seed=3)
    for win in [(10, 11, 12, 13), (0, 48, 0, 40), (41, 48, 33, 40)]:
        (u, psf, log, st), (u_r, psf_r, tr) = run_both(case, 48, 40, 7, win, 2, False)
        assert rel_err(u, u_r) < 1e-4
        if np.isfinite(tr.M_r[-1]):
            assert abs(st.M_r - tr.M_r[-1]) <= 5e-3 * abs(tr.M_r[-1])


def test_psf_45_as_in_the_reference_examples():
    """deconvolve.py:409 (commented example) uses a 45-px blur; build_pyramid(45) = [45, 31, 21, 15, 11, 7, 5, 3]."""
    M, N, MK = 120, 100, 45
    case = orc.synth_case(M, N, MK, seed=4, blind=True)
    (u, psf, log, st), (u_r, psf_r, tr) = run_both(case, M, N, MK, orc.default_window(M, N, MK), 1, True)
    assert rel_err(u, u_r) < 1e-4 and rel_err(psf, psf_r) < 1e-4


def test_repeated_calls_reuse_the_cached_job_and_stay_deterministic():
    case = orc.synth_case(65, 65, 7, seed=5, blind=True)
    win = orc.default_window(65, 65, 7)
    (u1, p1, _, _), _ = run_both(case, 65, 65, 7, win, 2, True)
    (u2, p2, _, _), _ = run_both(case, 65, 65, 7, win, 2, True)
    assert np.array_equal(u1, u2) and np.array_equal(p1, p2)       # bitwise reproducible run to run
    other = orc.synth_case(40, 80, 9, seed=6)
    (u3, _, _, _), (u3r, _, _) = run_both(other, 40, 80, 9, orc.default_window(40, 80, 9), 1, False)
    assert rel_err(u3, u3r) < 1e-4


def test_unsupported_sizes_fail_loudly():
    from lib import _native as nv
    with pytest.raises(nv.NativeError) as ei:
        nv.RLJob(32, 32, 257)            # (65 ... 255: tests/test_gpu_bigpsf.py)
    assert ei.value.code == nv.ICS_ENOSUP
    with pytest.raises(nv.NativeError) as ei:
        nv.RLJob(32, 32, 4)
    assert ei.value.code == nv.ICS_EINVAL
    job = nv.RLJob(32, 32, 3)
    with pytest.raises(nv.NativeError) as ei:            # window outside the image
        job.upload(np.zeros((32, 32, 3), np.float32), np.zeros((34, 34, 3), np.float32), np.full((3, 3, 3), 1 / 9, np.float32))
        job.run(job.params(0, 40, 0, 10, 0, 1, 1e-3, 1e4, False))
    assert ei.value.code == nv.ICS_EINVAL
    job.close()


def test_frames_beyond_the_32_bit_offset_range_are_refused_not_corrupted():
    """the matrix-core kernels use 32-bit byte offsets into a frame: 2 GiB per frame is the documented limit"""
    from lib import _native
    with pytest.raises(_native.NativeError) as ei:
        _native.RLJob(14000, 14000, 15)
    assert ei.value.code == _native.ICS_ENOSUP


def test_empty_stats_window_gives_nan_statistics_like_the_reference():
    """error[top:bottom, left:right] with bottom <= top is an empty array in the reference: M_r, Hu and varu are NaN
    (numpy warns, lib/deconvolution.pyx:600-601,627-638 do not raise), the stop test never fires, and u does not depend
    on the window at all."""
    from lib import _native as nv
    M, N, MK = 48, 56, 5
    case = orc.synth_case(M, N, MK, seed=2)
    res = []
    for win in ((10, 10, 5, 30), orc.default_window(M, N, MK)):
        job = nv.RLJob(M, N, MK)
        job.upload(case["image"], case["u0"], case["psf0"])
        st = job.run(job.params(*win, 0.0, 4, 1e-3, 1e4, False))
        res.append((job.download()[0], st))
        job.close()
    (u_e, st_e), (u_w, st_w) = res
    assert np.isnan(st_e.M_r) and np.isnan(st_e.Hu) and np.isnan(st_e.varu)
    assert st_e.iterations_done == 4 and not st_e.stopped
    if not st_w.stopped:
        assert np.array_equal(u_e, u_w)


def test_window_then_empty_window_then_window_again_on_one_job():
    """Round-2 advice: the cached-window early return of ensure_window came before win_empty was recomputed, so the call sequence
    W, empty window, W on one job (jobs are cached per size by lib/deconvolution.py) left the statistics NaN for good."""
    from lib import _native as nv
    M, N, MK = 48, 56, 5
    case = orc.synth_case(M, N, MK, seed=2)
    W = orc.default_window(M, N, MK)
    job = nv.RLJob(M, N, MK)
    res = []
    for win in (W, (10, 10, 5, 30), W):
        job.upload(case["image"], case["u0"], case["psf0"])
        st = job.run(job.params(*win, 1e9, 3, 1e-3, 1e4, False))
        res.append((st.M_r, st.Hu, st.varu, job.download()[0]))
    job.close()
    assert np.isnan(res[1][0]) and np.isnan(res[1][1]) and np.isnan(res[1][2])
    assert not np.isnan(res[0][0]) and res[0][:3] == res[2][:3], (res[0][:3], res[2][:3])
    assert np.array_equal(res[0][3], res[2][3]) and np.array_equal(res[0][3], res[1][3])
