"""The cooperative small-frame iteration kernel (csrc/ics_small.hip): one launch per outer iteration, a (tile, channel) of the u-frame per
compute unit, operands resident in LDS -- what ICS_CONV_AUTO runs the shipped loop with on frames up to ~290 px a side (the blind phase of
/root/reference/deconvolve.py:277-286 works on a 255 x 255 window at every pyramid level).

Gates: the north-star 1e-4 against the oracle (lib/deconvolution.pyx:460-654 restated in oracle/rl_mm_oracle.py) on u, the PSF and the
stop-test traces; 5e-6 against the multi-launch kernel families on the same inputs (both are fp32 sums of the same products in different
orders); identical stop decisions."""
import contextlib
import io

import numpy as np
import pytest

import rl_mm_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _run(case, M, N, MK, win, iters, blind, correlation=0, stop_test=2, tau=1e9, lambd=10000.0, profile=0, progress=None):
    from lib import _native as nv
    job = nv.RLJob(M, N, MK)
    try:
        job.upload(case["image"], case["u0"], case["psf0"])
        p = job.params(*win, tau, iters, 1e-3, lambd, blind, correlation, 3, stop_test=stop_test, profile=profile)
        route = job.describe(p)
        st = job.run(p, progress) if progress else job.run(p)
        u, psf, psf_caller = job.download()
        n = st.trace_len
        return dict(u=u, psf=psf, psf_caller=psf_caller, st=st, route=route, M_r=np.array(st.trace_M_r[:n]), Hu=np.array(st.trace_Hu[:n]), varu=np.array(st.trace_varu[:n]),
                    dof_min=np.array(st.trace_dof_min[:n]), dof_max=np.array(st.trace_dof_max[:n]))
    finally:
        job.close()


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / max(float(np.max(np.abs(b))), 1e-30))


def test_routing_of_small_frames(debug_switch):
    from lib import _native as nv
    P = nv.RLJob.params
    blind = P(10, 200, 10, 200, 1e9, 2, 1e-3, 1e4, True)
    nonblind = P(10, 200, 10, 200, 1e9, 2, 1e-3, 1e3, False)
    r = nv.describe(255, 255, 15, blind)
    assert (r.conv_family, r.gradk_family, r.conv_fp16_split, r.gradk_fp16_split, r.graph) == (6, 8, 0, 0, 0)
    assert nv.describe(255, 255, 31, blind).conv_family == 6 and nv.describe(20, 24, 3, blind).conv_family == 6
    assert nv.describe(255, 255, 23, nonblind).conv_family == 6 and nv.describe(128, 128, 7, nonblind).conv_family == 6
    assert nv.describe(255, 255, 15, nonblind).conv_family != 6          # measured level with the multi-launch path: stays there
    assert nv.describe(512, 512, 9, blind).conv_family != 6              # 3 x tiles of 32 no longer fit the compute units
    assert nv.describe(255, 255, 33, blind).conv_family != 6             # built for PSF sizes up to 31
    for kw in (dict(conv=1), dict(conv=2), dict(tv_mode=2), dict(tv_mode=1), dict(fuse=1)):
        assert nv.describe(255, 255, 15, P(10, 200, 10, 200, 1e9, 2, 1e-3, 1e4, True, **kw)).conv_family != 6, kw
    debug_switch("small_iter", 0)
    assert nv.describe(255, 255, 15, blind).conv_family != 6


CASES = [  # M, N, MK, blind, correlation, outer iterations
    (255, 255, 15, True, 0, 2), (255, 255, 15, True, 1, 2), (200, 131, 5, True, 0, 3), (140, 140, 31, True, 0, 2), (140, 150, 23, False, 0, 3),
    (97, 64, 3, True, 0, 3), (20, 24, 9, True, 0, 2), (33, 40, 9, False, 0, 3), (255, 255, 7, True, 0, 2), (100, 260, 11, True, 0, 2)]


@pytest.mark.parametrize("M,N,MK,blind,corr,iters", CASES)
def test_whole_run_against_the_oracle(M, N, MK, blind, corr, iters):
    case = orc.synth_case(M, N, MK, seed=11, blind=blind)
    win = orc.default_window(M, N, MK) if min(M, N) > 3 * MK + 8 else (1, M - 1, 1, N - 1)
    lambd = 10000.0 if blind else 1000.0
    u_o, psf_o = case["u0"].copy(), case["psf0"].copy()
    tr = orc.Trace()
    with contextlib.redirect_stdout(io.StringIO()):
        orc.richardson_lucy_MM(case["image"].copy(), u_o, psf_o, *win, 1e9, M, N, 3, MK, iters, 1e-3, lambd, blind=blind, correlation=bool(corr), trace=tr, quiet=True)
    g = _run(case, M, N, MK, win, iters, blind, corr, lambd=lambd)
    assert g["route"].conv_family == 6 and g["st"].iterations_done == iters
    assert _rel(g["u"], u_o) <= TOL                     # (the whole padded frame, as the reference updates it in place)
    if blind:
        assert _rel(g["psf"], tr.psf_final) <= TOL
        assert _rel(g["psf_caller"], psf_o) <= TOL      # (pyx:585: with `correlation` the caller's array keeps the first step's values)
    for name in ("M_r", "Hu", "varu", "dof_min", "dof_max"):
        ref = np.array(getattr(tr, name), np.float64)
        assert np.allclose(g[name], ref, rtol=2e-3, atol=1e-7, equal_nan=True), (name, g[name], ref)


@pytest.mark.parametrize("M,N,MK,blind,corr", [(255, 255, 15, True, 0), (255, 255, 15, True, 1), (160, 120, 21, False, 0), (64, 255, 9, True, 0), (255, 255, 31, True, 0)])
def test_against_the_multi_launch_families(M, N, MK, blind, corr, debug_switch):
    case = orc.synth_case(M, N, MK, seed=5, blind=blind)
    win = orc.default_window(M, N, MK)
    out = {}
    for sw in (1, 0):
        debug_switch("small_iter", sw)
        out[sw] = _run(case, M, N, MK, win, 4, blind, corr, lambd=10000.0 if blind else 1000.0)
    a, b = out[1], out[0]
    assert a["route"].conv_family == 6 and b["route"].conv_family != 6
    assert _rel(a["u"], b["u"]) <= 5e-6 and _rel(a["psf"], b["psf"]) <= 5e-6 and _rel(a["psf_caller"], b["psf_caller"]) <= 5e-6
    for name in ("M_r", "Hu", "varu", "dof_min", "dof_max"):
        assert np.allclose(a[name], b[name], rtol=1e-3, atol=1e-8, equal_nan=True), name
    assert a["st"].inner_iterations == b["st"].inner_iterations == 20


def test_abort_drops_the_iteration_that_ran_ahead():
    """The statistics of outer iteration i run beside iteration i + 1; when the run ends at i (here: the progress callback asks for it at 4 of 30,
    with iteration 5 already queued), i + 1 is dropped: u comes back from the majoriser frame, the PSF from the copy the cooperative kernel took
    when it started (IcsSmallArgs::psf_bak).  The state must be that of a 4-iteration run, bit for bit.  (Stops by the reference's own rule on
    the goldens, same mechanism: tests/test_gpu_runtime.py::test_statistics_overlapped_with_the_next_iteration_change_nothing runs on this kernel.)"""
    M = N = 200; MK = 9
    case = orc.synth_case(M, N, MK, seed=2, blind=True)
    win = orc.default_window(M, N, MK)
    ref = _run(case, M, N, MK, win, 4, True)
    got = _run(case, M, N, MK, win, 30, True, progress=lambda it, *a: it == 4)
    assert ref["route"].conv_family == 6 and got["st"].stopped == 2 and got["st"].iterations_done == 4 and got["st"].inner_iterations == 20
    for k in ("u", "psf", "psf_caller"):
        assert np.array_equal(ref[k], got[k]), k


def test_two_runs_are_bit_identical():
    """every sum in the kernel has a fixed order (partial sums meet in group order, the tiles' gradient shares in tile order, maxima are order-free):
    results do not depend on which workgroup reaches a barrier first"""
    M, N, MK = 230, 255, 13
    case = orc.synth_case(M, N, MK, seed=9, blind=True)
    win = orc.default_window(M, N, MK)
    a = _run(case, M, N, MK, win, 6, True)
    b = _run(case, M, N, MK, win, 6, True)
    assert a["route"].conv_family == 6
    for k in ("u", "psf", "psf_caller", "M_r", "Hu", "varu", "dof_min", "dof_max"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_profile_counts_one_launch_per_outer_iteration():
    M = N = 255; MK = 15
    case = orc.synth_case(M, N, MK, seed=1, blind=True)
    g = _run(case, M, N, MK, orc.default_window(M, N, MK), 3, True, profile=1)
    from lib import _native as nv
    k = nv.KERNEL_NAMES.index("small_iteration")
    assert g["st"].launches[k] == 3 and g["st"].ms_kernel[k] > 0 and g["st"].inner_iterations == 15
    assert sum(g["st"].launches[i] for i in range(6)) == 0          # no convolution / update / PSF launches of the multi-launch families


def test_refused_cooperative_launch_falls_back_to_the_multi_launch_path(debug_switch):
    """hipLaunchCooperativeKernel can refuse a launch (another cooperative kernel's reservation, a device partition with fewer compute units):
    the job goes on with the multi-launch families from that outer iteration on -- weight tables repacked from the PSF as the cooperative
    launches left it -- and stays there."""
    from lib import _native as nv
    M = N = 160; MK = 11
    case = orc.synth_case(M, N, MK, seed=4, blind=True)
    win = orc.default_window(M, N, MK)
    ref = _run(case, M, N, MK, win, 5, True)
    job = nv.RLJob(M, N, MK)
    try:
        job.upload(case["image"], case["u0"], case["psf0"])
        p = job.params(*win, 1e9, 2, 1e-3, 10000.0, True, 0, 3, stop_test=2)
        job.run(p)                                              # two outer iterations on the cooperative kernel
        debug_switch("fail_small_launch", 1)
        p3 = job.params(*win, 1e9, 3, 1e-3, 10000.0, True, 0, 3, stop_test=2, profile=1)
        st = job.run(p3)                                        # the first launch of this call is refused
        assert st.iterations_done == 3 and st.launches[nv.KERNEL_NAMES.index("small_iteration")] == 0 and st.launches[nv.KERNEL_NAMES.index("update")] == 15
        assert job.describe(p3).conv_family != 6                # ... and the job stays on the multi-launch path
        u, psf, _ = job.download()
    finally:
        job.close()
    assert _rel(u, ref["u"]) <= 5e-6 and _rel(psf, ref["psf"]) <= 5e-6
