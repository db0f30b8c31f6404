"""Host runtime of ics_rl_run (csrc/ics_api.hip), round 4: the abort channel of the progress callback (ABI 4), one hipGraph
submission per outer iteration on small frames, recovery from a failed stats-window allocation, ics_rl_describe."""
import contextlib
import io

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _run(z, meta, iters, **kw):
    from lib import deconvolution as dc
    image, u, psf = z["image"].copy(), z["u0"].copy(), z["psf0"].copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dc.richardson_lucy_MM(image, u, psf, *meta["window"], meta["tau"], meta["M"], meta["N"], 3, meta["MK"], iters, meta["step"], meta["lambd"],
                              blind=meta["blind"], correlation=meta["corr"], **kw)
    return u, psf, buf.getvalue(), dc.richardson_lucy_MM.last


@pytest.mark.parametrize("name,stop_at", [("nb_129x129_k15", 10), ("bl_65x49_k9", 2)])
def test_interrupt_from_the_progress_callback_keeps_the_partial_result(golden_dir, monkeypatch, name, stop_at):
    """deconvolve.py:338-342 swallows a KeyboardInterrupt and keeps the partial, in-place-updated u.  A 200-iteration run is interrupted
    while the progress line of outer iteration `stop_at` is printed: the exception comes out of richardson_lucy_MM, and the caller's
    u / psf hold the state of exactly that iteration -- the reference golden of it."""
    from lib import deconvolution as dc
    z, meta = load_golden(golden_dir, name)
    want_u, want_psf, _, _ = _run(z, meta, stop_at)
    real = dc._progress
    seen = []

    def interrupting(it, *a):
        real(it, *a)
        seen.append(it)
        if it == stop_at:
            raise KeyboardInterrupt

    monkeypatch.setattr(dc, "_progress", interrupting)
    image, u, psf = z["image"].copy(), z["u0"].copy(), z["psf0"].copy()
    with pytest.raises(KeyboardInterrupt):
        with contextlib.redirect_stdout(io.StringIO()):
            dc.richardson_lucy_MM(image, u, psf, *meta["window"], meta["tau"], meta["M"], meta["N"], 3, meta["MK"], 200, meta["step"], meta["lambd"],
                                  blind=meta["blind"], correlation=meta["corr"])
    st = dc.richardson_lucy_MM.last
    assert seen == list(range(1, stop_at + 1))                 # the device loop stopped there, it did not run to 200
    assert st.iterations_done == stop_at and st.stopped == 2
    assert np.array_equal(u, want_u) and np.array_equal(psf, want_psf)
    assert rel_err(u, z["u_%d" % stop_at]) < 1e-4 and rel_err(psf, z["psf_%d" % stop_at]) < 1e-4     # the compiled reference at that iteration


def test_callback_return_value_stops_the_run_and_params_are_left_alone():
    from lib import _native as nv
    M, N, MK = 96, 80, 9
    case = orc.synth_case(M, N, MK, seed=2)
    job = nv.RLJob(M, N, MK)
    try:
        job.upload(case["image"], case["u0"], case["psf0"])
        p = job.params(*orc.default_window(M, N, MK), 1e9, 50, 1e-3, 1e4, False)
        sentinel = nv.PROGRESS_FN(lambda *a: 0)
        p.progress = sentinel
        calls = []
        st = job.run(p, progress=lambda it, *a: calls.append(it) or it == 3)
        assert calls == [1, 2, 3] and st.iterations_done == 3 and st.stopped == 2 and st.trace_len == 3
        assert nv.C.cast(p.progress, nv.C.c_void_p).value == nv.C.cast(sentinel, nv.C.c_void_p).value    # the caller's struct keeps its own callback
        st = job.run(p)                                            # ... which returns 0: the run completes
        assert st.iterations_done == 50 and st.stopped == 0

        class Boom(Exception):
            pass

        def bad(it, *a):
            raise Boom("in the callback")
        job.upload(case["image"], case["u0"], case["psf0"])
        with pytest.raises(Boom) as ei:
            job.run(p, progress=bad)
        assert ei.value.ics_stats.iterations_done == 1 and ei.value.ics_stats.stopped == 2
    finally:
        job.close()


@pytest.mark.parametrize("blind,tv_mode,MK", [(False, 0, 9), (True, 0, 9), (True, 0, 21), (False, 1, 7), (True, 2, 15), (True, 3, 15)])
def test_graph_replay_is_bit_identical_to_eager_launches(blind, tv_mode, MK, debug_switch):
    """One hipGraph launch per outer iteration (use_graph, csrc/ics_api.hip): 8 outer iterations cross the three-frame rotation of
    u / ut / spare twice, so every captured executable is replayed at least once; u, psf, the traces and the stop-test scalars must
    be those of the eager launches bit for bit.  Then a second call on the same job with other parameters (lambd): the executables
    of the first parameter set must not be replayed."""
    from lib import _native as nv
    M, N = 150, 131
    case = orc.synth_case(M, N, MK, seed=7, blind=blind)
    win = orc.default_window(M, N, MK)
    out = {}
    debug_switch("small_iter", 0)      # (graphs belong to the multi-launch families: the cooperative small-frame kernel is one launch per outer iteration already)
    for graph in (0, 1):
        debug_switch("graph", graph)
        job = nv.RLJob(M, N, MK)
        try:
            res = []
            for lambd in (1e4, 3e3):
                job.upload(case["image"], case["u0"], case["psf0"])
                p = job.params(*win, 1e9, 8, 1e-3, lambd, blind, tv_mode=tv_mode, stop_test=2)      # (evaluate the stop test, never stop: all 8 iterations run)
                assert job.describe(p).graph == graph
                st = job.run(p)
                u, psf, _ = job.download()
                res.append((u, psf, np.array(st.trace_M_r[:8]), np.array(st.trace_Hu[:8]), np.array(st.trace_dof_max[:8]), st.inner_iterations))
            out[graph] = res
        finally:
            job.close()
    for a, b in zip(out[0], out[1]):
        assert a[5] == b[5] == 40
        for x, y in zip(a[:5], b[:5]):
            assert np.array_equal(x, y, equal_nan=True)
    assert not np.array_equal(out[1][0][0], out[1][1][0])       # the second parameter set did change the result


def test_stats_window_allocation_failure_leaves_the_job_usable(debug_switch):
    """ensure_window freed its buffers, failed to reallocate, and kept the OLD window key -- the next run with the
    previous window passed the cache check and launched the statistics kernels on freed / NULL buffers.  Window A, a failing window B
    (test hook: the n-th allocation fails once), window A again."""
    from lib import _native as nv
    M, N, MK = 120, 100, 9
    case = orc.synth_case(M, N, MK, seed=3)
    job = nv.RLJob(M, N, MK)
    try:
        def run(win):
            job.upload(case["image"], case["u0"], case["psf0"])
            st = job.run(job.params(*win, 1e9, 3, 1e-3, 1e4, False))
            return st.M_r, st.Hu, st.varu
        A, B = (10, 71, 8, 61), (5, 100, 5, 90)
        first = run(A)
        for nth in (1, 2, 3):
            debug_switch("fail_window_alloc", nth)
            with pytest.raises(nv.NativeError) as ei:
                run(B)
            assert ei.value.code == nv.ICS_ENOMEM
            assert run(A) == first
        assert run(B) != first and run(A) == first
    finally:
        job.close()


def test_job_describe_equals_shape_describe():
    from lib import _native as nv
    for M, MK, blind, conv in [(200, 15, True, 0), (200, 31, True, 0), (300, 45, True, 0), (200, 65, False, 0), (200, 15, True, 1)]:
        job = nv.RLJob(M, M, MK)
        try:
            p = job.params(1, 100, 1, 100, 1e9, 1, 1e-3, 1e4, blind, conv=conv)
            a, b = job.describe(p), nv.describe(M, M, MK, p)
            assert [getattr(a, f) for f, _ in nv.RLRoute._fields_] == [getattr(b, f) for f, _ in nv.RLRoute._fields_]
        finally:
            job.close()


@pytest.mark.parametrize("name,iters", [("nb_97x97_k5_tau", 40), ("bl_129x129_k15", 5), ("bl_65x65_k7_corr", 3), ("nb_129x129_k15", 20), ("bl_65x49_k9", 10)])
def test_statistics_overlapped_with_the_next_iteration_change_nothing(golden_dir, debug_switch, name, iters):
    """ics_rl_run queues outer iteration i + 1 before the stop-test scalars of iteration i are on the host (second stream; residual
    frame, reduction slots and DoF keys in two sets) and, when the stop test fires at i, drops iteration i + 1 again (u from the
    majoriser frame, PSF from its copy).  Against the loop that drains at every outer boundary (debug switch overlap = 0): same
    iterations done, same stop flag, u / PSF / every trace bit for bit -- on goldens that stop early (tau = 0 after 3 ... 40 iterations,
    blind M_r > M_r_prev), that run to the end, and with the correlation quirk (the caller's PSF frozen after the first step)."""
    z, meta = load_golden(golden_dir, name)
    res = {}
    for ov in (0, 2, 3):
        debug_switch("overlap", ov)
        u, psf, log, st = _run(z, meta, iters)
        res[ov] = (u, psf, log, st.iterations_done, st.stopped, st.inner_iterations, np.array(st.trace_M_r[:st.trace_len]), np.array(st.trace_Hu[:st.trace_len]),
                   np.array(st.trace_varu[:st.trace_len]), np.array(st.trace_dof_min[:st.trace_len]), np.array(st.trace_dof_max[:st.trace_len]), st.M_r, st.Hu, st.varu)
    a = res[0]
    print("%s: %d of %d outer iterations, stopped = %d" % (name, a[3], iters, a[4]))
    for b in (res[2], res[3]):       # 2: statistics on the second stream; 3: look-ahead on the job's own stream
        assert a[3:6] == b[3:6] and a[5] == 5 * a[3]
        assert a[2] == b[2]                                                  # the printed lines
        for x, y in zip(a[:2] + a[6:11], b[:2] + b[6:11]):
            assert np.array_equal(x, y, equal_nan=True)
        assert a[11:] == b[11:] or all(np.isnan(v) for v in a[11:] + b[11:])


def test_abort_under_overlap_drops_the_iteration_that_was_ahead(debug_switch):
    """the callback asks to stop at outer iteration 4 of 30 while iteration 5 is already queued: u and PSF must be those of a 4-iteration run"""
    from lib import _native as nv
    M, N, MK = 120, 100, 9
    case = orc.synth_case(M, N, MK, seed=11, blind=True)
    win = orc.default_window(M, N, MK)
    out = {}
    for mode in ("ref4", "abort"):
        job = nv.RLJob(M, N, MK)
        try:
            job.upload(case["image"], case["u0"], case["psf0"])
            if mode == "ref4":
                debug_switch("overlap", 0)
                st = job.run(job.params(*win, 1e9, 4, 1e-3, 1e4, True, stop_test=2))
            else:
                debug_switch("overlap", 2)
                st = job.run(job.params(*win, 1e9, 30, 1e-3, 1e4, True, stop_test=2), progress=lambda it, *a: it == 4)
                assert st.stopped == 2
            assert st.iterations_done == 4 and st.inner_iterations == 20
            out[mode] = job.download()
        finally:
            job.close()
    for x, y in zip(out["ref4"], out["abort"]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("blind,tv_mode", [(False, 0), (True, 0), (True, 3)])
def test_long_overlapped_run_equals_the_drained_one(debug_switch, blind, tv_mode):
    """150 outer iterations (750 inner) with the statistics one iteration behind the kernels, against the drained loop: the two sets of
    reduction slots / DoF keys and the residual ping-pong are each used 75 times, the frame rotation 50 times"""
    from lib import _native as nv
    M, N, MK = 140, 123, 11
    case = orc.synth_case(M, N, MK, seed=5, blind=blind)
    win = orc.default_window(M, N, MK)
    out = {}
    for ov in (0, 2, 3):
        debug_switch("overlap", ov)
        job = nv.RLJob(M, N, MK)
        try:
            job.upload(case["image"], case["u0"], case["psf0"])
            st = job.run(job.params(*win, 1e9, 150, 1e-4, 1e4, blind, tv_mode=tv_mode, stop_test=2))
            assert st.iterations_done == 150 and st.inner_iterations == 750
            out[ov] = job.download() + (np.array(st.trace_M_r[:150]), np.array(st.trace_dof_max[:150]), np.array(st.trace_varu[:150]))
        finally:
            job.close()
    for ov in (2, 3):
        for x, y in zip(out[0], out[ov]):
            assert np.array_equal(x, y, equal_nan=True)
