"""GPU parity, stage by stage, through the C ABI (ics_rl_stage / ics_rl_read).

Each stage of one inner iteration (lib/deconvolution.pyx:473-591) is compared with the oracle on the
same inputs ("teacher forcing": the inputs of a stage are what the device holds).  Convolutions are
checked against float64 direct sums (tolerance 5e-6 relative: sequential fp32 accumulation of up to 31*31 = 961 terms);
the elementwise stages reproduce the reference's rounding exactly and are checked bit for bit.
"""
import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import conv_valid64, corr_full64, gradk64, psf_step_f32, rel_err, update_f32

pytestmark = pytest.mark.gpu

CONV_TOL = 5e-6   # K <= 31; larger PSFs scale it with the number of accumulated terms


def make_job(M, N, MK, seed=0, blind=False, per_channel_psf=True):
    from lib import _native
    case = orc.synth_case(M, N, MK, seed=seed, blind=blind, per_channel_psf=per_channel_psf)
    rng = np.random.default_rng(seed + 1)
    # a PSF without symmetry so that flips / transposes are detected
    psf = (case["psf0"] * (0.5 + rng.random(case["psf0"].shape, dtype=np.float32))).astype(np.float32)
    orc.normalize_kernel(psf, MK)
    job = _native.RLJob(M, N, MK)
    job.upload(case["image"], case["u0"], psf)
    return job, case, psf


CONV_CASES = [(MK, 1) for MK in (3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 33, 39, 45, 55, 63)] + \
             [(MK, 2) for MK in (3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 33, 35, 37, 39, 41, 43, 45, 47, 49)]


@pytest.mark.parametrize("MK,conv", CONV_CASES)
def test_synth_residual_and_backprojection_all_psf_sizes(MK, conv):
    """conv = 1: packed-fp32 vector kernels (ics_conv.hip); conv = 2: matrix-core kernels with fp16-split
    operands (ics_conv_mfma.hip, MK <= 49).  Same tolerance for both against float64 direct sums."""
    from lib import _native as nv
    M, N = 70 + MK, 131
    job, case, psf = make_job(M, N, MK, seed=MK)
    # perturb u so that it is not just the padded image
    rng = np.random.default_rng(7)
    u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=False, conv=conv)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    e_ref = conv_valid64(u, psf) - case["image"]
    scale = np.max(np.abs(conv_valid64(u, psf)))
    tol = CONV_TOL * max(1.0, (MK / 31.0) ** 2)
    assert np.max(np.abs(e - e_ref)) / scale < tol
    job.stage(nv.STAGE_BACKPROJECT, p)
    g = job.read(nv.BUF_GRADU)
    g_ref = corr_full64(e.astype(np.float64), psf)
    assert g.shape == g_ref.shape
    assert rel_err(g, g_ref) < tol
    job.close()


@pytest.mark.parametrize("MK", [3, 9, 13])
@pytest.mark.parametrize("rs", ["2", "4", "1"])
def test_matrix_core_convolution_both_tile_heights(MK, rs, debug_switch):
    """ics_conv_mfma.hip builds 64-row and 32-row tiles for K <= 13 and picks by frame size (32-row up to 3000 tiles of
    64 x 64; round 4: 16-row tiles, K <= 15, where the frame has fewer 32-row tiles than compute units): the debug switch conv_rs forces
    any of them, on a frame several tiles high and wide with ragged edges."""
    from lib import _native as nv
    debug_switch("conv_rs", int(rs))
    M, N = 203, 277
    job, case, psf = make_job(M, N, MK, seed=MK + 40)
    rng = np.random.default_rng(9)
    u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=False, conv=2)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    full = conv_valid64(u, psf)
    assert np.max(np.abs(e - (full - case["image"]))) / np.max(np.abs(full)) < CONV_TOL
    job.stage(nv.STAGE_BACKPROJECT, p)
    assert rel_err(job.read(nv.BUF_GRADU), corr_full64(e.astype(np.float64), psf)) < CONV_TOL
    red = job.red_keys()
    job.close()
    # the step-size reductions of the back-projection do not depend on the tiling (maxima)
    debug_switch("conv_rs", 4 if rs in ("2", "1") else 2)
    job2, _, _ = make_job(M, N, MK, seed=MK + 40)
    job2.write(nv.BUF_U, u)
    job2.write(nv.BUF_UT, case["u0"])
    job2.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    job2.stage(nv.STAGE_BACKPROJECT, p)
    red2 = job2.red_keys()
    job2.close()
    assert np.array_equal(red[3:6], red2[3:6])          # max u: exact
    assert np.all(np.abs(red[:3].astype(np.int64) - red2[:3].astype(np.int64)) < 64)   # max |g|: per-tile scales differ in the last bits


# conv: 0 = ICS_CONV_AUTO, 1 = fp32 products (vector convolutions, fp32-MFMA gradient), 2 = matrix-core kernels
@pytest.mark.parametrize("M,N,MK,blind,conv", [(64, 64, 15, False, 0), (65, 191, 15, False, 0), (130, 67, 9, True, 0), (257, 300, 15, True, 0),
                                               (40, 50, 31, False, 0), (100, 90, 45, True, 0), (80, 120, 63, True, 0), (70, 70, 21, True, 0),
                                               (257, 300, 15, True, 1), (130, 67, 9, True, 1), (70, 70, 21, True, 2), (90, 100, 31, True, 2),
                                               (66, 70, 17, True, 2), (97, 133, 27, True, 0)])
def test_one_inner_iteration_stage_by_stage(M, N, MK, blind, conv):
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=M + N, blind=blind)
    pad = MK // 2
    step, lambd = 1e-3, 10000.0
    rng = np.random.default_rng(3)
    u = (case["u0"] + 0.02 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    ut = case["u0"]
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_UT, ut)
    win = orc.default_window(M, N, MK)
    p = job.params(*win, 1e9, 1, step, lambd, blind=blind, conv=conv)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    job.stage(nv.STAGE_BACKPROJECT, p)
    g_raw = job.read(nv.BUF_GRADU)
    job.stage(nv.STAGE_UPDATE, p)
    u_dev = job.read(nv.BUF_U)
    sc = job.scalars()
    u_ref, dt, DoF = update_f32(u, ut, g_raw, case["image"], step, lambd, blind, pad)
    # reductions (A7) and the step size are exact
    for k in range(3):
        assert sc["maxu%d" % k] == np.amax(u[..., k])
        assert sc["dt%d" % k] == dt[k]
    assert np.array_equal(u_dev, u_ref, equal_nan=True), "A5-A10 must reproduce the reference's float32 rounding bit for bit"
    if blind:
        job.stage(nv.STAGE_SYNTH_RESIDUAL, p)       # A11 on the updated u
        e2 = job.read(nv.BUF_ERROR)
        job.stage(nv.STAGE_PSF_GRADIENT, p)
        gk = job.read(nv.BUF_GRADK)
        gk_ref = gradk64(u_dev.astype(np.float64), e2.astype(np.float64))
        assert rel_err(gk, gk_ref) < 1e-5
        for correlation in (0,):
            job.stage(nv.STAGE_PSF_UPDATE, p)
            psf_dev = job.read(nv.BUF_PSF)
            psf_ref, _caller, dtpsf = psf_step_f32(psf, gk, step, MK, correlation)
            assert sc is not None
            assert np.array_equal(psf_dev, psf_ref), "A14-A17 must be bit exact given the device's gradk"
            assert job.scalars()["dtpsf"] == dtpsf
    job.close()


def test_psf_update_correlation_quirk():
    """pyx:584-585: with correlation the local psf is tied across channels and the caller's array only
    receives the first gradient step."""
    from lib import _native as nv
    M, N, MK = 90, 70, 7
    job, case, psf = make_job(M, N, MK, seed=11, blind=True)
    p = job.params(*orc.default_window(M, N, MK), 0.0, 1, 1e-3, 10000.0, blind=True, correlation=1)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk = job.read(nv.BUF_GRADK)
    job.stage(nv.STAGE_PSF_UPDATE, p)
    _, psf_local, psf_caller = job.download()
    ref_local, ref_caller, _ = psf_step_f32(psf, gk, 1e-3, MK, 1)
    assert np.array_equal(psf_local, ref_local)
    assert np.array_equal(psf_caller, ref_caller)
    # a second step moves the local psf but not the caller's copy
    job.stage(nv.STAGE_PSF_UPDATE, p)
    _, psf_local2, psf_caller2 = job.download()
    assert np.array_equal(psf_caller2, ref_caller)
    assert not np.array_equal(psf_local2, psf_local)
    job.close()


@pytest.mark.parametrize("M,N,MK,win", [(129, 129, 15, (8, 119, 8, 119)), (300, 280, 9, (5, 250, 5, 250)), (65, 49, 9, (5, 42, 5, 42)),
                                        (1300, 1250, 5, (20, 1221, 7, 1108)), (2300, 400, 3, (100, 2249, 30, 331))],
                         ids=["111px", "245px", "37px", "1201x1101px-P4096", "2149x301px-P8192"])
def test_window_statistics_and_whiteness_metric(M, N, MK, win):
    """A18/A19 on device (FFT autocorrelation) vs the oracle's numpy/scipy evaluation.  The last two windows are wider than 1024 px
    (4096- and 8192-point transforms: several butterflies per thread, 128 KB of LDS) -- the reference has no limit on `mask_size`
    (deconvolve.py:67, lib/deconvolution.pyx:623-638), round 2 stopped at 1024 px."""
    from lib import _native as nv
    if M > 1000:
        case = orc.synth_case_large(M, N, MK, seed=5)
        job = nv.RLJob(M, N, MK)
        job.upload(case["image"], case["u0"], case["psf0"])
    else:
        job, case, psf = make_job(M, N, MK, seed=5)
    pad = MK // 2
    p = job.params(*win, 1e9, 1, 1e-3, 10000.0, blind=False)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    u = job.read(nv.BUF_U)
    job.stage(nv.STAGE_STATS, p)
    sc = job.scalars()
    top, bottom, left, right = win
    ew = e[top:bottom, left:right]
    M_r = orc.residual_whiteness(ew, orc.stop_weights(*win), orc._conv_scipy)
    Hu = np.linalg.norm(ew) ** 2 / ((bottom - top) * (right - left) * 3)
    varu = np.std(u[top + pad:bottom - pad, left + pad:right - pad]) ** 2
    assert abs(sc["M_r"] - M_r) / M_r < 2e-4
    assert abs(sc["Hu"] - Hu) / Hu < 1e-5
    assert abs(sc["varu"] - varu) / varu < 1e-5
    job.close()


@pytest.mark.parametrize("M,N,MK,blind", [(64, 64, 15, False), (97, 191, 15, True), (130, 67, 9, False), (50, 45, 3, True), (70, 90, 31, False)])
def test_fused_update_synth_equals_separate_kernels(M, N, MK, blind):
    """ICS_STAGE_UPDATE_SYNTH (update recomputed on the halo while staging + convolution, u ping-pong) must
    give bit-identical u, error, dt and DoF extrema to ICS_STAGE_UPDATE followed by ICS_STAGE_SYNTH_RESIDUAL."""
    from lib import _native as nv
    res = []
    for fused in (False, True):
        job, case, psf = make_job(M, N, MK, seed=M * 3 + N, blind=blind)
        rng = np.random.default_rng(9)
        u = (case["u0"] + 0.03 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
        job.write(nv.BUF_U, u)
        job.write(nv.BUF_UT, case["u0"])
        p = job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=blind, conv=nv.CONV_VECTOR)
        job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
        job.stage(nv.STAGE_BACKPROJECT, p)
        if fused:
            job.stage(nv.STAGE_UPDATE_SYNTH, p)
        else:
            job.stage(nv.STAGE_UPDATE, p)
            job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
        job.stage(nv.STAGE_STATS, job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=blind, stop_test=0))
        sc = job.scalars()
        res.append((job.read(nv.BUF_U), job.read(nv.BUF_ERROR), [sc[k] for k in ("dt0", "dt1", "dt2", "dof_min", "dof_max")]))
        job.close()
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True)
    assert np.array_equal(res[0][1], res[1][1], equal_nan=True)
    assert res[0][2] == res[1][2]


@pytest.mark.parametrize("M,N,MK", [(64, 64, 15), (257, 300, 15), (130, 67, 9), (200, 333, 3), (97, 133, 5), (191, 129, 7), (150, 150, 11),
                                    (300, 260, 13), (640, 700, 15)])
def test_fused_synth_gradk_stage(M, N, MK):
    """ICS_STAGE_SYNTH_GRADK (ics_synth_gradk_mfma.hip: A11 + A13 in one kernel, e' on chip) against float64 direct sums, and
    against the two-kernel path it replaces: the residual it stores (whole frame in the stage call) within the convolution
    gate, gradk within the PSF-gradient gate of the separate kernels (1e-5 of max|gradk|)."""
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=M + 2 * N + MK, blind=True)
    rng = np.random.default_rng(5)
    u = (case["u0"] + 0.02 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    p = job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=True)
    job.stage(nv.STAGE_SYNTH_GRADK, p)
    e = job.read(nv.BUF_ERROR)
    gk = job.read(nv.BUF_GRADK)
    synth = conv_valid64(u, psf)
    e_ref = synth - case["image"]
    assert np.max(np.abs(e - e_ref)) / np.max(np.abs(synth)) < CONV_TOL
    gk_ref = gradk64(u.astype(np.float64), e.astype(np.float64))      # teacher-forced on the device's own residual
    assert rel_err(gk, gk_ref) < 1e-5
    # the two-kernel path on the same inputs
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e2 = job.read(nv.BUF_ERROR)
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk2 = job.read(nv.BUF_GRADK)
    assert np.max(np.abs(e - e2)) / np.max(np.abs(synth)) < 1e-6
    assert rel_err(gk, gk2) < 1e-5
    job.close()


def test_fused_synth_gradk_in_the_loop_matches_two_kernel_path():
    """Whole blind runs with and without the fused kernel (ICS_FLAG_NO_FUSED_GRADK): same u / psf within 1e-5, same
    statistics (the fused kernel stores e' only on the tiles of the stats window)."""
    from lib import _native as nv
    M, N, MK = 300, 333, 15
    out = []
    for flags in (0, nv.FLAG_NO_FUSED_GRADK):
        job, case, psf = make_job(M, N, MK, seed=21, blind=True)
        p = job.params(40, 295, 50, 305, 1e9, 3, 1e-3, 10000.0, blind=True, flags=flags, stop_test=2, profile=1)
        st = job.run(p)
        u, psf_l, _ = job.download()
        out.append((u, psf_l, st.M_r, st.Hu, st.varu, st.launches[8]))
        job.close()
    assert out[0][5] == 15 and out[1][5] == 0          # the fused kernel ran (or not) as requested
    assert rel_err(out[0][0], out[1][0]) < 1e-5 and rel_err(out[0][1], out[1][1]) < 1e-5
    for k in (2, 3, 4):
        assert abs(out[0][k] - out[1][k]) <= 2e-4 * abs(out[1][k])


def test_matrix_core_kernels_equal_the_vector_kernels_on_random_shapes():
    """Ragged frames (1 .. 200 px a side, any tile remainder) x odd PSF sizes 3 .. 37: the matrix-core convolutions (both tile
    heights, the 8-wave form at K >= 23) and the packed-fp32 ones agree within the stage tolerance on residual and
    back-projection, and bit for bit on the max-u reduction.  Seeded hypothesis run, 30 cases."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from lib import _native as nv

    @settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.integers(1, 200), st.integers(1, 200), st.integers(1, 18), st.integers(0, 2 ** 31 - 1))
    def check(M, N, kh, seed):
        MK = 2 * kh + 1
        job, case, psf = make_job(M, N, MK, seed=seed % 1000)
        rng = np.random.default_rng(seed)
        u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
        out = {}
        job.write(nv.BUF_U, u)
        job.write(nv.BUF_UT, case["u0"])
        res = {}
        for conv in (1, 2):
            job.stage(nv.STAGE_SYNTH_RESIDUAL, job.params(0, 1, 0, 1, 1e9, 1, 1e-3, 10000.0, blind=False, conv=conv))
            res[conv] = job.read(nv.BUF_ERROR)
        for conv in (1, 2):   # both back-project the SAME residual (it is a small difference of large numbers)
            job.write(nv.BUF_ERROR, res[1])
            job.stage(nv.STAGE_BACKPROJECT, job.params(0, 1, 0, 1, 1e9, 1, 1e-3, 10000.0, blind=False, conv=conv))
            out[conv] = (res[conv], job.read(nv.BUF_GRADU), job.red_keys()[3:6].copy())
        job.close()
        tol = 2 * CONV_TOL * max(1.0, (MK / 31.0) ** 2)
        scale = max(float(np.max(np.abs(u))), 1e-6)
        assert np.max(np.abs(out[1][0] - out[2][0])) <= tol * scale, (M, N, MK)
        assert np.max(np.abs(out[1][1] - out[2][1])) <= tol * max(float(np.max(np.abs(out[1][1]))), 1e-6), (M, N, MK)
        assert np.array_equal(out[1][2], out[2][2]), (M, N, MK)

    check()


def test_fused_synth_gradk_equals_the_two_kernel_path_on_random_shapes():
    """The fused A11 + A13 kernel against the two kernels it replaces over ragged frames (1 .. 260 px a side) and every PSF size
    it is built for (3 .. 15), with 1 .. 7 persistent workgroups (debug switch max_wgs: tile walk, next-tile prefetch, partial blocks).
    Seeded hypothesis run, 30 cases."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from lib import _native as nv

    @settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.integers(1, 260), st.integers(1, 260), st.integers(1, 7), st.integers(0, 7), st.integers(0, 2 ** 31 - 1))
    def check(M, N, kh, wgs, seed):
        MK = 2 * kh + 1
        # (the workgroup count of the gradient kernels is fixed when the job is created: the switch goes first)
        old = nv.debug_set("max_wgs", wgs)
        try:
            job, case, psf = make_job(M, N, MK, seed=seed % 1000, blind=True)
            rng = np.random.default_rng(seed)
            u = (case["u0"] + 0.02 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
            job.write(nv.BUF_U, u)
            p = job.params(0, M, 0, N, 1e9, 1, 1e-3, 10000.0, blind=True)
            job.stage(nv.STAGE_SYNTH_GRADK, p)
            e, gk = job.read(nv.BUF_ERROR), job.read(nv.BUF_GRADK)
        finally:
            nv.debug_set("max_wgs", old)
        job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
        e2 = job.read(nv.BUF_ERROR)
        job.write(nv.BUF_ERROR, e)                                         # the gradient of the same residual
        job.stage(nv.STAGE_PSF_GRADIENT, p)
        gk2 = job.read(nv.BUF_GRADK)
        job.close()
        scale = max(float(np.max(np.abs(u))), 1e-6)
        assert np.max(np.abs(e - e2)) <= 2e-6 * scale, (M, N, MK, wgs)
        ref = gradk64(u.astype(np.float64), e.astype(np.float64))
        gs = max(float(np.max(np.abs(ref))), 1e-12)
        assert np.max(np.abs(gk - ref)) <= 2e-5 * gs and np.max(np.abs(gk2 - ref)) <= 2e-5 * gs, (M, N, MK, wgs)

    check()


@pytest.mark.parametrize("MK,M,N", [(15, 700, 200), (31, 520, 150), (23, 300, 330)])
def test_two_kernel_gradient_strip_carry_is_race_free_and_deterministic(MK, M, N, debug_switch):
    """k_gradk_mfma walks down 64-column strips and keeps the NT - 1 shared rows of u in LDS between tiles (the carry).  Round-2
    advice: without a barrier between the last carry pass and the conversion of the new rows a fast wave could overwrite rows
    a lagging wave had not carried yet.  With 2 / 3 persistent workgroups every workgroup walks tens of tiles of tall narrow
    frames (the carry runs on nearly every tile; K = 23, 31 carry rows through two tiles); the gradient must equal float64 and be
    bit-identical over 12 repetitions -- a race shows up as a run that differs."""
    from lib import _native as nv
    results = []
    for wgs in (2, 3):
        debug_switch("max_wgs", wgs)
        job, case, psf = make_job(M, N, MK, seed=MK, blind=True)
        rng = np.random.default_rng(MK)
        u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
        u[40, 33, 1] = 50.0                                 # a bright pixel near the top of a strip: the carried scale bound must not stick
        job.write(nv.BUF_U, u)
        p = job.params(1, 9, 1, 9, 1e9, 1, 1e-3, 10000.0, blind=True, conv=2, flags=nv.FLAG_NO_FUSED_GRADK)
        job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
        e = job.read(nv.BUF_ERROR)
        ref = gradk64(u.astype(np.float64), e.astype(np.float64))
        first = None
        for rep in range(12):
            job.stage(nv.STAGE_PSF_GRADIENT, p)
            gk = job.read(nv.BUF_GRADK)
            if first is None:
                first = gk
                assert rel_err(gk, ref) < 1e-5, (MK, wgs, rel_err(gk, ref))
            assert np.array_equal(gk, first), (MK, wgs, rep)
        results.append(first)
        job.close()
    assert rel_err(results[0], results[1]) < 1e-5


@pytest.mark.parametrize("M,N,MK", [(64, 64, 15), (257, 300, 15), (130, 67, 9), (200, 333, 3), (300, 260, 13), (640, 700, 15)])
def test_fused_synth_gradk_32_row_form(M, N, MK, debug_switch):
    """k_synth_gradk2 (32-row tiles, three workgroups per CU; debug switch fused_rs = 2) against float64 and against the 64-row form
    that is the default: same residual within the convolution gate, same gradient within 1e-5 of max |gradk|."""
    from lib import _native as nv
    out = {}
    for rs in (2, 4):
        debug_switch("fused_rs", rs)
        job, case, psf = make_job(M, N, MK, seed=M + 2 * N + MK, blind=True)
        rng = np.random.default_rng(5)
        u = (case["u0"] + 0.02 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
        job.write(nv.BUF_U, u)
        p = job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=True)
        job.stage(nv.STAGE_SYNTH_GRADK, p)
        out[rs] = (job.read(nv.BUF_ERROR), job.read(nv.BUF_GRADK))
        job.close()
    synth = conv_valid64(u, psf)
    e_ref = synth - case["image"]
    for rs in (2, 4):
        e, gk = out[rs]
        assert np.max(np.abs(e - e_ref)) / np.max(np.abs(synth)) < CONV_TOL, rs
        assert rel_err(gk, gradk64(u.astype(np.float64), e.astype(np.float64))) < 1e-5, rs
    assert rel_err(out[2][1], out[4][1]) < 1e-5


@pytest.mark.parametrize("rs", [2, 4])
def test_blind_run_with_either_form_of_the_fused_kernel_matches_the_reference_golden(golden_dir, debug_switch, rs):
    """the 129 x 129 blind golden of the compiled reference with the 32-row and with the 64-row form of the fused A11 + A13 kernel forced
    (the launcher picks 32-row tiles up to ~3200^2 since round 4, 64-row tiles above) and few workgroups, so that each walks several tiles"""
    import contextlib
    import io
    from helpers import load_golden
    from lib import deconvolution as dc
    z, meta = load_golden(golden_dir, "bl_129x129_k15")
    debug_switch("fused_rs", rs)
    debug_switch("max_wgs", 3)
    dc._drop_jobs()
    n = meta["snaps"][1]
    u, psf = z["u0"].copy(), z["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(z["image"].copy(), u, psf, *meta["window"], meta["tau"], meta["M"], meta["N"], 3, meta["MK"], n, meta["step"], meta["lambd"], blind=True)
    assert rel_err(u, z["u_%d" % n]) < 1e-5 and rel_err(psf, z["psf_%d" % n]) < 1e-5
    dc._drop_jobs()


@pytest.mark.parametrize("MK,flags", [(31, 0), (23, 0), (15, "no_fused"), (9, "no_fused")])
@pytest.mark.parametrize("zero_rows,scale", [(70, 0.9), (100, 0.2), (70, 1e-6)])
def test_gradient_strip_carry_below_an_all_zero_tile(MK, flags, zero_rows, scale, debug_switch):
    """k_gradk_mfma carries the rows two consecutive tiles of a strip share and rescales them by the ratio of the tiles' power-of-two
    scales.  A black band at the top of u (all-zero tiles, scale 1) followed by values below 0.5 (scale 2^16 and up) made that ratio
    overflow fp16: inf * 0 = NaN in every carried zero, i.e. a NaN PSF from an image with a black border (round 3; found by the
    differential fuzz once the split gradient put all-zero tiles at the head of every strip).  Few workgroups, so that strips are walked."""
    from lib import _native as nv
    debug_switch("max_wgs", 3)
    M, N = 230, 150
    job, case, psf = make_job(M, N, MK, seed=MK, blind=True)
    rng = np.random.default_rng(3)
    u = (case["u0"] * np.float32(scale)).astype(np.float32)
    u[:zero_rows] = 0.0
    job.write(nv.BUF_U, u)
    e = np.zeros((M, N, 3), np.float32)
    e[:] = (0.01 * scale * rng.standard_normal((M, N, 3))).astype(np.float32)
    job.write(nv.BUF_ERROR, e)
    p = job.params(2, M - 2, 2, N - 2, 1e9, 1, 1e-3, 10000.0, blind=True, flags=nv.FLAG_NO_FUSED_GRADK if flags else 0)
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk = job.read(nv.BUF_GRADK)
    assert np.isfinite(gk).all()
    assert rel_err(gk, gradk64(u.astype(np.float64), e.astype(np.float64))) < 1e-5
    job.close()
