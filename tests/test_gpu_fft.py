"""The transform-tile pipeline (csrc/ics_conv_fft.hip, ics_planar.hip; conv = ICS_CONV_FFT): A1 / A3 / A11 as LDS-resident 128 x 128
overlap-save FFTs on channel-planar mirrors of the frames, the update pass and the matrix-core PSF gradient on the mirrors as well.

  * the two convolutions against float64 direct sums at the gate of every other convolution path (5e-6 of the largest convolution
    value; measured 2 - 5e-7), PSF sizes 3 ... 65, frames of one tile and of several ragged ones;
  * the update pass on planes bit-identical to the HWC pass, the planar PSF gradient bit-identical to the HWC one (same kernels'
    arithmetic, other loads);
  * whole runs: every reference golden of tests/test_gpu_rl.py and the deep goldens at BASELINE sizes run with conv = 3 there;
    here the routing (ICS_CONV_AUTO picks the tiles for wide PSFs on big frames, describe says so) and the step-size maxima.
"""
import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import conv_valid64, corr_full64, gradk64, rel_err, update_f32

pytestmark = pytest.mark.gpu

CONV_TOL = 5e-6
FFT = 3


def make_job(M, N, MK, seed=0, blind=False):
    from lib import _native
    case = orc.synth_case(M, N, MK, seed=seed, blind=blind, per_channel_psf=True)
    rng = np.random.default_rng(seed + 1)
    psf = (case["psf0"] * (0.5 + rng.random(case["psf0"].shape, dtype=np.float32))).astype(np.float32)   # no symmetry: flips show
    orc.normalize_kernel(psf, MK)
    job = _native.RLJob(M, N, MK)
    job.upload(case["image"], case["u0"], psf)
    return job, case, psf


@pytest.mark.parametrize("M,N,MK", [(40, 50, 3), (70, 131, 9), (90, 100, 15), (114, 114, 15), (115, 229, 15), (150, 260, 17), (99, 197, 31),
                                    (200, 120, 31), (84, 169, 45), (130, 70, 63), (64, 129, 65), (300, 310, 23),
                                    (300, 100, 5), (260, 99, 9), (176, 108, 17), (400, 60, 31),      # (these four: ONE tile column, several tile rows)
                                    (150, 170, 67), (200, 130, 85), (120, 140, 97), (100, 120, 129)])   # (round 6: 67 ... 85 in one tile of 62 ... 44 valid pixels a side, above as tap blocks)
def test_fft_convolutions_against_float64(M, N, MK):
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=MK + M)
    rng = np.random.default_rng(7)
    u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=False, conv=FFT)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    full = conv_valid64(u, psf)
    err_e = np.max(np.abs(e - (full - case["image"]))) / np.max(np.abs(full))
    job.stage(nv.STAGE_BACKPROJECT, p)
    g = job.read(nv.BUF_GRADU)
    g_ref = corr_full64(e.astype(np.float64), psf)
    err_g = rel_err(g, g_ref)
    print("%dx%d K=%d: residual %.2e back-projection %.2e" % (M, N, MK, err_e, err_g))
    assert err_e < CONV_TOL and err_g < CONV_TOL
    # the step-size maxima of A7 (pyx:523-524) over the finished back-projection
    red = job.red_keys()
    gg = (np.float32(10000.0) * g + (u - case["u0"]) * np.float32(0.5)).astype(np.float32)
    def key_to_float(k):   # ics_key2f (csrc/ics_common.h)
        k = int(k)
        return np.array([(k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)], np.uint32).view(np.float32)[0]
    for c in range(3):
        assert key_to_float(red[c]) == np.max(np.abs(gg[..., c]))
        assert key_to_float(red[3 + c]) == np.max(u[..., c])
    # the residual frame outside the image stays zero; u untouched
    assert np.array_equal(job.read(nv.BUF_U), u)
    job.close()


@pytest.mark.parametrize("M,N,MK,blind", [(130, 67, 9, True), (257, 300, 15, True), (100, 90, 31, False), (97, 133, 45, True)])
def test_planar_update_and_gradient_equal_the_hwc_passes(M, N, MK, blind, debug_switch):
    from lib import _native as nv
    debug_switch("fft_gradk", 0)      # the pipeline's matrix-core gradient on the mirrors (k_gradk_mfma<NB, true>), not the one on the tiles
    out = {}
    for conv in (2, FFT):
        job, case, psf = make_job(M, N, MK, seed=M + N, blind=blind)
        rng = np.random.default_rng(3)
        u = (case["u0"] + 0.02 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
        e = np.zeros_like(case["image"]); e[...] = (0.01 * rng.standard_normal(e.shape)).astype(np.float32)
        g = (1e-4 * rng.standard_normal(u.shape)).astype(np.float32)
        job.write(nv.BUF_U, u); job.write(nv.BUF_UT, case["u0"]); job.write(nv.BUF_ERROR, e); job.write(nv.BUF_GRADU, g)
        p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=blind, conv=conv)
        # the maxima the update reads: taken by the back-projection stage normally; here by the row-band reduction over all rows
        pb = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=blind, conv=0, band_rows=(0, u.shape[0]))
        job.stage(nv.STAGE_BAND_REDUCE, pb)
        job.stage(nv.STAGE_PSF_GRADIENT, p)
        gk = job.read(nv.BUF_GRADK)
        job.stage(nv.STAGE_UPDATE, p)
        out[conv] = (job.read(nv.BUF_U), gk, job.red_keys())
        job.close()
    assert np.array_equal(out[2][0], out[FFT][0], equal_nan=True)      # A5-A10: bit-identical
    assert np.array_equal(out[2][1], out[FFT][1])                      # A13: bit-identical
    assert np.array_equal(out[2][2], out[FFT][2])


def test_auto_picks_the_tiles_for_wide_psfs_on_big_frames():
    from lib import _native as nv

    def describe(M, N, MK, **kw):
        return nv.describe(M, N, MK, nv.RLJob.params(1, 200, 1, 200, 1e9, 1, 1e-3, 1e4, True, **kw))
    r = describe(2048, 2048, 31)
    assert r.conv_family == 5 and r.conv_fp16_split == 0 and r.gradk_family == 7 and r.gradk_fp16_split == 0      # A11 + A13 fused on the tiles (round 6)
    assert describe(2048, 2048, 31, flags=nv.FLAG_NO_FUSED_GRADK).gradk_family == 6
    r = describe(2048, 2048, 17)                     # round 6: blind 17 x 17 from 2 Mpx, 13 x 13 / 15 x 15 from 8 Mpx
    assert r.conv_family == 5 and describe(1024, 1024, 17).conv_family == 1
    r = describe(2048, 2048, 15)
    assert r.conv_family == 1 and r.gradk_family == 1
    r = describe(4096, 4096, 15)
    assert r.conv_family == 5 and r.gradk_family == 7
    r = describe(300, 300, 31)
    assert r.conv_family == 1
    r = describe(2048, 2048, 45, conv=FFT)
    assert r.conv_family == 5 and r.gradk_family == 7
    r = describe(2048, 2048, 31, tv_mode=2)          # the PAM kinds: convolutions and PSF gradient on the tiles, the rest on the HWC frames
    assert r.conv_family == 5 and r.gradk_family == 7
    r = describe(2048, 2048, 31, tv_mode=1)          # active MM-TV: matrix cores
    assert r.conv_family == 1


@pytest.mark.parametrize("blind", [False, True])
def test_whole_run_on_the_tiles_equals_the_matrix_core_run(blind):
    """a 1400 x 1100 frame, 31 x 31: AUTO takes the tiles; against the matrix-core run of the same call (both within 1e-5 of each other
    after two outer iterations: two correct evaluations of the same sums) and with the same stop-test scalars."""
    from lib import deconvolution as dc
    M, N, MK = 1100, 1400, 31
    case = orc.synth_case_large(M, N, MK, seed=5, blind=blind)
    res = {}
    for conv in (0, 2):
        u, psf, image = case["u0"].copy(), case["psf0"].copy(), case["image"].copy()
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            dc.richardson_lucy_MM(image, u, psf, *orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 2, 1e-3, 10000.0, blind=blind, conv=conv)
        st = dc.richardson_lucy_MM.last
        res[conv] = (u, psf, st)
        assert st.iterations_done == 2 and not st.has_nan
    eu, ep = rel_err(res[0][0], res[2][0]), rel_err(res[0][1], res[2][1])
    print("tiles vs matrix cores, blind=%s: u %.2e psf %.2e" % (blind, eu, ep))
    assert eu < 1e-5 and ep < 1e-5
    for a, b in ((res[0][2].trace_M_r, res[2][2].trace_M_r), (res[0][2].trace_Hu, res[2][2].trace_Hu), (res[0][2].trace_varu, res[2][2].trace_varu)):
        np.testing.assert_allclose(np.array(a[:2]), np.array(b[:2]), rtol=2e-3)
    dc._drop_jobs()


@pytest.mark.parametrize("blind", [False, True])
def test_overlapped_statistics_and_dropped_iteration_on_the_tiles(blind, debug_switch):
    """ics_rl_run queues outer iteration i + 1 before the stop decision of i is known and drops it when the test fires (undo): the mirrors
    rotate with the frames they belong to (a mirror is looked up by its frame buffer), the residual's mirror ping-pongs with e / e2, the
    spectra are rebuilt from the restored PSF.  The overlapped run must equal the drained one bit for bit, early stop included."""
    from lib import deconvolution as dc
    import contextlib, io
    M, N, MK = 300, 340, 17
    case = orc.synth_case(M, N, MK, seed=77, blind=blind)
    res = {}
    for ov in (0, 2):
        debug_switch("overlap", ov)
        dc._drop_jobs()
        u, psf, image = case["u0"].copy(), case["psf0"].copy(), case["image"].copy()
        with contextlib.redirect_stdout(io.StringIO()):
            dc.richardson_lucy_MM(image, u, psf, *orc.default_window(M, N, MK), 0.0, M, N, 3, MK, 12, 5e-3, 10000.0, blind=blind, conv=FFT)
        st = dc.richardson_lucy_MM.last
        res[ov] = (u, psf, st.iterations_done, st.stopped, list(st.trace_M_r[:st.trace_len]))
    print("blind=%s: %d outer iterations, stopped=%d" % (blind, res[0][2], res[0][3]))
    assert res[0][2] == res[2][2] and res[0][3] == res[2][3] and res[0][4] == res[2][4]
    assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1])
    dc._drop_jobs()


def test_few_persistent_workgroups_walk_many_units(debug_switch):
    """the unit walk with a grid that is no multiple of eight and far smaller than the unit count (test hook max_wgs)"""
    from lib import _native as nv
    debug_switch("max_wgs", 5)
    M, N, MK = 333, 410, 23
    job, case, psf = make_job(M, N, MK, seed=5)
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=False, conv=FFT)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    full = conv_valid64(case["u0"], psf)
    assert np.max(np.abs(e - (full - case["image"]))) / np.max(np.abs(full)) < CONV_TOL
    job.stage(nv.STAGE_BACKPROJECT, p)
    assert rel_err(job.read(nv.BUF_GRADU), corr_full64(e.astype(np.float64), psf)) < CONV_TOL
    job.close()


@pytest.mark.parametrize("M,N,MK", [(90, 100, 15), (150, 260, 17), (200, 120, 31), (300, 310, 23), (230, 333, 45), (190, 170, 63), (300, 100, 5), (176, 108, 17),
                                    (150, 170, 67), (120, 140, 97)])
def test_fft_psf_gradient_against_float64(M, N, MK):
    """A12 + A13 on the tiles (k_gradk_fft): the residual of the frame's own synthesis (e' = conv(u, psf) - image, as in the loop) against u,
    float64 direct sums, the gate of every other gradient kernel (1e-5 of max |gradk|; measured 1 - 3e-7)."""
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=MK + N, blind=True)
    rng = np.random.default_rng(17)
    u = (case["u0"] + 0.03 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=True, conv=FFT)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk = job.read(nv.BUF_GRADK)
    ref = gradk64(u.astype(np.float64), e.astype(np.float64))
    err = rel_err(gk, ref)
    print("%dx%d K=%d: PSF gradient on the tiles %.2e" % (M, N, MK, err))
    assert err < 1e-5
    job.close()


@pytest.mark.parametrize("M,N,MK", [(90, 100, 15), (114, 114, 15), (115, 229, 15), (150, 260, 17), (200, 120, 31), (300, 310, 23), (230, 333, 45), (190, 170, 63),
                                    (64, 129, 65), (300, 100, 5), (176, 108, 17), (40, 50, 3), (700, 900, 15), (150, 170, 67), (200, 130, 85)])
def test_fused_residual_and_gradient_unit(M, N, MK):
    """A11 + A12 + A13 (pyx:555-571) as ONE unit per tile pair on the tiles (k_synth_gradk_fft: transform of the window, product, inverse,
    residual in the tile buffer, its transform, product with the kept window spectrum -- three transforms instead of four).
    Against float64 direct sums at the gates of the two kernels it replaces, and BIT FOR BIT against those two kernels."""
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=MK + N, blind=True)
    rng = np.random.default_rng(17)
    u = (case["u0"] + 0.03 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=True, conv=FFT)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e2 = job.read(nv.BUF_ERROR)
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk2 = job.read(nv.BUF_GRADK)
    job.write(nv.BUF_ERROR, np.full_like(e2, 7.0))           # the fused stage must rewrite every pixel of the residual
    job.write(nv.BUF_GRADK, np.zeros_like(gk2))
    job.stage(nv.STAGE_SYNTH_GRADK, p)
    e, gk = job.read(nv.BUF_ERROR), job.read(nv.BUF_GRADK)
    full = conv_valid64(u, psf)
    err_e = np.max(np.abs(e - (full - case["image"]))) / np.max(np.abs(full))
    err_g = rel_err(gk, gradk64(u.astype(np.float64), e.astype(np.float64)))
    print("%dx%d K=%d: fused unit residual %.2e gradient %.2e" % (M, N, MK, err_e, err_g))
    assert err_e < CONV_TOL and err_g < 1e-5
    assert np.array_equal(e, e2) and np.array_equal(gk, gk2)
    assert np.array_equal(job.read(nv.BUF_U), u)
    job.close()


@pytest.mark.parametrize("MK,tv_mode", [(15, 0), (31, 0), (23, 2)])
def test_whole_blind_run_with_the_fused_unit_equals_the_two_kernel_run(MK, tv_mode, debug_switch):
    """ics_rl_run on the tiles with A11 + A13 fused (the residual stored under the stop-test window only) against the same run with the two
    kernels: u, PSF, every stop-test scalar bit for bit; with few persistent workgroups (many pairs per workgroup) as well."""
    from lib import deconvolution as dc
    import contextlib, io
    M, N = 500, 620
    case = orc.synth_case(M, N, MK, seed=MK, blind=True)
    for wgs in (0, 7):
        res = {}
        for fused in (1, 0):
            debug_switch("fft_fused", fused)
            debug_switch("max_wgs", wgs)
            dc._drop_jobs()
            u, psf, image = case["u0"].copy(), case["psf0"].copy(), case["image"].copy()
            with contextlib.redirect_stdout(io.StringIO()):
                dc.richardson_lucy_MM(image, u, psf, *orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 3, 1e-3, 10000.0, blind=True, conv=FFT, tv_mode=tv_mode)
            st = dc.richardson_lucy_MM.last
            assert st.iterations_done == 3 and not st.has_nan
            res[fused] = (u, psf, list(st.trace_M_r[:3]), list(st.trace_Hu[:3]), list(st.trace_varu[:3]))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert res[0][2:] == res[1][2:]
    dc._drop_jobs()


@pytest.mark.parametrize("M,N,MK", [(500, 640, 15), (420, 700, 9), (330, 350, 21), (300, 310, 23), (40, 50, 3), (700, 300, 5), (300, 100, 5), (620, 410, 13), (450, 450, 17)])
def test_synthesis_and_back_projection_in_one_unit(M, N, MK):
    """Mode 2 of the tiles (k_conv_fft<2>, ICS_STAGE_SYNTH_BACKPROJECT): A1 + A2 + A3 (pyx:477-491) of a tile pair in one unit -- tiles whose
    residual window lies inside the image stay in the frequency domain between the two convolutions (G = S1 (S0 T - F), the image as the
    precomputed spectra of its windows), the tiles of the outer ring mask the residual in between.  Against float64 sums formed from u and
    the image alone: the back-projection of a SMALL residual e = conv(u) - image carries conv's absolute rounding (~1e-7 |u|) on every
    path, so the error is quoted against max |gradu| and held to three times what the two kernels show on the same frame (+ 1e-6), and in
    absolute terms to the convolutions' own stage gate (5e-6 of max |conv(u)|); the
    step-size maxima are exactly those of the stage's own output."""
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=MK + M)
    rng = np.random.default_rng(7)
    u = (case["u0"] + 0.01 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=False, conv=FFT)
    e64 = conv_valid64(u, psf) - case["image"].astype(np.float64)
    g_ref = corr_full64(e64, psf)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    job.stage(nv.STAGE_BACKPROJECT, p)
    err2 = rel_err(job.read(nv.BUF_GRADU), g_ref)
    job.write(nv.BUF_GRADU, np.full_like(u, 3.0))
    job.stage(nv.STAGE_SYNTH_BACKPROJECT, p)
    g = job.read(nv.BUF_GRADU)
    err = rel_err(g, g_ref)
    print("%dx%d K=%d: one unit %.2e, two kernels %.2e (of max |gradu| = %.2e)" % (M, N, MK, err, err2, np.max(np.abs(g_ref))))
    assert err < 3 * err2 + 1e-6 and np.max(np.abs(g - g_ref)) < CONV_TOL * np.max(np.abs(conv_valid64(u, psf)))
    red = job.red_keys()
    gg = (np.float32(10000.0) * g + (u - case["u0"]) * np.float32(0.5)).astype(np.float32)

    def key_to_float(k):
        k = int(k)
        return np.array([(k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)], np.uint32).view(np.float32)[0]
    for c in range(3):
        assert key_to_float(red[c]) == np.max(np.abs(gg[..., c]))
        assert key_to_float(red[3 + c]) == np.max(u[..., c])
    assert np.array_equal(job.read(nv.BUF_U), u)
    job.close()


@pytest.mark.parametrize("MK,blind,tv_mode", [(9, False, 0), (9, True, 0), (15, False, 0), (15, True, 0), (15, True, 2), (9, False, 3), (21, True, 3)])
def test_whole_run_with_one_unit_per_tile_pair_equals_the_two_kernel_run(MK, blind, tv_mode, debug_switch):
    """ics_rl_run on the tiles with A1 + A3 as one unit (fft_conv2, default for small PSFs) against the same run with the two kernels:
    two correct evaluations of the same sums (the PAM kinds, tv_mode 2 / 3, included: their epilogue rides on the same units) -- u, PSF within 1e-5
    after three outer iterations, the stop-test scalars (whose residual
    window comes from a window-sized launch of mode 0, non-blind, or from the fused A11 + A13 unit, blind) within 2e-3."""
    from lib import deconvolution as dc
    import contextlib, io
    M, N = 700, 820
    case = orc.synth_case_large(M, N, MK, seed=MK, blind=blind)
    res = {}
    for sw in (1, 0):
        debug_switch("fft_conv2", sw)
        dc._drop_jobs()
        u, psf, image = case["u0"].copy(), case["psf0"].copy(), case["image"].copy()
        with contextlib.redirect_stdout(io.StringIO()):
            dc.richardson_lucy_MM(image, u, psf, *orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 3, 1e-3, 10000.0, blind=blind, conv=FFT, tv_mode=tv_mode)
        st = dc.richardson_lucy_MM.last
        assert st.iterations_done == 3 and not st.has_nan
        res[sw] = (u, psf, np.array(st.trace_M_r[:3]), np.array(st.trace_Hu[:3]), np.array(st.trace_varu[:3]))
    eu, ep = rel_err(res[1][0], res[0][0]), rel_err(res[1][1], res[0][1])
    print("K=%d blind=%s tv_mode=%d: one unit vs two kernels u %.2e psf %.2e" % (MK, blind, tv_mode, eu, ep))
    assert eu < 1e-5 and ep < 1e-5
    for k in (2, 3, 4):
        np.testing.assert_allclose(res[1][k], res[0][k], rtol=2e-3)
    dc._drop_jobs()


def test_few_persistent_workgroups_walk_many_units_of_one_unit_mode(debug_switch):
    """mode 2 with a grid far smaller than its unit count and no multiple of eight (test hook max_wgs): interior and outer-ring units in turn
    on the same workgroup, the next unit's window prefetched across both kinds"""
    from lib import _native as nv
    debug_switch("max_wgs", 5)
    M, N, MK = 520, 610, 15
    job, case, psf = make_job(M, N, MK, seed=9)
    rng = np.random.default_rng(5)
    u = (case["u0"] + 0.01 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u); job.write(nv.BUF_UT, case["u0"])
    p = job.params(1, 5, 1, 5, 1e9, 1, 1e-3, 10000.0, blind=True, conv=FFT)
    job.stage(nv.STAGE_SYNTH_BACKPROJECT, p)
    g5, red5 = job.read(nv.BUF_GRADU), job.red_keys()[:6].copy()
    debug_switch("max_wgs", 0)
    job.write(nv.BUF_GRADU, np.zeros_like(u))
    job.stage(nv.STAGE_SYNTH_BACKPROJECT, p)
    assert np.array_equal(job.read(nv.BUF_GRADU), g5) and np.array_equal(job.red_keys()[:6], red5)      # the walk does not change a bit
    g_ref = corr_full64(conv_valid64(u, psf) - case["image"].astype(np.float64), psf)
    assert np.max(np.abs(g5 - g_ref)) < CONV_TOL * np.max(np.abs(conv_valid64(u, psf)))
    job.close()


@pytest.mark.parametrize("MK,blind", [(71, True), (71, False), (85, True), (97, True), (111, True), (141, False), (255, True)])
def test_whole_run_of_a_wide_psf_on_the_tiles_against_the_oracle(MK, blind):
    """PSF sizes 67 ... 85 (late round 6): ICS_CONV_AUTO takes the tiles from 0.5 Mpx -- the stage functions never depended on the size, a
    tile's valid part is 128 - K + 1 pixels a side -- where the matrix cores ran tap blocks; 87 ... 255: tap blocks ON the tiles (2 x 2 to
    4 x 4 blocks of at most 65 x 65 taps, their products summed in the frequency domain).  A whole call against the pinned oracle
    (scipy's complex64 FFT, the reference's own method), gate 1e-4 of the reference's maximum, same printed lines."""
    from lib import deconvolution as dc, _native as nv
    import contextlib, io
    M, N = 760, 800
    case = orc.synth_case_large(M, N, MK, seed=MK, blind=blind)
    r = nv.describe(M, N, MK, nv.RLJob.params(1, 200, 1, 200, 1e9, 1, 1e-3, 1e4, blind))
    assert r.conv_family == 5 and r.gradk_family == (0 if not blind else (7 if MK <= 85 else 6))
    win = (40, 441, 60, 461)      # (the 255-px default window of the drivers is narrower than the widest PSFs here)
    u_r, psf_r = case["u0"].copy(), case["psf0"].copy()
    buf_r = io.StringIO()
    with contextlib.redirect_stdout(buf_r), np.errstate(all="ignore"):
        orc.richardson_lucy_MM(case["image"].copy(), u_r, psf_r, *win, 1e9, M, N, 3, MK, 2, 1e-3, 10000.0, blind=blind)
    u, psf = case["u0"].copy(), case["psf0"].copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *win, 1e9, M, N, 3, MK, 2, 1e-3, 10000.0, blind=blind)
    st = dc.richardson_lucy_MM.last
    eu, ep = rel_err(u, u_r), rel_err(psf, psf_r)
    print("K=%d blind=%s on the tiles against the oracle: u %.2e psf %.2e" % (MK, blind, eu, ep))
    assert st.iterations_done == 2 and not st.has_nan and eu < 1e-4 and ep < 1e-4
    assert len(buf.getvalue().splitlines()) == len(buf_r.getvalue().splitlines())
    dc._drop_jobs()


@pytest.mark.parametrize("blind", [False, True])
def test_nan_in_the_image_on_the_tiles_is_reported_not_raised(blind):
    """pyx:671-672: NaN is printed, never raised.  The reference's own frame-wide FFT convolution turns one NaN pixel into an all-NaN
    frame; on the tiles it fills the tiles it touches and the step sizes (maxima as integer keys: a NaN is the largest key) carry it to
    every pixel with the first update -- either way the call returns, reports it and prints the reference's line."""
    import contextlib, io
    from lib import deconvolution as dc
    M, N, MK = 260, 300, 21
    case = orc.synth_case(M, N, MK, seed=2, blind=blind)
    case["image"][100, 120, 1] = np.nan
    u, psf = case["u0"].copy(), case["psf0"].copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 1, 1e-3, 1e4, blind=blind, conv=FFT)
    st = dc.richardson_lucy_MM.last
    assert st.has_nan and "has NaN after DoF correction" in buf.getvalue()
    u_r, psf_r = case["u0"].copy(), case["psf0"].copy()
    with np.errstate(all="ignore"):
        orc.richardson_lucy_MM(case["image"].copy(), u_r, psf_r, *orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 1, 1e-3, 1e4, blind=blind, quiet=True)
    assert np.array_equal(np.isnan(u), np.isnan(u_r)) and np.isnan(u[:, :, 1]).all()      # the NaN's channel, whole frame: as far as in the reference
    if blind:
        assert np.array_equal(np.isnan(psf), np.isnan(psf_r))
    dc._drop_jobs()
