"""Active MM-TV mode (tv_mode = 1): BUILD-DEFINED extension, PARITY UNPINNED (the reference holds this
arithmetic only as unreachable code).  The HIP path is validated against the build's own
CPU restatement (oracle/rl_ext_oracle.py, <= 1e-5) and through properties."""
import contextlib
import io

import numpy as np
import pytest

import rl_ext_oracle as ext
import rl_mm_oracle as orc
from helpers import rel_err


def test_tv_term_properties_cpu():
    rng = np.random.default_rng(0)
    flat = np.full((12, 14, 3), 0.3, np.float32)
    T, act = ext.tv_term(flat, flat, 1e-2)
    assert np.all(T == 0) and act[1:-1, 1:-1].all() and not act[0].any() and not act[:, -1].any()
    u = rng.random((15, 17, 3), dtype=np.float32)
    ut = rng.random((15, 17, 3), dtype=np.float32)
    T, act = ext.tv_term(u, ut, 1e-6)
    assert np.all(T[0] == 0) and np.all(T[:, 0] == 0) and np.isfinite(T).all() and np.abs(T[1:-1, 1:-1]).max() > 0
    # T is -grad of a TV-like energy: a single bright pixel is pushed down (T > 0 there means u decreases)
    spike = np.full((9, 9, 3), 0.5, np.float32); spike[4, 4] = 0.9
    T, _ = ext.tv_term(spike, spike, 1e-2)
    assert np.all(T[4, 4] > 0)


def test_ext_oracle_differs_from_shipped_and_modifies_image():
    # (lambd = 100: with lambd = 1e4 the image step dt*T/lambd of pyx:549 is below float32 resolution)
    case = orc.synth_case(40, 36, 5, seed=4)
    args = (*orc.default_window(40, 36, 5), 1e9, 40, 36, 3, 5, 2, 1e-3, 100.0)
    img1, u1 = case["image"].copy(), case["u0"].copy()
    orc.richardson_lucy_MM(img1, u1, case["psf0"].copy(), *args, blind=False, quiet=True, conv="direct")
    img2, u2 = case["image"].copy(), case["u0"].copy()
    ext.richardson_lucy_MM_tv(img2, u2, case["psf0"].copy(), *args, blind=False)
    assert np.array_equal(img1, case["image"]) and not np.array_equal(img2, case["image"])
    assert rel_err(u2, u1) > 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,MK,blind", [(70, 131, 9, False), (64, 64, 15, True), (33, 37, 3, False)])
def test_gpu_tv_term_stage_matches_oracle(M, N, MK, blind):
    from lib import _native as nv
    case = orc.synth_case(M, N, MK, seed=M + 1, blind=blind)
    rng = np.random.default_rng(5)
    u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job = nv.RLJob(M, N, MK)
    job.upload(case["image"], u, case["psf0"])
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=blind, tv_mode=1)
    job.stage(nv.STAGE_TVTERM, p)
    T = job.read(nv.BUF_TV)
    T_ref, act = ext.tv_term(u, case["u0"], 1e-2 if blind else 1e-6)
    scale = np.abs(T_ref).max()
    assert np.max(np.abs(T - T_ref)) / scale < 1e-5          # sqrtf vs powf(x, 0.5): <= 1 ulp apart
    assert np.all(T[~act] == 0)
    job.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,MK,blind,outer,lambd", [(65, 49, 9, False, 2, 1e4), (97, 81, 7, True, 2, 1e4), (129, 129, 15, False, 2, 200.0),
                                                       (80, 70, 5, True, 3, 50.0)])
def test_gpu_tv_mode_run_matches_ext_oracle(M, N, MK, blind, outer, lambd):
    from lib import deconvolution as dc
    case = orc.synth_case(M, N, MK, seed=M + MK, blind=blind)
    win = orc.default_window(M, N, MK)
    args = (*win, 1e9, M, N, 3, MK, outer, 1e-3, lambd)
    img_r, u_r, psf_r = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    ext.richardson_lucy_MM_tv(img_r, u_r, psf_r, *args, blind=blind)
    img, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        out = dc.richardson_lucy_MM(img, u, psf, *args, blind=blind, tv_mode=1)
    assert np.shares_memory(out, u)
    eu, ei, ep = rel_err(u, u_r), rel_err(img, img_r), rel_err(psf, psf_r)
    print("tv_mode=1 %dx%d k%d blind=%d: rel err u=%.2e image=%.2e psf=%.2e" % (M, N, MK, blind, eu, ei, ep))
    assert eu < 1e-5 and ei < 1e-5 and ep < 1e-5
    if lambd < 1e3 or blind:
        assert not np.array_equal(img, case["image"])                   # pyx:549 is live in this mode
    if blind:
        assert np.all(psf >= 0) and np.allclose(psf.sum(axis=(0, 1)), 1, atol=1e-5)   # PSF stays on the simplex
    # and it is a different algorithm from the shipped one
    u_s = case["u0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u_s, case["psf0"].copy(), *args, blind=blind)
    assert rel_err(u, u_s) > 1e-6


# ---- tv_mode 2 / 3: PAM with isotropic / collaborative L-inf,1,1 TV (build-defined, parity unpinned) ---------
def tv_energy(u, eps, collaborative):
    u = u.astype(np.float64)
    dx = np.zeros_like(u); dx[:-1] = u[1:] - u[:-1]
    dy = np.zeros_like(u); dy[:, :-1] = u[:, 1:] - u[:, :-1]
    if collaborative:
        return float(np.sum(np.sqrt(np.max(np.abs(dx), axis=2) ** 2 + eps ** 2) + np.sqrt(np.max(np.abs(dy), axis=2) ** 2 + eps ** 2)))
    return float(np.sum(np.sqrt(dx ** 2 + dy ** 2 + eps ** 2)))


@pytest.mark.parametrize("collaborative", [False, True])
def test_pam_tv_term_is_the_gradient_of_the_tv_energy(collaborative):
    """finite differences on a 9x11 frame ('gradient check by finite differences on 9x9')"""
    u = np.random.default_rng(3).random((9, 11, 3), dtype=np.float32)
    T = ext.pam_tv_term(u, 1e-2, collaborative).astype(np.float64)
    for (i, j, c) in [(4, 5, 0), (3, 3, 2), (6, 8, 1), (2, 7, 0), (1, 1, 1), (7, 9, 2)]:
        h = 1e-5
        up, um = u.astype(np.float64).copy(), u.astype(np.float64).copy()
        up[i, j, c] += h; um[i, j, c] -= h
        fd = (tv_energy(up, 1e-2, collaborative) - tv_energy(um, 1e-2, collaborative)) / (2 * h)
        assert abs(fd - T[i, j, c]) < 1e-5
    flat = np.full((8, 8, 3), 0.4, np.float32)
    assert np.all(ext.pam_tv_term(flat, 1e-2, collaborative) == 0)      # TV gradient of a constant image vanishes


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 3])
def test_gpu_pam_tv_term_stage_matches_oracle(kind):
    from lib import _native as nv
    M, N, MK = 70, 131, 9
    case = orc.synth_case(M, N, MK, seed=8, blind=True)
    rng = np.random.default_rng(6)
    u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job = nv.RLJob(M, N, MK)
    job.upload(case["image"], u, case["psf0"])
    p = job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 10000.0, blind=True, tv_mode=kind)
    job.stage(nv.STAGE_TVTERM, p)
    T = job.read(nv.BUF_TV)
    T_ref = ext.pam_tv_term(u, 1e-2, kind == 3)
    assert np.max(np.abs(T - T_ref)) < 2e-6 * np.abs(T_ref).max()
    job.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,MK,blind,kind", [(65, 49, 9, False, 2), (97, 81, 7, True, 2), (80, 70, 5, True, 3), (129, 129, 15, True, 3),
                                               (150, 140, 31, True, 3), (120, 131, 31, True, 2)])   # MK = 31: BASELINE.json configs[3]
def test_gpu_pam_run_matches_ext_oracle(M, N, MK, blind, kind):
    # (collaborative TV with the non-blind epsilon = 1e-6 is not trajectory-comparable: the arg-max channel of an
    #  almost flat pixel flips on 1e-7 differences; that mode is covered by the teacher-forced stage test above)
    from lib import deconvolution as dc
    case = orc.synth_case(M, N, MK, seed=M + MK + kind, blind=blind)
    args = (*orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 2, 1e-3, 50.0)
    img_r, u_r, psf_r = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    margins = [] if kind == 3 else None
    ext.richardson_lucy_PAM(img_r, u_r, psf_r, *args, blind=blind, collaborative=(kind == 3), margins=margins)
    img, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(img, u, psf, *args, blind=blind, tv_mode=kind)
    eu, ep = rel_err(u, u_r), rel_err(psf, psf_r)
    print("tv_mode=%d %dx%d k%d blind=%d: rel err u=%.2e psf=%.2e" % (kind, M, N, MK, blind, eu, ep))
    assert ep < 1e-5
    if kind == 3:
        # the collaborative term takes the arg-max channel per pixel: a 1e-7 difference in the convolution (fp32
        # chain in the oracle, fp16-split MFMA or packed fp32 on the device) can flip it at an isolated, almost
        # flat pixel, which then differs by ~1e-4.  Gate the bulk at 1e-5 and the outliers by count and size.
        d = np.abs(u - u_r) / np.abs(u_r).max()
        print("   collaborative TV: fraction of pixels beyond 1e-5: %.2e, max %.2e" % (np.mean(d > 1e-5), d.max()))
        assert np.mean(d > 1e-5) < 2e-3 and d.max() < 5e-3      # (measured 1.1e-3 at 150x140 / 31x31, 0 ... 4e-4 at the small PSFs)
        # ... and that explanation is CHECKED, not assumed: every outlier pixel must sit at (or next to: the term is a divergence, and
        # a flipped pixel then tips its neighbours) a pixel whose arg-max margin in the oracle's own trajectory came within the size
        # of the deviations themselves.  Primary flips need a margin below the convolutions' rounding differences (~1e-6 of the data
        # range); once two trajectories differ by 1e-4 at a pixel, margins up to that size flip next to it.
        out = d.max(axis=2) > 1e-5
        if out.any():
            from scipy.ndimage import binary_dilation
            m = np.min(np.stack(margins), axis=0) / np.abs(u_r).max()
            primary = binary_dilation(m < 1e-6, iterations=1)
            chain = binary_dilation(m < 2.0 * d.max(), iterations=2)
            n_out, n_pri, n_chain = int(out.sum()), int((out & primary).sum()), int((out & chain).sum())
            print("   outliers %d: next to a margin < 1e-6: %d, next to a margin < 2 x max deviation: %d; tie pixels (margin < 1e-6): %d of %d"
                  % (n_out, n_pri, n_chain, int((m < 1e-6).sum()), m.size))
            assert n_chain == n_out, "an outlier that no near-tie explains: not an arg-max flip"
            assert n_pri >= 1
    else:
        assert eu < 1e-5
    assert np.array_equal(img, case["image"])                           # PAM leaves the blurry image alone
    if blind:
        assert np.all(psf >= 0) and np.allclose(psf.sum(axis=(0, 1)), 1, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,MK,blind,kind", [(150, 140, 31, True, 2), (200, 180, 21, False, 2), (150, 140, 31, True, 3), (260, 300, 45, True, 2)])
def test_gpu_pam_with_the_convolutions_on_the_transform_tiles(M, N, MK, blind, kind):
    """The PAM kinds with conv = ICS_CONV_FFT (AUTO takes this route for wide PSFs on large frames): the whole inner iteration on the
    channel-planar mirrors -- TV term (k_tvterm_pam<.., PL>), convolutions and PSF gradient on the fp32 transform tiles, the back-projection's
    epilogue forming G = T + lambd gradu with its maxima (k_conv_fft<1, true>), the update (k_update_planar).  Same arithmetic around other
    convolution kernels: the run must agree with the fp32-product HWC path."""
    from lib import deconvolution as dc
    case = orc.synth_case(M, N, MK, seed=M + MK + kind, blind=blind)
    args = (*orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 2, 1e-3, 50.0)
    res = {}
    for conv in (1, 3):
        img, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            dc.richardson_lucy_MM(img, u, psf, *args, blind=blind, tv_mode=kind, conv=conv)
        st = dc.richardson_lucy_MM.last
        assert np.array_equal(img, case["image"]) and not st.has_nan and st.iterations_done == 2
        res[conv] = (u, psf, st.M_r, st.Hu, buf.getvalue())
    d = np.abs(res[3][0] - res[1][0]) / np.abs(res[1][0]).max()
    ep = rel_err(res[3][1], res[1][1])
    print("tv_mode=%d %dx%d k%d blind=%d tiles vs HWC fp32: u max %.2e (beyond 1e-5: %.2e), psf %.2e" % (kind, M, N, MK, blind, d.max(), np.mean(d > 1e-5), ep))
    assert ep < 1e-5
    if kind == 3:
        assert np.mean(d > 1e-5) < 2e-3 and d.max() < 5e-3          # (arg-max flips of the collaborative term, as in the oracle comparison above)
    elif blind:
        assert d.max() < 1e-5
    else:   # non-blind epsilon = 1e-6: the TV term of nearly flat pixels amplifies 1e-7 differences of the convolutions (see the test below)
        assert d.max() < 1e-4 and np.mean(d > 1e-5) < 1e-3
    assert abs(res[3][2] - res[1][2]) <= 2e-3 * abs(res[1][2]) and abs(res[3][3] - res[1][3]) <= 1e-4 * abs(res[1][3])
    assert len(res[3][4].splitlines()) == len(res[1][4].splitlines())
    from lib import _native as nv
    with pytest.raises(nv.NativeError):                                 # the active MM-TV kind stays off the tiles
        dc.richardson_lucy_MM(case["image"].copy(), case["u0"].copy(), case["psf0"].copy(), *args, blind=blind, tv_mode=1, conv=3)
    job = nv.RLJob(M, N, MK)                                            # ... and so do single stages of the TV variants (ICS_ENOSUP, not a silent other path)
    job.upload(case["image"], case["u0"], case["psf0"])
    with pytest.raises(nv.NativeError) as ei:
        job.stage(nv.STAGE_BACKPROJECT, job.params(*orc.default_window(M, N, MK), 1e9, 1, 1e-3, 50.0, blind, tv_mode=kind, conv=3))
    assert ei.value.code == nv.ICS_ENOSUP
    job.close()


@pytest.mark.gpu
@pytest.mark.parametrize("MK", [45, 63])
def test_nonblind_pam_at_large_psf_deviation_is_the_tv_term_of_nearly_flat_pixels(MK):
    """Round-3 fuzz: non-blind PAM (tv_mode 2, epsilon = 1e-6) at PSF sizes 45 ... 63 deviated by 1e-5 ... 1e-3 from the extended oracle.
    Which term is it, and is it the device's?  (1) / (2): the oracle's OWN two forms -- scipy's FFT convolutions and float64 direct sums,
    which differ by ~1e-7 per convolution -- deviate from each other by as much as the device deviates from either: three evaluations
    of the same arithmetic, three trajectories ~1e-3 apart after ten inner iterations.  (3) Stage by stage on the state after one inner
    iteration: the two forms' u differ by ~1e-7 (convolution rounding), their TV terms by O(1) -- and only next to pixels where
    |grad u| is within a few hundred epsilon, i.e. where the normalised gradient grad u / sqrt(|grad u|^2 + eps^2) of a nearly flat
    pixel turns on that rounding.  So the term that flips is the TV term of nearly flat pixels, what flips it is ANY rounding difference
    between two correct convolutions, and the deviation is a property of the mode at epsilon = 1e-6 (wide PSFs flatten u), not of an
    implementation.  The stage tests above gate the device's TV term itself teacher-forced."""
    from lib import deconvolution as dc
    M, N = 150, 140
    case = orc.synth_case(M, N, MK, seed=MK, blind=False)
    args = (*orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 2, 1e-3, 50.0)

    def run_oracle(conv, iters=2):
        img, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
        a = list(args); a[9] = iters
        ext.richardson_lucy_PAM(img, u, psf, *a, blind=False, collaborative=False, conv=conv)
        return u
    u_dir, u_fft = run_oracle("direct"), run_oracle("scipy")
    img, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(img, u, psf, *args, blind=False, tv_mode=2)
    e_dir, e_fft, o_fft = rel_err(u, u_dir), rel_err(u, u_fft), rel_err(u_fft, u_dir)
    print("non-blind PAM %dx%d k%d: device vs direct oracle %.2e | device vs FFT oracle %.2e | FFT oracle vs direct oracle %.2e" % (M, N, MK, e_dir, e_fft, o_fft))
    assert e_dir < 3.0 * o_fft + 1e-6 and e_fft < 3.0 * o_fft + 1e-6      # (1), (2): no further from either form than they are from each other
    assert o_fft > 1e-5                                                   # ... and that distance is what the fuzz saw, device or not
    # (3) one inner iteration (iterations = 1 runs five: take the state of a one-iteration chain instead -- the u-step of the first
    # inner iteration is reproduced here from the oracle's own pieces)
    eps = 1e-6
    img0, u0 = case["image"], case["u0"]
    rot = ext.rotate_180(case["psf0"])
    terms = {}
    for name, cv in (("direct", ext.base._conv_direct), ("scipy", ext.base._conv_scipy)):
        synth = np.stack([cv(u0[..., c], case["psf0"][..., c], "valid") for c in range(3)], axis=-1).astype(np.float32)
        gradu = np.stack([cv((synth - img0)[..., c], rot[..., c], "full") for c in range(3)], axis=-1).astype(np.float32)
        T = ext.pam_tv_term(u0, eps, False)
        g = (T.astype(np.float64) + (np.float32(50.0) * gradu).astype(np.float64)).astype(np.float32)
        un = u0.copy()
        for k in range(3):
            un[..., k] -= (np.float32(1e-3) * np.amax(u0[..., k]) / (np.amax(np.abs(g[..., k])) + np.float32(1e-15))) * g[..., k]
        terms[name] = (un, ext.pam_tv_term(un, eps, False))
    du = np.abs(terms["direct"][0].astype(np.float64) - terms["scipy"][0]).max() / np.abs(u0).max()
    dT = np.abs(terms["direct"][1].astype(np.float64) - terms["scipy"][1])
    # gradient magnitude of the direct form's u at every pixel (forward differences, as pam_tv_term takes them)
    ud = terms["direct"][0].astype(np.float64)
    gm = np.zeros(ud.shape)
    gm[:-1, :-1] = np.hypot(ud[1:, :-1] - ud[:-1, :-1], ud[:-1, 1:] - ud[:-1, :-1])
    flipped = dT > 0.05
    print("   after one inner iteration: |u_fft - u_direct| <= %.1e of the range; TV terms differ by > 0.05 at %d values, largest %.2f" % (du, int(flipped.sum()), dT.max()))
    assert du < 5e-6
    if flipped.any():
        from scipy.ndimage import maximum_filter
        near_flat = maximum_filter((gm < 1e3 * eps).any(axis=2), size=3)        # the term is a divergence: a flat pixel tips its neighbours
        assert np.all(near_flat[flipped.any(axis=2)]), "a TV term that differs without a nearly flat pixel next to it"


@pytest.mark.gpu
@pytest.mark.parametrize("M,MK,blind,kind", [(2048, 15, False, 2), (6144, 31, True, 3)], ids=["configs1-tv2-2048-k15", "configs3-tv3-6144-k31"])
def test_tv_variants_named_by_baseline_configs_at_full_size(M, MK, blind, kind):
    """BASELINE.json configs[1] names "non-blind RL + isotropic TV, 2048^2, 15x15", configs[3] "blind RL-TV with collaborative L-inf,1,1 RGB TV,
    6144^2, 31x31": the build-defined tv_mode 2 / 3 (parity unpinned: the reference has no such code) at THOSE sizes, not only on the
    150-px frames above.  One inner iteration stage by stage on the full frame, teacher-forced -- each stage's inputs are what the device
    holds -- and checked on crops (centre, a corner shared by four 64 x 64 tiles, frame corner, frame origin) against oracle/rl_ext_oracle.py
    and float64 direct sums: the TV term T = -div(p) (the gradient of the TV energy: test_pam_tv_term_is_the_gradient_of_the_tv_energy),
    G = T + lambd * back-projection, the step-size maxima over the WHOLE frame, the update bit for bit; for the collaborative kind every
    deviating pixel must sit next to an arg-max near-tie (rl_ext_oracle.argmax_margin).  Then one outer iteration of the loop: the
    image untouched, the PSF on the simplex, no NaN."""
    from lib import _native as nv
    from lib import deconvolution as dc
    dc._drop_jobs()
    N = M
    case = orc.synth_case_large(M, N, MK, seed=M + kind, blind=blind)
    rng = np.random.default_rng(11)
    u = case["u0"] + np.float32(0.02) * rng.standard_normal(case["u0"].shape, dtype=np.float32)
    pad, lambd, step = MK // 2, np.float32(10000.0), np.float32(1e-3)
    eps = 1e-2 if blind else 1e-6
    job = nv.RLJob(M, N, MK)
    job.upload(case["image"], u, case["psf0"])
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(*orc.default_window(M, N, MK), 1e9, 1, float(step), float(lambd), blind=blind, tv_mode=kind)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    job.stage(nv.STAGE_TVTERM, p)
    T = job.read(nv.BUF_TV)
    job.stage(nv.STAGE_BACKPROJECT, p)
    G = job.read(nv.BUF_GRADU)                              # PAM kinds: the back-projection stores G = T + lambd * gradu
    red = job.red_keys().copy()
    job.stage(nv.STAGE_UPDATE, p)
    u1 = job.read(nv.BUF_U)
    job.close()
    uM, uN = u.shape[:2]
    C = 96
    sy, sx = (uM // 2 // 64) * 64 + pad, (uN // 3 // 64) * 64 + pad
    crops = {"centre": (uM // 2 - C // 2, uN // 2 - C // 2), "seam": (sy - C // 2, sx - C // 2), "corner": (uM - C - 1, uN - C - 1), "origin": (1, 1)}
    psf = case["psf0"]
    rot = psf[::-1, ::-1]
    urange = float(np.abs(u).max())
    for name, (y0, x0) in crops.items():
        y1, x1 = y0 + C, x0 + C
        # TV term: a 1-px halo of u decides it
        ya, xa, yb, xb = max(y0 - 1, 0), max(x0 - 1, 0), min(y1 + 1, uM), min(x1 + 1, uN)
        T_ref = ext.pam_tv_term(u[ya:yb, xa:xb], eps, kind == 3)[y0 - ya:y0 - ya + C, x0 - xa:x0 - xa + C]
        if ya == 0 or xa == 0 or yb == uM or xb == uN:       # (the frame's 1-px border holds T = 0: compare away from the sub-array's own border)
            inner = (slice(1, C - 1), slice(1, C - 1))
        else:
            inner = (slice(0, C), slice(0, C))
        dT = np.abs(T[y0:y1, x0:x1][inner] - T_ref[inner]) / np.abs(T_ref).max()
        if kind == 2:
            assert dT.max() < 2e-6, (name, dT.max())
        else:
            bad = dT.max(axis=2) > 2e-6
            if bad.any():                                    # an arg-max flip: only next to a near-tie of the oracle's own margins
                from scipy.ndimage import binary_dilation
                m = ext.argmax_margin(u[ya:yb, xa:xb])[y0 - ya:y0 - ya + C, x0 - xa:x0 - xa + C][inner] / urange
                assert not (bad & ~binary_dilation(m < 1e-6, iterations=1)).any(), (name, int(bad.sum()))
            print("   %s: TV-term pixels beyond 2e-6: %d of %d" % (name, int(bad.sum()), bad.size))
        # G = float32(T + lambd * corr_full(e, psf)): float64 direct sums over the residual rows / columns the crop depends on
        ey0, ex0 = y0 - 2 * pad, x0 - 2 * pad                # image coordinates of the first residual row / column that reaches the crop
        ea, eb, ec, ed = max(ey0, 0), min(y1 - pad + pad, M), max(ex0, 0), min(x1 - pad + pad, N)
        esub = np.zeros((y1 - ey0, x1 - ex0, 3), np.float64)  # rows ey0 .. y1 - 1 (image coordinates), zero outside the image
        ra, rb, rc, rd = max(ey0, 0), min(y1, M), max(ex0, 0), min(x1, N)
        if rb > ra and rd > rc:
            esub[ra - ey0:rb - ey0, rc - ex0:rd - ex0] = e[ra:rb, rc:rd]
        full = np.stack([orc._conv_direct(esub[..., c], rot[..., c].astype(np.float64), "full") for c in range(3)], axis=-1)
        # full[s, t] = gradu at u-frame (ey0 + s, ex0 + t); rows s >= MK - 1 depend on esub only
        g_ref = full[y0 - ey0:y1 - ey0, x0 - ex0:x1 - ex0]
        G_ref = (T[y0:y1, x0:x1].astype(np.float64) + (lambd * g_ref.astype(np.float32)).astype(np.float64)).astype(np.float32)
        dG = np.abs(G[y0:y1, x0:x1] - G_ref).max() / max(float(np.abs(lambd * g_ref).max()), 1e-30)
        print("   %s: T %.1e, G %.1e" % (name, float(dT.max()), dG))
        assert dG < 1e-5, (name, dG)

    def key_to_float(k):
        k = int(k)
        return np.array([(k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)], np.uint32).view(np.float32)[0]
    for c in range(3):                                       # the maxima of A7 (pyx:523-524) over the whole frame, from the device's own G and u
        maxg, maxu = key_to_float(red[c]), key_to_float(red[3 + c])
        assert maxg == np.abs(G[..., c]).max() and maxu == u[..., c].max()
        dt = np.float32(step * maxu) / np.float32(maxg + np.float32(1e-15))
        un = u[..., c] - dt * G[..., c]                      # PAM: no DoF blend (oracle/rl_ext_oracle.py richardson_lucy_PAM)
        assert np.array_equal(un, u1[..., c])
    del e, T, G, u1
    # one outer iteration of the loop at this size
    import contextlib as _c, io as _io
    img, uu, pp = case["image"].copy(), u.copy(), case["psf0"].copy()
    with _c.redirect_stdout(_io.StringIO()):
        dc.richardson_lucy_MM(img, uu, pp, *orc.default_window(M, N, MK), 1e9, M, N, 3, MK, 1, float(step), float(lambd), blind=blind, tv_mode=kind)
    st = dc.richardson_lucy_MM.last
    assert st.iterations_done == 1 and not st.has_nan and np.isfinite(uu).all()
    assert np.array_equal(img, case["image"])                # PAM leaves the blurry image alone
    if blind:
        assert np.all(pp >= 0) and np.allclose(pp.sum(axis=(0, 1)), 1, atol=1e-5)
    dc._drop_jobs()
