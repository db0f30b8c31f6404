"""deconvolve.py call surface (SURVEY.md 8c 'deconve.py' row): pyramid schedule pinned to the integers
the reference's build_pyramid produces, mask-window arithmetic, validation errors, and the arguments the
driver passes to richardson_lucy_MM (recorded with a stub solver; no GPU needed)."""
import numpy as np
import pytest


def test_build_pyramid_matches_reference_integers():
    import deconvolve as dv
    assert dv.build_pyramid(9, 10)[1] == [9, 7, 5, 3]
    assert dv.build_pyramid(15, 10)[1] == [15, 11, 7, 5, 3]
    assert dv.build_pyramid(31, 10)[1] == [31, 21, 15, 11, 7, 5, 3]
    images, kernels = dv.build_pyramid(7, 10)
    assert kernels == [7, 5, 3] and np.allclose(images, [1, 2 ** -0.5, 0.5])
    assert dv.build_pyramid(3, 1) == ([1.], [3])


def test_pad_image_is_edge_replication_float32():
    import deconvolve as dv
    img = np.arange(2 * 3 * 3, dtype=np.float64).reshape(2, 3, 3)
    out = dv.pad_image(img, (1, 1))
    assert out.shape == (4, 5, 3) and out.dtype == np.float32 and out.flags.c_contiguous
    assert np.array_equal(out[0, 0], img[0, 0]) and np.array_equal(out[-1, -1], img[-1, -1])
    assert dv.pad_image(img, ((1, 0), (0, 0))).shape == (3, 3, 3)


def test_validation_errors():
    import deconvolve as dv
    pic = np.full((64, 64, 3), 128, np.uint8)
    with pytest.raises(ValueError, match="at least 3"):
        dv.deblur_module(pic, "x", ".", 1, save=False)
    with pytest.raises(ValueError, match="should be odd. You can use 5"):
        dv.deblur_module(pic, "x", ".", 4, save=False)
    with pytest.raises(ValueError, match="mask is outside"):
        dv.deblur_module(pic, "x", ".", 3, mask=[5, 5], mask_size=31, save=False)


def test_driver_call_shapes_recorded(capsys, monkeypatch):
    """Blind call on the mask window (deconvolve.py:277-286) then the full-frame non-blind call (:304-313)."""
    import deconvolve as dv
    import rl_mm_oracle as orc
    monkeypatch.setattr(dv.dc, "normalize_kernel", orc.normalize_kernel)   # no GPU in this test: oracle as stand-in
    calls = []

    def solver(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations, step, lambd, **kw):
        calls.append(dict(image=image.shape, u=u.shape, psf=psf.shape, win=(top, bottom, left, right), tau=tau, M=M, N=N, C=C, MK=MK,
                          it=iterations, step=step, lambd=lambd, kw=kw, contiguous=(image.flags.c_contiguous, u.flags.c_contiguous)))
        assert u.shape == (M + 2 * (MK // 2), N + 2 * (MK // 2), 3) and image.shape == (M, N, 3) and psf.shape == (MK, MK, 3)
        pad = (u.shape[0] - M) // 2
        return u[pad:pad + M, pad:pad + N]

    rng = np.random.default_rng(0)
    pic = (rng.random((90, 100, 3)) * 255).astype(np.uint8)
    out, psf = dv.deblur_module(pic, "x", ".", 5, mask=[46, 50], mask_size=41, display=False, pyramid=False, solver=solver,
                                save=False, iterations=7, tolerance=2, confidence=10)
    assert len(calls) == 2
    blind, full = calls
    # picture 90x100 -> +1 px each side = 92x102 -> made odd: 93x103; mask box 46+-20 / 50+-20 -> 40 wide -> made odd 41
    assert blind["kw"]["blind"] is True and blind["MK"] == 5 and blind["M"] == 43 and blind["N"] == 43
    assert blind["win"] == (3, 38, 3, 38) and blind["tau"] == 0 and blind["it"] == 7 and blind["step"] == 1e-3 and blind["lambd"] == 10000
    assert blind["contiguous"] == (False, False)                       # window views, as deconvolve.py:278-279
    assert full["kw"]["blind"] is False and full["M"] == 95 and full["N"] == 105 and full["tau"] == pytest.approx(0.02)
    assert full["win"] == blind["win"] and full["contiguous"] == (True, True)
    assert out.shape == (90, 100, 3) and psf.shape == (5, 5, 3)
    assert out.min() >= 0 and out.max() <= 65535


def test_mask_window_reference_tie_breaking():
    import deconvolve as dv
    assert dv.mask_window(1.0, 26, 66, 30, 70) == (25, 66, 30, 71)
    assert dv.mask_window(1.0, 10, 51, 10, 51) == (10, 51, 10, 51)
    t = dv.mask_window(2 ** -0.5, 100, 354, 120, 374)
    assert (t[1] - t[0]) % 2 == 1


@pytest.mark.gpu
def test_deblur_module_end_to_end_on_gpu(tmp_path):
    """Full driver on a synthetic blurred picture: blind estimate on the mask window then non-blind pass."""
    import deconvolve as dv
    import rl_mm_oracle as orc
    case = orc.synth_case(120, 140, 5, seed=1)
    pic = np.clip(case["image"] ** 2.2 * 255, 0, 255).astype(np.uint8)
    out, psf = dv.deblur_module(pic, "gpu", str(tmp_path), 5, mask=[60, 70], mask_size=61, display=False, iterations=4,
                                pyramid=True)
    assert out.shape == (120, 140, 3) and np.isfinite(out).all() and (tmp_path / "gpu.tif").exists()
    assert np.all(psf >= 0) and np.allclose(psf.sum(axis=(0, 1)), 1, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("pyramid,preview,blur", [(False, False, "static"), (True, False, "static"), (True, True, "static"), (False, False, "motion")])
def test_device_resident_driver_equals_the_host_driver(pyramid, preview, blur, capsys):
    """the same picture through `deblur_module` with the frames on the host (numpy pads, slices and gamma,
    one upload / download per solver call) and with the frames resident in HBM.  The two paths run the same solver on
    the same windows; they differ by float32 `powf` (device vs numpy) in the two gamma steps only."""
    import deconvolve as dv
    import rl_mm_oracle as orc
    case = orc.synth_case(118, 141, 5, seed=2)
    pic = np.clip(case["image"] ** 2.2 * 255, 0, 255).astype(np.uint8)
    kw = dict(mask=[60, 70], mask_size=61, display=False, iterations=3, pyramid=pyramid, save=False, preview=preview, blur=blur)
    out_h, psf_h = dv.deblur_module(pic, "h", ".", 5, **kw)
    log_h = capsys.readouterr().out
    out_d, psf_d = dv.deblur_module(pic, "d", ".", 5, device_resident=True, **kw)
    log_d = capsys.readouterr().out
    # the resident call leaves its wall time per phase behind (what bench.py's deblur_module_end_to_end reports)
    assert set(dv.deblur_module.last_phase_seconds) == {"blind", "non-blind"} and all(v > 0 for v in dv.deblur_module.last_phase_seconds.values())
    assert out_d.shape == out_h.shape == ((61 - 1, 61 - 1, 3) if preview else (118, 141, 3))
    assert np.abs(psf_d - psf_h).max() < 1e-5
    assert np.abs(out_d - out_h).max() / 65535 < 2e-5, np.abs(out_d - out_h).max()
    strip = lambda t: [l for l in t.splitlines() if not l.startswith("'deblur_module'") and "sec" not in l]
    assert [l.split("=")[0] for l in strip(log_h)] == [l.split("=")[0] for l in strip(log_d)]   # same progress lines


@pytest.mark.gpu
@pytest.mark.parametrize("pyramid,preview,blur", [(False, False, "static"), (True, False, "static"), (False, True, "static"), (False, False, "motion")])
def test_device_resident_driver_against_the_oracle_solver(pyramid, preview, blur, capsys):
    """SURVEY.md 8f N1 against an ORACLE run of the driver: `deblur_module(solver=<oracle richardson_lucy_MM>)` -- the host-frame
    driver with the pinned numpy restatement of lib/deconvolution.pyx doing every solver call (blind on the mask window, then
    the non-blind pass / preview; with `correlation` for motion blur) -- versus `deblur_module(device_resident=True)`, where
    frames, pads, gamma, windows and the solver all run on the device.  (The bicubic resize between pyramid levels is the
    device kernel in both runs: the reference's skimage dependency is absent from the image, parity unpinned.)"""
    import deconvolve as dv
    import rl_mm_oracle as orc
    case = orc.synth_case(118, 141, 5, seed=2)
    pic = np.clip(case["image"] ** 2.2 * 255, 0, 255).astype(np.uint8)
    kw = dict(mask=[60, 70], mask_size=61, display=False, iterations=3, pyramid=pyramid, save=False, preview=preview, blur=blur)

    def oracle_solver(image, u, psf, *args, **kwargs):
        return orc.richardson_lucy_MM(image, u, psf, *args, **kwargs)      # prints the reference's lines itself

    out_o, psf_o = dv.deblur_module(pic, "o", ".", 5, solver=oracle_solver, **kw)
    log_o = capsys.readouterr().out
    out_d, psf_d = dv.deblur_module(pic, "d", ".", 5, device_resident=True, **kw)
    log_d = capsys.readouterr().out
    assert out_d.shape == out_o.shape
    ep = np.abs(psf_d - psf_o).max() / np.abs(psf_o).max()
    eo = np.abs(out_d - out_o).max() / 65535
    print("driver vs oracle-solver driver (pyramid=%s preview=%s blur=%s): psf %.2e, picture %.2e of full scale" % (pyramid, preview, blur, ep, eo))
    assert ep < 1e-4 and eo < 1e-4
    strip = lambda t: [l.split("=")[0] for l in t.splitlines() if not l.startswith("'deblur_module'") and "sec" not in l]
    assert strip(log_o) == strip(log_d)                                    # same progress lines, same iteration counts


@pytest.mark.gpu
def test_driver_with_a_wide_blur_on_a_large_picture_takes_the_transform_tiles(debug_switch, capsys):
    """`deblur_module` on a 1500 x 1400 picture with a 21-px blur: the non-blind passes over the whole picture (2.2 Mpx, 21 x 21 and the
    pyramid's 15 x 15 ... below) are what ICS_CONV_AUTO sends to the transform tiles; the blind passes on the 255-px mask window stay
    on the matrix cores.  The same call with the tiles switched off (debug switch conv_path = matrix) must give the same picture and PSF."""
    import deconvolve as dv
    rng = np.random.default_rng(3)
    coarse = rng.random((1500 // 8 + 2, 1400 // 8 + 2, 3))
    pic = (np.repeat(np.repeat(coarse, 8, 0), 8, 1)[:1500, :1400] * 200 + 20).astype(np.uint8)
    kw = dict(mask=[750, 700], mask_size=255, display=False, iterations=3, save=False)
    out_t, psf_t = dv.deblur_module(pic, "t", ".", 21, **kw)
    capsys.readouterr()
    debug_switch("conv_path", 2)
    out_m, psf_m = dv.deblur_module(pic, "m", ".", 21, **kw)
    capsys.readouterr()
    assert out_t.shape == out_m.shape == (1500, 1400, 3) and np.isfinite(out_t).all()
    assert np.abs(psf_t - psf_m).max() < 1e-5
    assert np.abs(out_t.astype(np.float64) - out_m).max() / 65535 < 5e-5, np.abs(out_t.astype(np.float64) - out_m).max()


@pytest.mark.gpu
def test_device_image_operations_match_numpy():
    from lib import _native
    import resize_oracle as ro
    rng = np.random.default_rng(5)
    a = rng.random((37, 45, 3), dtype=np.float32)
    d = _native.DeviceImage.from_host(a)
    assert d.shape == (37, 45, 3) and np.array_equal(d.to_host(), a)
    assert np.array_equal(d.pad_edge(2, 0, 1, 3).to_host(), np.pad(a, ((2, 0), (1, 3), (0, 0)), mode="edge"))
    assert np.array_equal(d.crop(3, 30, 5, 44).to_host(), a[3:30, 5:44])
    e = d.copy()
    e.paste(4, 6, d.crop(0, 10, 0, 12))
    ref = a.copy(); ref[4:14, 6:18] = a[0:10, 0:12]
    assert np.array_equal(e.to_host(), ref)
    g = d.copy(); g.gamma(2.0, 1 / 2.2)
    assert np.abs(g.to_host() - (a / np.float32(2.0)) ** np.float32(1 / 2.2)).max() < 3e-7
    g = d.copy(); g.gamma(0.5, 2.2, 65535, clip01=True)
    assert np.abs(g.to_host() - np.clip(a / np.float32(0.5), 0, 1) ** np.float32(2.2) * np.float32(65535)).max() < 0.02
    r = d.resize(27, 31).to_host()
    assert np.abs(r - ro.resize_scipy(a, (27, 31)).astype(np.float32)).max() < 1e-6
    with pytest.raises(_native.NativeError):
        d.crop(0, 38, 0, 45)
    # 8- and 16-bit pictures are converted on the device: exactly np.float32(v)
    for dt, top in ((np.uint8, 255), (np.uint16, 65535)):
        px = rng.integers(0, top + 1, size=(37, 45, 3)).astype(dt)
        px[0, 0] = top
        di = _native.DeviceImage.from_host(px)
        assert np.array_equal(di.to_host(), px.astype(np.float32))
        di.close()
