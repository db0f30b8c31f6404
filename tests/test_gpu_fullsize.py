"""Size-independent properties at BASELINE.json's full sizes (configs[2]: 4096x4096x3 / 15x15, configs[3]:
6144x6144x3 / 31x31), where the oracle is too slow to run: adjointness of the two convolutions, linearity,
constant-image fixed point, the PSF gradient against direct dot products, the update formula on crops
(teacher-forced with the device's own back-projection and step size), simplex constraint of the PSF, and
agreement of a full blind run's window statistics with numpy evaluated on the downloaded frames."""
import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import update_f32

pytestmark = pytest.mark.gpu


def rand_psf(MK, seed):
    rng = np.random.default_rng(seed)
    psf = (orc.gaussian_psf(MK) * (0.5 + rng.random((MK, MK, 3), dtype=np.float32))).astype(np.float32)
    orc.normalize_kernel(psf, MK)
    return psf


def dot64(a, b):
    return float(np.sum(a.astype(np.float64) * b.astype(np.float64)))


@pytest.mark.parametrize("S,MK", [(4096, 15), (6144, 31)])
def test_full_size_convolution_properties(S, MK):
    from lib import _native as nv
    rng = np.random.default_rng(S)
    pad = MK // 2
    job = nv.RLJob(S, S, MK)
    psf = rand_psf(MK, 1)
    u = rng.random((S + 2 * pad, S + 2 * pad, 3), dtype=np.float32)
    zero_img = np.zeros((S, S, 3), np.float32)
    job.upload(zero_img, u, psf)
    p = job.params(pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1, 1e9, 1, 1e-3, 10000.0, blind=True)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    synth = job.read(nv.BUF_ERROR)                       # image = 0  ->  error = conv(u, psf)
    # (1) spot check against direct float64 sums on three crops (corner, centre, far corner)
    for (y, x) in [(0, 0), (S // 2 - 8, S // 2 + 3), (S - 16, S - 16)]:
        ref = np.stack([orc._conv_direct(u[y:y + 16 + 2 * pad, x:x + 16 + 2 * pad, c], psf[..., c], "valid") for c in range(3)], -1)
        assert np.max(np.abs(synth[y:y + 16, x:x + 16] - ref)) < 5e-6 * np.max(np.abs(ref))
    # (2) adjointness: <conv(u), e> == <u, corr_full(e)>
    e = rng.standard_normal((S, S, 3), dtype=np.float32)
    job.write(nv.BUF_ERROR, e)
    job.write(nv.BUF_UT, u)
    job.stage(nv.STAGE_BACKPROJECT, p)
    g = job.read(nv.BUF_GRADU)
    lhs, rhs = dot64(synth, e), dot64(u, g)
    assert abs(lhs - rhs) <= 2e-6 * (np.sqrt(dot64(synth, synth) * dot64(e, e)))
    # (3) the fused reductions saw the whole frame: max u per channel is exact
    job.stage(nv.STAGE_UPDATE, p)
    sc = job.scalars()
    for k in range(3):
        assert sc["maxu%d" % k] == float(np.max(u[..., k]))
    # (4) PSF gradient: gradk[a, b, c] = <e, u shifted>, a few taps against float64 dot products
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_ERROR, e)
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk = job.read(nv.BUF_GRADK)
    for (a, b, c) in [(0, 0, 0), (MK - 1, MK - 1, 2), (pad, pad, 1), (1, MK - 2, 0)]:
        ref = dot64(e[..., c], u[MK - 1 - a:MK - 1 - a + S, MK - 1 - b:MK - 1 - b + S, c])
        assert abs(gk[a, b, c] - ref) < 2e-5 * np.sqrt(dot64(e[..., c], e[..., c]) * S * S / 3), (a, b, c)
    # (5) linearity: conv(2u) = 2 conv(u) exactly in binary floating point
    job.write(nv.BUF_U, (2.0 * u).astype(np.float32))
    job.write(nv.BUF_IMAGE, zero_img)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    assert np.array_equal(job.read(nv.BUF_ERROR), 2.0 * synth)
    # (6) a constant image is a fixed point of a normalised PSF
    job.write(nv.BUF_U, np.full_like(u, 0.375))
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    c = job.read(nv.BUF_ERROR)
    assert np.max(np.abs(c - 0.375)) < 3e-6
    job.close()


def test_config3_blind_4096_one_outer_iteration_consistency():
    """BASELINE.json configs[2] shape: one full outer iteration (5 inner) of the blind loop at 4096^2/15x15,
    then every quantity that can be re-derived from the downloaded frames is re-derived with numpy."""
    from lib import _native as nv
    import bench
    S, MK = 4096, 15
    pad = MK // 2
    image, u0, psf_true, psf_uniform = bench.synth_frame(S, S, MK, seed=3)
    job = nv.RLJob(S, S, MK)
    job.upload(image, u0, psf_uniform)
    win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)
    st = job.run(job.params(*win, 0.0, 1, 1e-3, 10000.0, blind=True, stop_test=1))
    assert st.iterations_done == 1 and st.inner_iterations == 5 and not st.has_nan
    u, psf, psf_caller = job.download()
    e = job.read(nv.BUF_ERROR)
    assert np.array_equal(psf, psf_caller)
    assert np.all(psf >= 0) and np.allclose(psf.astype(np.float64).sum(axis=(0, 1)), 1, atol=2e-6)   # simplex (A16)
    assert not np.allclose(psf, psf_uniform)                                                          # PSF was refined
    # residual consistency: error == conv(u, psf) - image on crops (A11 ran on the final u with the previous psf,
    # so re-run A11 with the final psf through the stage API and compare that)
    p = job.params(*win, 0.0, 1, 1e-3, 10000.0, blind=True)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e2 = job.read(nv.BUF_ERROR)
    for (y, x) in [(5, 9), (2000, 2100), (4070, 4060)]:
        ref = np.stack([orc._conv_direct(u[y:y + 20 + 2 * pad, x:x + 20 + 2 * pad, c], psf[..., c], "valid") for c in range(3)], -1) - image[y:y + 20, x:x + 20]
        assert np.max(np.abs(e2[y:y + 20, x:x + 20] - ref)) < 5e-6
    # window statistics of the run (A18/A19) against numpy on the frames of the run
    top, bottom, left, right = win
    ew = e[top:bottom, left:right]
    M_r = orc.residual_whiteness(ew, orc.stop_weights(*win), orc._conv_scipy)
    Hu = np.linalg.norm(ew) ** 2 / ((bottom - top) * (right - left) * 3)
    varu = np.std(u[top + pad:bottom - pad, left + pad:right - pad]) ** 2
    assert abs(st.M_r - M_r) / M_r < 1e-3 and abs(st.Hu - Hu) / Hu < 1e-4 and abs(st.varu - varu) / varu < 1e-4
    job.close()


def test_update_formula_on_crops_at_4096():
    """A5-A10 at full size: teacher-forced with the device's own back-projection and step size, crops are
    compared bit for bit with the numpy float32 restatement."""
    from lib import _native as nv
    import bench
    S, MK = 4096, 15
    pad = MK // 2
    image, u0, psf_true, _ = bench.synth_frame(S, S, MK, seed=5)
    rng = np.random.default_rng(0)
    u = (u0 + np.float32(0.02) * rng.standard_normal(u0.shape, dtype=np.float32)).astype(np.float32)
    job = nv.RLJob(S, S, MK)
    job.upload(image, u, psf_true)
    job.write(nv.BUF_UT, u0)
    p = job.params(pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1, 1e9, 1, 1e-3, 10000.0, blind=False)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    job.stage(nv.STAGE_BACKPROJECT, p)
    g_raw = job.read(nv.BUF_GRADU)
    job.stage(nv.STAGE_UPDATE, p)
    u_dev = job.read(nv.BUF_U)
    u_ref, dt, DoF = update_f32(u, u0, g_raw, image, 1e-3, 10000.0, False, pad)
    sc = job.scalars()
    assert [sc["dt0"], sc["dt1"], sc["dt2"]] == [float(x) for x in dt]
    assert np.array_equal(u_dev, u_ref, equal_nan=True)
    job.close()


# ---- whole frames against float64 FFT convolutions (scipy.signal.fftconvolve) --------------------------------------------
def _fft_valid64(u, psf):
    from scipy.signal import fftconvolve
    return np.stack([fftconvolve(u[..., c].astype(np.float64), psf[..., c].astype(np.float64), mode="valid") for c in range(3)], -1)


def _fft_full_corr64(e, psf):
    from scipy.signal import fftconvolve
    rot = psf[::-1, ::-1]
    return np.stack([fftconvolve(e[..., c].astype(np.float64), rot[..., c].astype(np.float64), mode="full") for c in range(3)], -1)


def _fft_gradk64(u, e):
    """gradk = convolve(rot180(u), e, "valid") (pyx:567-571) in float64."""
    from scipy.signal import fftconvolve
    return np.stack([fftconvolve(u[::-1, ::-1, c].astype(np.float64), e[..., c].astype(np.float64), mode="valid") for c in range(3)], -1)


@pytest.mark.parametrize("S,MK,conv", [(2048, 15, 2), (2048, 15, 1), (4096, 15, 2), (4096, 15, 1), (6144, 31, 2)])
def test_whole_frame_stage_pass_against_float64_fft(S, MK, conv):
    """BASELINE.json configs[1..3] sizes: the ENTIRE residual, back-projection and PSF gradient of one stage pass against
    float64 convolutions (every tile of the persistent walk, the band split, the next-tile prefetch and the interior-origin
    grid are covered, not crops).  Gates: 5e-6 of the frame maximum (7e-6 scaled by (MK/31)^2 above 31), and the number of
    pixels beyond 1e-6 of the maximum is reported.  conv = 2: matrix-core kernels (fp16-split), 1: fp32 products."""
    from lib import _native as nv
    import bench
    pad = MK // 2
    image, u0, psf_true, _ = bench.synth_frame(S, S, MK, seed=S + MK)
    psf = rand_psf(MK, 7)
    rng = np.random.default_rng(2)
    u = (u0 + np.float32(0.02) * rng.standard_normal(u0.shape, dtype=np.float32)).astype(np.float32)
    job = nv.RLJob(S, S, MK)
    job.upload(image, u, psf)
    job.write(nv.BUF_UT, u0)
    p = job.params(pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1, 1e9, 1, 1e-3, 10000.0, blind=True, conv=conv)
    # A1 + A2
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    synth = _fft_valid64(u, psf)
    d = np.abs(e - (synth - image)) / np.max(np.abs(synth))
    print("%d^2 k%d conv=%d residual: max %.2e, fraction > 1e-6: %.2e" % (S, MK, conv, d.max(), np.mean(d > 1e-6)))
    assert d.max() < 5e-6
    del synth, d
    # A3 on the device's own residual
    job.stage(nv.STAGE_BACKPROJECT, p)
    g = job.read(nv.BUF_GRADU)
    g_ref = _fft_full_corr64(e, psf)
    d = np.abs(g - g_ref) / np.max(np.abs(g_ref))
    print("   back-projection: max %.2e, fraction > 1e-6: %.2e" % (d.max(), np.mean(d > 1e-6)))
    assert g.shape == g_ref.shape and d.max() < 5e-6
    del g, g_ref, d
    # A13 (two-kernel path) and, where it exists, the fused A11 + A13 kernel
    gk_ref = _fft_gradk64(u, e)
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk = job.read(nv.BUF_GRADK)
    err = np.max(np.abs(gk - gk_ref)) / np.max(np.abs(gk_ref))
    print("   PSF gradient: %.2e of max|gradk|" % err)
    assert err < 1e-5
    if MK <= 15 and conv == 2:
        job.stage(nv.STAGE_SYNTH_GRADK, p)
        e2, gk2 = job.read(nv.BUF_ERROR), job.read(nv.BUF_GRADK)
        assert np.max(np.abs(e2 - e)) / np.max(np.abs(u)) < 1e-6
        err2 = np.max(np.abs(gk2 - _fft_gradk64(u, e2))) / np.max(np.abs(gk_ref))
        print("   fused A11 + A13: PSF gradient %.2e of max|gradk|" % err2)
        assert err2 < 1e-5
    job.close()


@pytest.mark.parametrize("MK", [15, 9])
def test_largest_accepted_frame_offsets_near_2_gib(MK):
    """A 12544 x 12544 x 3 frame is 1.9 GB per buffer -- just under the 2 GiB that the kernels' 32-bit byte offsets reach
    (tests/test_gpu_edges.py: beyond it the job is refused).  Crops at the far corner against float64 direct sums for the
    synthesis, the back-projection and the fused A11 + A13 pass; the step-size reduction saw every pixel.  (MK = 15 runs the
    32-row tiles, MK = 9 at this size the 64-row ones.)"""
    from lib import _native as nv
    S = 12544
    pad = MK // 2
    rng = np.random.default_rng(MK)
    psf = rand_psf(MK, 3)
    u = rng.random((S + 2 * pad, S + 2 * pad, 3), dtype=np.float32)
    job = nv.RLJob(S, S, MK)
    image = np.zeros((S, S, 3), np.float32)
    image[-40:, -40:] = rng.random((40, 40, 3), dtype=np.float32)
    job.upload(image, u, psf)
    del image
    p = job.params(pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1, 1e9, 1, 1e-3, 10000.0, blind=True)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    spots = [(0, 0), (S // 2 + 5, S - 16), (S - 16, S // 3), (S - 16, S - 16)]

    def conv_crop(y, x):
        return np.stack([orc._conv_direct(u[y:y + 16 + 2 * pad, x:x + 16 + 2 * pad, c], psf[..., c], "valid") for c in range(3)], -1)

    for (y, x) in spots:
        e = job.read_rows(nv.BUF_ERROR, y, 16)[:, x:x + 16]
        ref = conv_crop(y, x)
        if y == S - 16 and x == S - 16:
            ref = ref - job.read_rows(nv.BUF_IMAGE, y, 16)[:, x:x + 16]
        assert np.max(np.abs(e - ref)) < 5e-6 * np.max(np.abs(conv_crop(y, x))), (y, x)
    # back-projection of that residual: rows near the end of the u frame, from the residual rows that reach them
    job.stage(nv.STAGE_MAJORIZE, p)                          # ut = u on the device
    job.stage(nv.STAGE_BACKPROJECT, p)
    y0 = S + 2 * pad - 24                                   # u-frame rows [y0, y0 + 24)
    g = job.read_rows(nv.BUF_GRADU, y0, 24)
    er = job.read_rows(nv.BUF_ERROR, y0 - 2 * pad, S - (y0 - 2 * pad)).astype(np.float64)   # residual rows y0 - 2 pad .. S - 1
    for x in (0, S // 2 + 7, S + 2 * pad - 20):
        # gradu[Y, X] = sum_{a, b} psf[a, b] * e[Y - 2 pad + a, X - 2 pad + b]  (zero outside the M x N residual)
        ep = np.zeros((24 + 2 * pad, 20 + 2 * pad, 3))
        for i in range(ep.shape[0]):
            Yi = y0 - 2 * pad + i
            if Yi >= S:
                continue
            for jx in range(ep.shape[1]):
                Xj = x - 2 * pad + jx
                if 0 <= Xj < S:
                    ep[i, jx] = er[Yi - (y0 - 2 * pad), Xj]
        ref = np.zeros((24, 20, 3))
        for a in range(MK):
            for b in range(MK):
                ref += psf[a, b].astype(np.float64) * ep[a:a + 24, b:b + 20]
        assert np.max(np.abs(g[:, x:x + 20] - ref)) < 5e-6 * max(np.max(np.abs(ref)), 1e-3), x
    del er, g
    # the reductions of the back-projection cover the whole frame: max u per channel is exact
    job.stage(nv.STAGE_UPDATE, p)
    sc = job.scalars()
    for k in range(3):
        assert sc["maxu%d" % k] == float(np.max(u[..., k]))
    # fused A11 + A13 on the original u: the residual it leaves where the statistics read it, and two taps of the gradient
    job.write_rows(nv.BUF_U, 0, u)
    job.stage(nv.STAGE_SYNTH_GRADK, p)
    gk = job.read(nv.BUF_GRADK)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    for (a, b, c) in [(0, 0, 0), (MK - 1, pad, 2)]:
        acc = 0.0
        for r0 in range(0, S, 1568):                        # float64 dot product in row blocks (bounded host memory)
            eb = job.read_rows(nv.BUF_ERROR, r0, 1568)[..., c].astype(np.float64)
            acc += float(np.sum(eb * u[MK - 1 - a + r0:MK - 1 - a + r0 + 1568, MK - 1 - b:MK - 1 - b + S, c]))
        ee = 0.25 * S * S                                   # |e|^2 scale: residual ~ conv(u) ~ 0.5
        assert abs(gk[a, b, c] - acc) < 2e-5 * np.sqrt(ee * S * S / 3), (a, b, c, gk[a, b, c], acc)
    job.close()
