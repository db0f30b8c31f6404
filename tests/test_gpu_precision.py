"""Dynamic range of the fp16-split matrix-core arithmetic (ics_conv_mfma.hip, ics_gradk_mfma.hip, ics_synth_gradk_mfma.hip).

Every fp32 operand x of a tile is scaled by a power of two s (tile maximum m -> [2^14, 2^15)) and split into two fp16 terms,
x s = hi + lo + r.  fp16 has 11 significand bits down to 2^-14 and a fixed quantum 2^-24 below, hence (include/ics_hip.h):

        |r| / s  <=  max( 2^-22 |x| ,  2^-39 m )                                      (*)

i.e. 22 bits for every element within 2^17 of the tile maximum, and an ABSOLUTE error of 2^-39 of the tile maximum for
smaller ones.  The weights obey the same bound with their own maximum.  Products drop lo*lo (< 2^-22 |x w|) and are
accumulated in fp32.  For an output y = sum w x this gives

        |err(y)|  <=  c1 2^-22 sum |w||x|  +  2^-38 ( m_tile sum|w| + w_max sum|x| )      c1 ~ 8 (split terms + fp32 accumulation)

The tests below drive that bound with adversarial inputs and gate on LOCAL error -- error of outputs whose support excludes
the hot pixel, relative to sum |w||x| over their own support -- not on max|d|/max|ref|, which cannot see local loss.  For
scale: the reference's own convolution (scipy FFT in complex64, lib/deconvolution.pyx:478) has an absolute error of ~1e-7
of the FRAME maximum everywhere, 4-5 orders of magnitude above the 2^-39 m term."""
import numpy as np
import pytest
from scipy.ndimage import maximum_filter
from scipy.signal import fftconvolve

import rl_mm_oracle as orc

pytestmark = pytest.mark.gpu

C1 = 8.0 * 2.0 ** -22
C2 = 2.0 ** -38


def conv_valid64(u, psf):
    return np.stack([fftconvolve(u[..., c].astype(np.float64), psf[..., c].astype(np.float64), mode="valid") for c in range(3)], -1)


def corr_full64(e, psf):
    return np.stack([fftconvolve(e[..., c].astype(np.float64), psf[::-1, ::-1, c].astype(np.float64), mode="full") for c in range(3)], -1)


def asym_psf(MK, seed):
    rng = np.random.default_rng(seed)
    psf = (orc.gaussian_psf(MK) * (0.5 + rng.random((MK, MK, 3), dtype=np.float32))).astype(np.float32)
    orc.normalize_kernel(psf, MK)
    return psf


def bound_valid(u, psf):
    """per-output bound of the forward convolution: C1 sum|w||u| + C2 (m_tile sum|w| + w_max sum|u|), m_tile taken as the
    maximum over a neighbourhood that contains every 78 x 80 staged tile the pixel can belong to"""
    MK = psf.shape[0]
    pad = MK // 2
    au = np.abs(u.astype(np.float64))
    s_wu = conv_valid64(au, np.abs(psf))
    s_u = conv_valid64(au, np.ones_like(psf))
    m_loc = np.stack([maximum_filter(au[..., c], size=(161, 193), mode="constant")[pad:-pad, pad:-pad] for c in range(3)], -1)
    sw = np.abs(psf).sum(axis=(0, 1))
    return C1 * s_wu + C2 * (m_loc * sw + np.abs(psf).max() * s_u)


def run_conv(u, psf, conv):
    """conv(u, psf) through A1 + A2 with a zero image"""
    from lib import _native as nv
    MK = psf.shape[0]
    M, N = u.shape[0] - 2 * (MK // 2), u.shape[1] - 2 * (MK // 2)
    job = nv.RLJob(M, N, MK)
    job.upload(np.zeros((M, N, 3), np.float32), u, psf)
    job.stage(nv.STAGE_SYNTH_RESIDUAL, job.params(1, 9, 1, 9, 1e9, 1, 1e-3, 1e4, False, conv=conv))
    out = job.read(nv.BUF_ERROR)
    job.close()
    return out


@pytest.mark.parametrize("MK", [15, 9, 31])
def test_hot_pixel_1e4_inside_a_tile_keeps_22_bits_elsewhere(MK):
    rng = np.random.default_rng(MK)
    M, N = 300, 330
    pad = MK // 2
    u = (0.2 + 0.6 * rng.random((M + 2 * pad, N + 2 * pad, 3), dtype=np.float32)).astype(np.float32)
    hot = [(70, 81), (150, 200), (199, 64), (260, 300)]
    for (y, x) in hot:
        u[y, x] *= np.float32(1e4)
    psf = asym_psf(MK, 3)
    ref = conv_valid64(u, psf)
    local = conv_valid64(np.abs(u), np.abs(psf))
    far = np.ones((M, N), bool)                       # outputs whose support excludes every hot pixel
    for (y, x) in hot:
        far[max(0, y - 2 * pad):y + 1, max(0, x - 2 * pad):x + 1] = False
    for conv in (2, 1):
        d = np.abs(run_conv(u, psf, conv) - ref)
        rel_far = (d / local)[far].max()
        print("MK=%d conv=%d: local relative error away from the hot pixels %.2e, at them %.2e" % (MK, conv, rel_far, (d / local)[~far].max()))
        assert rel_far < 2e-6 * max(1.0, MK / 15.0)   # 22 bits: the hot pixel is only 2^13 above the rest (fp32 accumulation of MK^2 terms)
        if conv == 2:
            assert np.all(d <= bound_valid(u, psf))


def test_tiny_residual_with_an_isolated_1_backprojection_and_gradient():
    """|e| ~ 1e-7 with a few isolated 1.0 (24 binades above): the back-projection (A3) and the PSF gradient (A13) of the
    small part keep the absolute error bound 2^-38 m, i.e. ~4e-12 for m = 1 -- the reference's complex64 FFT is at 1e-8 there."""
    from lib import _native as nv
    rng = np.random.default_rng(5)
    M, N, MK = 280, 300, 15
    pad = MK // 2
    e = (1e-7 * rng.standard_normal((M, N, 3))).astype(np.float32)
    hot = [(64, 64), (130, 201), (222, 90)]
    for (y, x) in hot:
        e[y, x] = 1.0
    u = (0.2 + 0.6 * rng.random((M + 2 * pad, N + 2 * pad, 3), dtype=np.float32)).astype(np.float32)
    psf = asym_psf(MK, 8)
    job = nv.RLJob(M, N, MK)
    job.upload(np.zeros((M, N, 3), np.float32), u, psf)
    job.write(nv.BUF_UT, u)
    far = np.ones((M + 2 * pad, N + 2 * pad), bool)
    for (y, x) in hot:
        far[y:y + 2 * pad + 1, x:x + 2 * pad + 1] = False
    g_ref = corr_full64(e, psf)
    local = corr_full64(np.abs(e), np.abs(psf))
    for conv in (2, 1):
        p = job.params(1, 9, 1, 9, 1e9, 1, 1e-3, 1e4, True, conv=conv)
        job.write(nv.BUF_ERROR, e)
        job.stage(nv.STAGE_BACKPROJECT, p)
        d = np.abs(job.read(nv.BUF_GRADU) - g_ref)
        print("conv=%d back-projection: abs err away from the 1.0s %.2e (values ~%.1e), local relative %.2e; at them %.2e" % (
            conv, d[far].max(), np.abs(g_ref[far]).mean(), (d[far] / local[far]).max(), d[~far].max()))
        assert d[far].max() < 1e-11                    # 2^-38 m sum|w| = 3.6e-12 + 22-bit term of the small values
        assert d[~far].max() < 2e-6                    # around the hot pixels: 22 bits of O(1) values
        assert np.isfinite(d).all()
        # PSF gradient: a frame-wide sum; the hot pixels contribute O(1) terms, gate relative to sum |e||u|
        job.write(nv.BUF_ERROR, e)
        job.stage(nv.STAGE_PSF_GRADIENT, p)
        gk = job.read(nv.BUF_GRADK)
        gk_ref = np.stack([fftconvolve(u[::-1, ::-1, c].astype(np.float64), e[..., c].astype(np.float64), mode="valid") for c in range(3)], -1)
        gk_abs = np.stack([fftconvolve(np.abs(u[::-1, ::-1, c]).astype(np.float64), np.abs(e[..., c]).astype(np.float64), mode="valid") for c in range(3)], -1)
        assert np.max(np.abs(gk - gk_ref) / gk_abs) < 2e-6
    job.close()


def test_sixteen_bit_range_frame():
    """frames in 0 .. 65535 (a 16-bit picture that was not normalised): same relative accuracy, no overflow of the fp16 terms"""
    rng = np.random.default_rng(9)
    M, N, MK = 260, 270, 15
    pad = MK // 2
    u = (65535.0 * rng.random((M + 2 * pad, N + 2 * pad, 3))).astype(np.float32)
    u[40:120, 50:140] *= np.float32(1e-3)              # a dark region of ~65 counts inside bright tiles
    u[200, 200] = 65535.0
    psf = asym_psf(MK, 4)
    ref = conv_valid64(u, psf)
    local = conv_valid64(np.abs(u), np.abs(psf))
    out = run_conv(u, psf, 2)
    d = np.abs(out - ref)
    assert np.isfinite(out).all()
    print("16-bit range: max|d|/max|ref| %.2e, local relative %.2e (dark region %.2e)" % (d.max() / ref.max(), (d / local).max(), (d / local)[50:100, 60:120].max()))
    assert d.max() / ref.max() < 5e-6 and (d / local).max() < 3e-6
    assert np.all(d <= bound_valid(u, psf))


def test_zero_and_denormal_tiles():
    """all-zero tiles give exact zeros; tiles of fp32 denormals give finite, tiny results (the scale saturates at 2^113)"""
    M, N, MK = 200, 210, 9
    pad = MK // 2
    u = np.zeros((M + 2 * pad, N + 2 * pad, 3), np.float32)
    u[:, 150:] = 0.5                                    # one populated band, zero tiles left of it
    u[100:140, 20:60] = np.float32(1e-40)               # denormals in otherwise empty tiles
    psf = asym_psf(MK, 2)
    ref = conv_valid64(u, psf)
    # (a float64 FFT leaks 1e-16 of the populated band everywhere: the denormal region is compared with direct sums)
    ref_den = np.stack([orc._conv_direct(u[90:150 + 2 * pad, 10:70 + 2 * pad, c], psf[..., c], "valid") for c in range(3)], -1)
    for conv in (2, 1):
        out = run_conv(u, psf, conv)
        assert np.isfinite(out).all()
        assert np.all(out[:80, :60] == 0.0)             # tiles that hold nothing but zeros
        d_den = np.abs(out[90:150, 10:70].astype(np.float64) - ref_den)
        print("conv=%d denormal tile: max |d| %.2e (values up to %.2e)" % (conv, d_den.max(), ref_den.max()))
        assert d_den.max() < 2e-41                      # 1e-40 inputs: error of a few fp32 denormal quanta (1.4e-45) ... fp16 quantum / 2^113
        assert (np.abs(out - ref)[:, 160:] / 0.5).max() < 2e-6
