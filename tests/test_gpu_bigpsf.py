"""PSF sizes above 63 (csrc/ics_big.hip): the reference takes any size (lib/deconvolution.pyx:341, FFT convolutions); the tuned
kernels of the library are compiled per size up to 63, beyond that run-time-sized fp32 kernels take over, up to 127; 129 ... 255 run
only as tap blocks on the matrix cores.

Stage by stage against float64 direct sums (the gate scales with the number of accumulated terms, as for the sizes <= 63 in
test_gpu_stages.py), whole runs against the pinned oracle, and the refusals (sizes above 255, the fp32 kernels above 127, the extended TV modes)."""
import contextlib
import io

import numpy as np
import pytest

import rl_mm_oracle as orc
from helpers import conv_valid64, corr_full64, gradk64, rel_err

pytestmark = pytest.mark.gpu


def make_job(M, N, MK, seed=0, blind=False):
    from lib import _native
    case = orc.synth_case(M, N, MK, seed=seed, blind=blind, per_channel_psf=True)
    rng = np.random.default_rng(seed + 1)
    psf = (case["psf0"] * (0.5 + rng.random(case["psf0"].shape, dtype=np.float32))).astype(np.float32)   # no symmetry: flips are detected
    orc.normalize_kernel(psf, MK)
    job = _native.RLJob(M, N, MK)
    job.upload(case["image"], case["u0"], psf)
    return job, case, psf


@pytest.mark.parametrize("conv", [0, 1, 3])
@pytest.mark.parametrize("MK,M,N", [(65, 150, 131), (67, 40, 300), (99, 97, 70), (127, 140, 150), (127, 31, 33)])
def test_big_psf_stages_against_float64_direct_sums(MK, M, N, conv):
    """conv = 0 (ICS_CONV_AUTO; these frames are below the tiles' thresholds): tap blocks on the matrix cores -- convolutions as blocks of
    <= 33 x 33 taps (do_conv_blocks), the gradient as blocks of <= 31 x 31 (do_gradk_split); conv = 1 (ICS_CONV_VECTOR): the run-time-sized
    fp32 kernels of ics_big.hip; conv = 3 (ICS_CONV_FFT, round 6): the transform tiles -- one tile to 97, tap blocks whose products meet in
    the frequency domain above (k_conv_fft_blk, lag blocks of k_gradk_fft)."""
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=MK + M)
    rng = np.random.default_rng(7)
    u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(2, M - 2, 2, N - 2, 1e9, 1, 1e-3, 10000.0, blind=True, conv=conv)
    tol = 5e-6 * (MK / 31.0) ** 2 / 4          # rows of the kernel are summed on their own: a quarter of the plain chain's bound
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    synth = conv_valid64(u, psf)
    assert np.max(np.abs(e - (synth - case["image"]))) / np.max(np.abs(synth)) < tol
    job.stage(nv.STAGE_BACKPROJECT, p)
    g = job.read(nv.BUF_GRADU)
    g_ref = corr_full64(e.astype(np.float64), psf)
    assert g.shape == g_ref.shape
    assert rel_err(g, g_ref) < tol
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk = job.read(nv.BUF_GRADK)
    gk_ref = gradk64(u.astype(np.float64), e.astype(np.float64))
    assert gk.shape == gk_ref.shape == (MK, MK, 3)
    assert rel_err(gk, gk_ref) < 1e-5
    # the maxima of A7 (at these sizes a pass of its own behind the back-projection): the update pass records what it used
    job.stage(nv.STAGE_UPDATE, p)
    sc = job.scalars()
    gfull = (np.float32(10000.0) * g + (u - case["u0"]) * np.float32(0.5)).astype(np.float32)
    for c in range(3):
        assert np.float32(sc["maxg%d" % c]) == np.max(np.abs(gfull[..., c]))
        assert np.float32(sc["maxu%d" % c]) == np.max(u[..., c])
    job.close()


@pytest.mark.parametrize("path", ["auto", "vector"])
@pytest.mark.parametrize("MK,M,N,blind", [(51, 90, 130, True), (65, 120, 110, False), (65, 120, 110, True), (71, 100, 150, True), (127, 160, 170, True)])
def test_big_psf_runs_against_the_oracle(MK, M, N, blind, path, debug_switch):
    from lib import deconvolution as dc
    debug_switch("conv_path", 1 if path == "vector" else 0)      # what ICS_CONV_AUTO resolves to (the drop-in call has no conv argument)
    case = orc.synth_case(M, N, MK, seed=3 + MK, blind=blind)
    win = (8, M - 10, 8, N - 10)
    args = (*win, 1e9, M, N, 3, MK, 2, 1e-3, 10000.0)
    u_ref, psf_ref = case["u0"].copy(), case["psf0"].copy()
    orc.richardson_lucy_MM(case["image"].copy(), u_ref, psf_ref, *args, blind=blind, quiet=True)
    u, psf = case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *args, blind=blind)
    assert rel_err(u, u_ref) < 1e-4
    assert rel_err(psf, psf_ref) < 1e-4


def _fft64(a, b, mode):
    from scipy.signal import fftconvolve
    return np.stack([fftconvolve(a[..., c].astype(np.float64), b[..., c].astype(np.float64), mode=mode) for c in range(3)], axis=-1)


@pytest.mark.parametrize("conv", [0, 2, 3])
@pytest.mark.parametrize("MK,M,N", [(129, 140, 150), (133, 64, 200), (191, 33, 300), (255, 90, 70), (255, 300, 280)])
def test_psf_129_to_255_stages_against_float64(MK, M, N, conv):
    """PSF sizes above 127 (csrc/ics_api.hip psf_blocks_only): convolutions as up to 8 x 8 blocks of <= 33 x 33 taps, the gradient as up
    to 9 x 9 blocks of <= 31 x 31, nothing else behind them.  Against float64 FFT products (direct float64 sums of 65 025 taps per
    output value take minutes in numpy; a float64 FFT is exact to ~1e-15 of the largest value), same gates as the sizes below."""
    from lib import _native as nv
    job, case, psf = make_job(M, N, MK, seed=MK + M)
    rng = np.random.default_rng(7)
    u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
    job.write(nv.BUF_U, u)
    job.write(nv.BUF_UT, case["u0"])
    p = job.params(2, M - 2, 2, N - 2, 1e9, 1, 1e-3, 10000.0, blind=True, conv=conv)
    tol = 5e-6 * (MK / 31.0) ** 2 / 4
    job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
    e = job.read(nv.BUF_ERROR)
    synth = _fft64(u, psf, "valid")
    assert np.max(np.abs(e - (synth - case["image"]))) / np.max(np.abs(synth)) < tol
    job.stage(nv.STAGE_BACKPROJECT, p)
    g = job.read(nv.BUF_GRADU)
    g_ref = _fft64(e, psf[::-1, ::-1], "full")
    assert g.shape == g_ref.shape
    assert rel_err(g, g_ref) < tol
    job.stage(nv.STAGE_PSF_GRADIENT, p)
    gk = job.read(nv.BUF_GRADK)
    gk_ref = _fft64(u[::-1, ::-1], e, "valid")
    assert gk.shape == gk_ref.shape == (MK, MK, 3)
    assert rel_err(gk, gk_ref) < 1e-5
    job.stage(nv.STAGE_UPDATE, p)
    sc = job.scalars()
    gfull = (np.float32(10000.0) * g + (u - case["u0"]) * np.float32(0.5)).astype(np.float32)
    for c in range(3):
        assert np.float32(sc["maxg%d" % c]) == np.max(np.abs(gfull[..., c]))
        assert np.float32(sc["maxu%d" % c]) == np.max(u[..., c])
    job.close()


@pytest.mark.parametrize("MK,M,N,blind", [(129, 150, 140, True), (129, 150, 140, False), (201, 100, 260, True), (255, 64, 80, True), (255, 270, 300, False)])
def test_psf_129_to_255_runs_against_the_oracle(MK, M, N, blind):
    from lib import deconvolution as dc
    case = orc.synth_case(M, N, MK, seed=3 + MK, blind=blind)
    win = (8, M - 10, 8, N - 10)
    args = (*win, 1e9, M, N, 3, MK, 2, 1e-3, 10000.0)
    u_ref, psf_ref = case["u0"].copy(), case["psf0"].copy()
    orc.richardson_lucy_MM(case["image"].copy(), u_ref, psf_ref, *args, blind=blind, quiet=True)
    u, psf = case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *args, blind=blind)
    assert rel_err(u, u_ref) < 1e-4
    assert rel_err(psf, psf_ref) < 1e-4


@pytest.mark.parametrize("MK,M,N", [(129, 108, 75), (253, 179, 163), (255, 60, 50)])
def test_psf_above_127_on_frames_smaller_than_the_psf_against_float64_convolutions(MK, M, N, monkeypatch):
    """Non-blind, two outer iterations, frames smaller than the PSF: here the reference's own complex64 FFT noise reaches 1e-4 ... 3e-4 of
    the result (scripts/dbg/fuzz_runs.py ... big, FUZZ_F64=1: device 1.3e-5 / 2.2e-6 from the float64 trajectory, the reference 3.1e-4 /
    1.4e-4), so the pinned oracle is no yardstick at the 1e-4 gate; the same loop with float64 convolutions is."""
    from scipy.signal import fftconvolve
    from lib import deconvolution as dc
    case = orc.synth_case(M, N, MK, seed=MK + N, blind=False)
    args = (M // 4, M - M // 4, N // 8, N - N // 8, 1e9, M, N, 3, MK, 2, 1e-3, 10000.0)
    monkeypatch.setattr(orc, "_conv_direct", lambda a, b, mode: fftconvolve(np.asarray(a, np.float64), np.asarray(b, np.float64), mode=mode))
    u64, psf64 = case["u0"].copy(), case["psf0"].copy()
    orc.richardson_lucy_MM(case["image"].copy(), u64, psf64, *args, blind=False, quiet=True, conv="direct")
    u, psf = case["u0"].copy(), case["psf0"].copy()
    with contextlib.redirect_stdout(io.StringIO()):
        dc.richardson_lucy_MM(case["image"].copy(), u, psf, *args, blind=False)
    assert not np.isnan(u).any()
    assert rel_err(u, u64) < 5e-5


def test_big_psf_refusals():
    from lib import _native as nv
    with pytest.raises(nv.NativeError, match="3..255"):
        nv.RLJob(64, 64, 257)
    job = nv.RLJob(64, 64, 129)
    case = orc.synth_case(64, 64, 129, seed=1)
    job.upload(case["image"], case["u0"], case["psf0"])
    with pytest.raises(nv.NativeError, match="above 127 only run as tap blocks"):
        job.run(job.params(4, 60, 4, 60, 1e9, 1, 1e-3, 10000.0, False, conv=1))
    job.close()
    job = nv.RLJob(64, 64, 65)
    case = orc.synth_case(64, 64, 65, seed=1)
    job.upload(case["image"], case["u0"], case["psf0"])
    for tv in (1, 2, 3):
        with pytest.raises(nv.NativeError, match="PSF sizes <= 63"):
            job.run(job.params(4, 60, 4, 60, 1e9, 1, 1e-3, 10000.0, False, tv_mode=tv))
    with pytest.raises(nv.NativeError, match="fuse"):
        job.run(job.params(4, 60, 4, 60, 1e9, 1, 1e-3, 10000.0, False, fuse=1))
    job.close()


@pytest.mark.parametrize("MK", [45, 51, 59, 63])
def test_auto_path_above_37_against_the_kernels_compiled_per_size(MK):
    """ICS_CONV_AUTO (csrc/ics_api.hip): matrix-core kernels to 49 x 49, ics_big.hip (use_big_conv) above; ICS_CONV_VECTOR keeps the
    packed-fp32 kernels compiled per size.  Both against float64 direct sums with the same gate, and against each other."""
    from lib import _native as nv
    M, N = 70 + MK, 131
    out = {}
    for conv in (0, 1):
        job, case, psf = make_job(M, N, MK, seed=MK)
        rng = np.random.default_rng(7)
        u = (case["u0"] + 0.05 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
        job.write(nv.BUF_U, u)
        job.write(nv.BUF_UT, case["u0"])
        p = job.params(2, M - 2, 2, N - 2, 1e9, 1, 1e-3, 10000.0, blind=False, conv=conv)
        job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
        e = job.read(nv.BUF_ERROR)
        synth = conv_valid64(u, psf)
        tol = 5e-6 * (MK / 31.0) ** 2
        assert np.max(np.abs(e - (synth - case["image"]))) / np.max(np.abs(synth)) < tol
        job.stage(nv.STAGE_BACKPROJECT, p)
        g = job.read(nv.BUF_GRADU)
        assert rel_err(g, corr_full64(e.astype(np.float64), psf)) < tol
        job.stage(nv.STAGE_UPDATE, p)
        sc = job.scalars()
        out[conv] = (e, g, job.read(nv.BUF_U), [sc["maxg%d" % c] for c in range(3)], [sc["maxu%d" % c] for c in range(3)])
        job.close()
    assert np.max(np.abs(out[0][0] - out[1][0])) / np.max(np.abs(synth)) < 2 * tol   # (the residual is small against the synthesis it is the difference of)
    assert rel_err(out[0][1], out[1][1]) < 4 * tol
    assert rel_err(out[0][2], out[1][2]) < 1e-5
    assert out[0][4] == out[1][4]


@pytest.mark.parametrize("MK,M,N", [(33, 97, 140), (39, 150, 131), (45, 70, 200), (47, 129, 129), (49, 160, 90)])
def test_split_matrix_core_gradient_33_to_49(MK, M, N):
    """csrc/ics_api.hip do_gradk_split: the K x K taps as four blocks (rows / columns [0, 31) and [31, K)) on the 31 x 31 / 15 x 15
    fp16-split kernel with shifted frame pointers.  Against float64 direct sums (the gate of every gradient kernel) and against the
    fp32-MFMA kernel (conv = ICS_CONV_VECTOR) on the same inputs."""
    from lib import _native as nv
    out, e = {}, None
    for conv in (0, 1):
        job, case, psf = make_job(M, N, MK, seed=MK + N, blind=True)
        rng = np.random.default_rng(5)
        u = (case["u0"] + 0.02 * rng.standard_normal(case["u0"].shape)).astype(np.float32)
        job.write(nv.BUF_U, u)
        p = job.params(2, M - 2, 2, N - 2, 1e9, 1, 1e-3, 10000.0, blind=True, conv=conv)
        if e is None:
            job.stage(nv.STAGE_SYNTH_RESIDUAL, p)
            e = job.read(nv.BUF_ERROR)
        else:
            job.write(nv.BUF_ERROR, e)       # the same residual for both gradient kernels (it is a small difference of large numbers)
        job.stage(nv.STAGE_PSF_GRADIENT, p)
        gk = job.read(nv.BUF_GRADK)
        assert rel_err(gk, gradk64(u.astype(np.float64), e.astype(np.float64))) < 1e-5
        out[conv] = gk
        job.close()
    assert rel_err(out[0], out[1]) < 2e-5
