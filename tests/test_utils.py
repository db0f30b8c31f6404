"""lib.utils surface: window kernels on the host, blurs / USM / bilateral on
the GPU, against tests/golden/utils.npz (outputs of the reference's lib/utils.py) and the oracle."""
import json
import os

import numpy as np
import pytest

import utils_oracle as uo


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "utils.npz"))


def test_oracle_matches_reference_outputs(gold):
    src = gold["src"]
    assert np.array_equal(uo.gaussian_blur(src, 7, 1.5), gold["gaussian_blur_7_1.5"])
    assert np.array_equal(uo.bessel_blur(src, 9, 4.0), gold["bessel_blur_9_4"])
    assert np.array_equal(uo.gaussian_blur(src, 4, 1.0), gold["gaussian_blur_4_1"])
    assert np.array_equal(uo.USM(src, 5, 3.0, 0.7), gold["usm_bessel_5_3_0.7"])
    assert np.array_equal(uo.USM(src, 5, 1.2, 1.5, method="gauss"), gold["usm_gauss_5_1.2_1.5"])
    assert "NameError" in json.loads(str(gold["meta"]))["bilateral_in_reference"]  # broken as shipped -> unpinned


def test_host_kernels_match_reference(gold):
    from lib import utils
    for size in (3, 7, 15):
        assert np.array_equal(utils.uniform_kernel(size), gold["uniform_%d" % size])
        np.testing.assert_allclose(utils.gaussian_kernel(size, size / 6.0), gold["gaussian_%d" % size], rtol=1e-15, atol=0)
        assert np.array_equal(utils.kaiser_kernel(size, 3.5), gold["kaiser_%d" % size])
        np.testing.assert_allclose(utils.poisson_kernel(size, 2.0), gold["poisson_%d" % size], rtol=1e-15, atol=0)
    assert np.array_equal(utils.lens_blur(9), gold["lens_9"])


def test_timeit_and_fft_convolve_and_save(tmp_path, capsys):
    from lib import utils

    @utils.timeit
    def f(x):
        return x + 1
    assert f(1) == 2 and "'f'" in capsys.readouterr().out
    rng = np.random.default_rng(0)
    a, b = rng.random((20, 17)), rng.random((5, 3))
    from scipy.signal import convolve2d
    np.testing.assert_allclose(utils.convolve(a, b, "full"), convolve2d(a, b, mode="full"), atol=1e-12)
    pic = (rng.random((9, 7, 3)) * 65535)
    utils.save(pic, "t", str(tmp_path))
    from PIL import Image
    raw = open(tmp_path / "t.tif", "rb").read()
    assert raw[:4] == b"II*\x00" and len(raw) > 9 * 7 * 6
    data = np.frombuffer(raw[-9 * 7 * 6:], dtype="<u2").reshape(9, 7, 3)
    assert np.array_equal(data, pic.astype(np.uint16))


def test_bilateral_oracle_properties():
    """parity unpinned (reference raises NameError): property tests of the restatement."""
    rng = np.random.default_rng(1)
    const = np.full((12, 9), 0.37)
    assert np.allclose(uo.bilateral_filter(const, 2, 0.1, 1.0), const)
    src = rng.random((15, 13))
    # std_i -> infinity: plain spatial Gaussian with symmetric padding
    r, ss = 2, 1.3
    ker = np.array([[np.exp(-(i * i + j * j) / (2 * ss * ss)) for j in range(-r, r + 1)] for i in range(-r, r + 1)])
    ker /= ker.sum()
    np.testing.assert_allclose(uo.bilateral_filter(src, r, 1e9, ss), uo.conv2d_symm(src, ker), rtol=1e-9)


@pytest.mark.gpu
def test_gpu_blurs_and_usm_match_reference(gold):
    from lib import utils
    src = gold["src"]
    tol = dict(rtol=1e-12, atol=1e-14)  # float64 sums in a different order than scipy's C loop
    np.testing.assert_allclose(utils.gaussian_blur(src, 7, 1.5), gold["gaussian_blur_7_1.5"], **tol)
    np.testing.assert_allclose(utils.bessel_blur(src, 9, 4.0), gold["bessel_blur_9_4"], **tol)
    np.testing.assert_allclose(utils.gaussian_blur(src, 4, 1.0), gold["gaussian_blur_4_1"], **tol)
    np.testing.assert_allclose(utils.USM(src, 5, 3.0, 0.7), gold["usm_bessel_5_3_0.7"], **tol)
    np.testing.assert_allclose(utils.USM(src, 5, 1.2, 1.5, method="gauss"), gold["usm_gauss_5_1.2_1.5"], **tol)


@pytest.mark.gpu
def test_gpu_bilateral_matches_oracle_and_properties():
    from lib import utils
    rng = np.random.default_rng(2)
    src = rng.random((70, 53))
    np.testing.assert_allclose(utils.bilateral_filter(src, 3, 0.2, 1.5), uo.bilateral_filter(src, 3, 0.2, 1.5), rtol=1e-11)
    const = np.full((20, 31), 0.25)
    np.testing.assert_allclose(utils.bilateral_filter(const, 2, 0.1, 1.0), const, rtol=1e-14)
    np.testing.assert_allclose(utils.bilateral_filter(src, 0, 0.2, 1.5), src, rtol=1e-14)   # radius 0 = identity


@pytest.mark.gpu
@pytest.mark.parametrize("order,norm", [(2, 1), (2, 2), (1, 1), (1, 2)])
def test_gpu_tv_stencil_matches_oracle(ctx, order, norm):
    """A4 (lib/deconvolution.pyx:137-239): all four order/norm variants; borders stay untouched (zero)."""
    import rl_mm_oracle as orc
    rng = np.random.default_rng(order * 10 + norm)
    M, N = 37, 53
    u = rng.random((M, N, 3), dtype=np.float32)
    for eps in (1e-2, 1e-6):
        out, div = ctx.tv(u, eps, order, norm)
        ro, rd = orc.TV(u, M, N, eps, order, norm)    # pinned bit for bit to the compiled reference (tests/golden/tv.npz)
        assert np.array_equal(div, rd)
        if norm == 1:
            assert np.array_equal(out, ro)
        else:   # norm 2 ends in libm's powf(s, 0.5) (within 0.82 ulp) where the device rounds sqrt correctly
            np.testing.assert_allclose(out, ro, rtol=4e-7, atol=0)   # sum of two such terms, then a division
        assert np.all(out[0] == 0) and np.all(out[-1] == 0) and np.all(div[:, 0] == 0) and np.all(div[:, -1] == 0)
    flat = np.full((9, 9, 3), 0.5, np.float32)
    out, div = ctx.tv(flat, 1e-3, 2, 1)
    assert np.all(div == 0)   # TV gradient of a constant image vanishes


@pytest.mark.gpu
def test_gpu_tv_against_the_reference_golden(ctx, golden_dir):
    """the device stencil against the COMPILED REFERENCE's TV directly (tests/golden/tv.npz)"""
    import json
    import os
    z = np.load(os.path.join(golden_dir, "tv.npz"))
    for name, eps, order, norm in json.loads(str(z["meta"]))["cases"]:
        key = "%s_e%g_o%d_n%d" % (name, eps, order, norm)
        out, div = ctx.tv(z["u_" + name], eps, order, norm)
        assert np.array_equal(div, z["div_" + key]), key
        if norm == 1:
            assert np.array_equal(out, z["out_" + key]), key
        else:
            np.testing.assert_allclose(out, z["out_" + key], rtol=4e-7, atol=0)


def test_oracle_tv_equals_the_compiled_reference(golden_dir):
    """A4: oracle/rl_mm_oracle.TV is pinned bit for bit (all four order/norm variants, two epsilons, four inputs)"""
    import json
    import os
    import rl_mm_oracle as orc
    z = np.load(os.path.join(golden_dir, "tv.npz"))
    cases = json.loads(str(z["meta"]))["cases"]
    assert len(cases) == 32
    for name, eps, order, norm in cases:
        u = z["u_" + name]
        out, div = orc.TV(u, u.shape[0], u.shape[1], eps, order, norm)
        key = "%s_e%g_o%d_n%d" % (name, eps, order, norm)
        assert np.array_equal(out, z["out_" + key]) and np.array_equal(div, z["div_" + key]), key


@pytest.mark.gpu
def test_gpu_blur_usm_4096_channel_against_scipy(ctx):
    """U2/U3 at the size of a BASELINE frame: one 4096 x 4096 float64 channel, 15 x 15 Gaussian window (rank 1: row + column
    pass) and a non-separable 9 x 7 kernel (the general 2-D tile path), against scipy.signal.convolve2d itself -- the call
    the reference makes (lib/utils.py:243-262)."""
    from scipy.signal import convolve2d
    from lib import utils
    rng = np.random.default_rng(4)
    src = rng.random((4096, 4096))
    kern = utils.gaussian_kernel(15, 2.5)
    ref = convolve2d(src, kern, mode="same", boundary="symm")
    out = utils.gaussian_blur(src, 15, 2.5)
    t_sep = ctx.last_kernel_ms()
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(utils.USM(src, 15, 2.5, 0.8, method="gauss"), src + (src - ref) * 0.8, rtol=1e-12, atol=1e-13)
    k2 = rng.random((9, 7)); k2 /= k2.sum()
    out2 = ctx.conv2d_symm(src[:1500, :1300], k2)
    t_2d = ctx.last_kernel_ms()
    np.testing.assert_allclose(out2, convolve2d(src[:1500, :1300], k2, mode="same", boundary="symm"), rtol=1e-12, atol=1e-14)
    print("gaussian_blur 4096^2, 15x15 (separable): %.3f ms on the device; general 9x7 at 1500x1300: %.3f ms" % (t_sep, t_2d))
    assert t_sep < 20.0


@pytest.mark.gpu
def test_gpu_bilateral_1024_against_the_restatement(ctx):
    from lib import utils
    rng = np.random.default_rng(6)
    src = rng.random((1024, 1000))
    out = utils.bilateral_filter(src, 4, 0.15, 2.0)
    t = ctx.last_kernel_ms()
    np.testing.assert_allclose(out, uo.bilateral_filter(src, 4, 0.15, 2.0), rtol=1e-11)
    print("bilateral 1024x1000, radius 4: %.3f ms on the device" % t)
