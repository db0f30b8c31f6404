"""Bicubic resize between pyramid levels (deconvolve.py:245-249).  skimage is an un-vendored
dependency of the reference and absent here: parity with it is UNPINNED.  Pinned instead: the written-out algorithm
(oracle/resize_oracle.py: resize_explicit) against the scipy.ndimage calls it restates (CPU), and the HIP
implementation against both (GPU), at the shapes the pyramid produces."""
import numpy as np
import pytest

import resize_oracle as ro

SHAPES = [((41, 57), (29, 41)), ((29, 41), (41, 57)), ((33, 33), (47, 47)), ((15, 15), (11, 11)), ((7, 7), (5, 5)),
          ((3, 3), (3, 3)), ((101, 77), (71, 55)), ((64, 65), (65, 64)), ((5, 5), (15, 15))]


@pytest.mark.parametrize("src,dst", SHAPES)
def test_written_out_algorithm_equals_scipy(src, dst):
    rng = np.random.default_rng(1)
    img = rng.random((*src, 3))
    a, b = ro.resize_scipy(img, dst), ro.resize_explicit(img, dst)
    assert a.shape == (*dst, 3)
    assert np.abs(a - b).max() < 1e-13


def test_properties_of_the_resize():
    c = np.full((21, 33, 3), 0.37)
    assert np.abs(ro.resize_explicit(c, (15, 23)) - 0.37).max() < 1e-14          # constants are preserved
    yy, xx = np.mgrid[0:100, 0:120].astype(float)
    ramp = np.dstack((yy, xx, yy + xx))
    up = ro.resize_explicit(ramp, (200, 240))                                     # cubic splines reproduce linear ramps
    ys = ro.sample_grid(100, 200)[:, None]                                         # (away from the edge-replicated border:
    xs = ro.sample_grid(120, 240)[None, :]                                         #  its influence decays like 0.268^k)
    inner = (slice(50, -50), slice(50, -50))
    assert np.abs(up[..., 0] - np.broadcast_to(ys, (200, 240)))[inner].max() < 1e-10
    assert np.abs(up[..., 1] - np.broadcast_to(xs, (200, 240)))[inner].max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", SHAPES + [((513, 771), (363, 545)), ((363, 545), (513, 771))])
def test_gpu_resize_matches_the_oracle(src, dst):
    from lib import _native
    rng = np.random.default_rng(2)
    img = rng.random((*src, 3))
    got = _native.Context.get().resize_bicubic(img, dst)
    ref = ro.resize_scipy(img, dst) if src != dst else img
    assert got.shape == (*dst, 3) and got.dtype == np.float64
    assert np.abs(got - ref).max() < 1e-12, np.abs(got - ref).max()


@pytest.mark.gpu
def test_gpu_resize_single_channel_and_psf_shapes():
    from lib import _native
    ctx = _native.Context.get()
    rng = np.random.default_rng(3)
    a = rng.random((31, 31))
    assert np.abs(ctx.resize_bicubic(a, (21, 21)) - ro.resize_scipy(a[..., None], (21, 21))[..., 0]).max() < 1e-12
    psf = np.full((15, 15, 3), 1 / 225.0)
    assert np.abs(ctx.resize_bicubic(psf, (11, 11)) - 1 / 225.0).max() < 1e-15


@pytest.mark.gpu
def test_driver_resize_goes_through_the_device(monkeypatch):
    import deconvolve as dv
    rng = np.random.default_rng(4)
    img = rng.random((45, 61, 3)).astype(np.float32)
    out = dv.resize_bicubic(img, (33, 43, 3))
    assert np.abs(out - ro.resize_scipy(img, (33, 43))).max() < 1e-12
    assert np.array_equal(dv.resize_bicubic(img, img.shape), img.astype(np.float64))


@pytest.mark.gpu
def test_gpu_resize_extreme_ratios_and_many_calls():
    """large up- and down-scaling factors (wide anti-aliasing kernels), 2 x 2 sources, and a create / resize / destroy
    loop on device images (same result every time, nothing left behind)"""
    from lib import _native
    ctx = _native.Context.get()
    rng = np.random.default_rng(7)
    for src, dst in [((2, 2), (9, 7)), ((120, 90), (13, 11)), ((13, 11), (120, 90)), ((64, 3), (31, 5)), ((300, 300), (7, 299))]:
        img = rng.random((*src, 3))
        got = ctx.resize_bicubic(img, dst)
        assert np.abs(got - ro.resize_scipy(img, dst)).max() < 1e-11, (src, dst)
    a = rng.random((256, 256, 3), dtype=np.float32)
    first = None
    for i in range(60):
        d = _native.DeviceImage.from_host(a)
        r = d.resize(181, 181).pad_edge(1, 1, 1, 1)
        out = r.to_host()
        if first is None:
            first = out
        assert np.array_equal(out, first)
        d.close(); r.close()
