"""TEST INFRASTRUCTURE (checker for the GPU resize, never imported by the product path).

Bicubic resize between pyramid levels, deconvolve.py:245-249 of the reference:
    skimage.transform.resize(img, shape, order=3, mode="edge", preserve_range=True)
skimage is an un-vendored dependency of the reference and is not installed here -> PARITY UNPINNED.  What is
restated is skimage's documented behaviour on top of scipy.ndimage (which IS installed and is the checker):
Gaussian anti-aliasing with sigma = (scale - 1) / 2 when shrinking, then a cubic B-spline interpolation at the
pixel-centre grid, edge mode "nearest".

  resize_scipy(img, shape)     -- the scipy.ndimage calls themselves (gaussian_filter + map_coordinates)
  resize_explicit(img, shape)  -- the same algorithm written out (what csrc/ics_resize.hip implements):
      1. separable Gaussian, radius int(4 sigma + 0.5), weights exp(-x^2 / 2 sigma^2) / sum, edge-replicated
      2. edge padding by 12 samples, then the cubic B-spline prefilter (pole z = sqrt(3) - 2, gain 6) along
         each axis with the mirror initialisation of Unser et al. (scipy ni_splines.c: nearest -> pad + mirror)
      3. out[i, j] = sum_{a,b<4} w_a(y) w_b(x) coef[floor(y) - 1 + a, floor(x) - 1 + b] at
         y = (i + 0.5) H / OH - 0.5 (+12), cubic B-spline weights
"""
import numpy as np

NPAD = 12
POLE = np.sqrt(3.0) - 2.0


def sample_grid(n_in, n_out):
    return (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5


def aa_sigma(n_in, n_out):
    return max(0.0, (n_in / n_out - 1.0) / 2.0)


def resize_scipy(img, shape):
    from scipy import ndimage
    img = np.asarray(img, dtype=np.float64)
    out_h, out_w = int(shape[0]), int(shape[1])
    in_h, in_w = img.shape[0], img.shape[1]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()
    sy, sx = aa_sigma(in_h, out_h), aa_sigma(in_w, out_w)
    if sy > 0 or sx > 0:
        img = ndimage.gaussian_filter(img, (sy, sx, 0), mode="nearest")
    yy, xx = np.meshgrid(sample_grid(in_h, out_h), sample_grid(in_w, out_w), indexing="ij")
    out = np.empty((out_h, out_w, img.shape[2]))
    for c in range(img.shape[2]):
        out[..., c] = ndimage.map_coordinates(img[..., c], [yy, xx], order=3, mode="nearest")
    return out


def gauss_weights(sigma):
    radius = int(4.0 * sigma + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 / (sigma * sigma) * x * x)
    return w / w.sum(), radius


def gauss_axis(a, sigma, axis):
    if sigma <= 1e-15:
        return a
    w, r = gauss_weights(sigma)
    n = a.shape[axis]
    out = np.zeros_like(a)
    for k in range(-r, r + 1):
        idx = np.clip(np.arange(n) + k, 0, n - 1)
        out += w[k + r] * np.take(a, idx, axis=axis)
    return out


def prefilter_axis(a, axis):
    """cubic B-spline coefficients along `axis`, mirror initialisation (exact sums)."""
    a = np.moveaxis(a, axis, 0).copy()
    n = a.shape[0]
    z = POLE
    a *= (1.0 - z) * (1.0 - 1.0 / z)
    zn = z ** (n - 1)
    c0 = a[0] + zn * a[n - 1]
    zi = z
    for i in range(1, n - 1):
        c0 = c0 + zi * (a[i] + zn * a[n - 1 - i])
        zi *= z
    a[0] = c0 / (1.0 - zn * zn)
    for i in range(1, n):
        a[i] += z * a[i - 1]
    a[n - 1] = (z * a[n - 2] + a[n - 1]) * z / (z * z - 1.0)
    for i in range(n - 2, -1, -1):
        a[i] = z * (a[i + 1] - a[i])
    return np.moveaxis(a, 0, axis)


def bspline3_weights(t):
    """weights of samples floor(x)-1 .. floor(x)+2 for fractional part t"""
    return np.stack(((1 - t) ** 3 / 6.0, (3 * t ** 3 - 6 * t ** 2 + 4) / 6.0, (-3 * t ** 3 + 3 * t ** 2 + 3 * t + 1) / 6.0, t ** 3 / 6.0))


def resize_explicit(img, shape):
    img = np.asarray(img, dtype=np.float64)
    out_h, out_w = int(shape[0]), int(shape[1])
    in_h, in_w = img.shape[0], img.shape[1]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()
    img = gauss_axis(img, aa_sigma(in_h, out_h), 0)
    img = gauss_axis(img, aa_sigma(in_w, out_w), 1)
    coef = np.pad(img, ((NPAD, NPAD), (NPAD, NPAD), (0, 0)), mode="edge")
    coef = prefilter_axis(coef, 0)
    coef = prefilter_axis(coef, 1)
    ys = sample_grid(in_h, out_h) + NPAD
    xs = sample_grid(in_w, out_w) + NPAD
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    wy = bspline3_weights(ys - y0)   # [4, OH]
    wx = bspline3_weights(xs - x0)   # [4, OW]
    out = np.zeros((out_h, out_w, img.shape[2]))
    for a in range(4):
        rows = coef[y0 - 1 + a]                          # [OH, Wp, C]
        for b in range(4):
            out += (wy[a][:, None, None] * wx[b][None, :, None]) * rows[:, x0 - 1 + b]
    return out
