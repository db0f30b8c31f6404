"""Drop-in for the parts of the reference's `lib/utils.py` that sit on the deconvolution path.

Same names and argument meaning as the reference (file:line in /root/reference/lib/utils.py):

    timeit                                   :30-42
    disc_blur, lens_blur                     :134-143
    uniform_kernel / gaussian_kernel /
    kaiser_kernel / poisson_kernel           :146-170   (host side: tiny separable windows)
    bilateral_filter(source, radius, std_i, std_s, parallel=1)      :194-234   -> HIP (ics_bilateral)
    bessel_blur(src, radius, amount)         :237-249   -> HIP (ics_conv2d_symm)
    gaussian_blur(src, radius, amount)       :252-264   -> HIP (ics_conv2d_symm)
    USM(src, radius, strength, amount, method="bessel")             :267-277   -> HIP (ics_usm)
    save(pic, name, dest_path)               :303-312   (16-bit RGB TIFF)
    convolve(a, b, domain)                   :420-447   (FFT convolution; unused by the reference)

The filters run on the GPU through libics_hip.so in float64 like the reference (scipy's
convolve2d(mode="same", boundary="symm")); there is no CPU fallback for them.  The colour tools of
the reference (Lagrange_interpolation, grey_point, auto_vibrance, overlay, blending) and its dead
code (divTV, gradTVEM) are outside the deconvolution path and are not provided.

Notes on reference quirks that are kept:
  * `gaussian_kernel(radius, std)`: `radius` is the window SIZE (scipy.signal.gaussian(radius, std)).
  * `bilateral_filter` as shipped raises NameError (`gaussian` is undefined, :186-187); the intended
    gaussian(x, s) = exp(-x^2 / (2 s^2)) is used (any normalisation cancels in filtered / W).
"""
from __future__ import annotations

import os
import struct
import time
from os.path import join

import numpy as np

from . import _native


def timeit(method):
    """lib/utils.py:30-42"""
    def timed(*args, **kw):
        ts = time.time()
        result = method(*args, **kw)
        te = time.time()
        print('%r %2.2f sec' % (method.__name__, te - ts))
        return result
    return timed


# ---- PSF window builders (host) ------------------------------------------------------------------
def disc_blur(x):
    half = [1 / (np.pi * x ** 2) for x in range(1, int(x / 2) + 1)]
    return half


def lens_blur(size):
    window = disc_blur(size)
    kern = np.outer(window, window)
    kern = kern / kern.sum()
    return kern


def uniform_kernel(size):
    kern = np.ones((size, size))
    kern /= np.sum(kern)
    return kern


def _gaussian_window(M, std):
    """scipy.signal.windows.gaussian(M, std, sym=True)"""
    if M < 1:
        return np.array([])
    if M == 1:
        return np.ones(1)
    n = np.arange(0, M) - (M - 1.0) / 2.0
    return np.exp(-n ** 2 / (2 * std * std))


def _exponential_window(M, tau):
    """scipy.signal.windows.exponential(M, center=None, tau=tau, sym=True)"""
    if M < 1:
        return np.array([])
    if M == 1:
        return np.ones(1)
    center = (M - 1) / 2
    n = np.arange(0, M)
    return np.exp(-np.abs(n - center) / tau)


def gaussian_kernel(radius, std):
    window = _gaussian_window(radius, std)
    kern = np.outer(window, window)
    kern = kern / kern.sum()
    return kern


def kaiser_kernel(radius, beta):
    window = np.kaiser(radius, beta)
    kern = np.outer(window, window)
    kern = kern / kern.sum()
    return kern


def poisson_kernel(radius, tau):
    window = _exponential_window(radius, tau)
    kern = np.outer(window, window)
    kern = kern / kern.sum()
    return kern


# ---- filters (GPU) -------------------------------------------------------------------------------
def _as2d(src):
    src = np.asarray(src)
    if src.ndim != 2:
        raise ValueError("expected a 2-D channel, got shape %s" % (src.shape,))
    return np.ascontiguousarray(src, dtype=np.float64)


def bilateral_filter(source, radius, std_i, std_s, parallel=1):
    """lib/utils.py:194-234: symmetric padding by `radius`, all (2r+1)^2 offsets,
    w = gaussian(neighbour - source, std_i) * gaussian(distance, std_s), result = sum(neighbour*w)/sum(w)."""
    return _native.Context.get().bilateral(_as2d(source), int(radius), float(std_i), float(std_s))


def bessel_blur(src, radius, amount):
    """lib/utils.py:237-249: convolve2d(src, kaiser_kernel(radius, amount), mode="same", boundary="symm")"""
    return _native.Context.get().conv2d_symm(_as2d(src), kaiser_kernel(radius, amount))


def gaussian_blur(src, radius, amount):
    """lib/utils.py:252-264: convolve2d(src, gaussian_kernel(radius, amount), mode="same", boundary="symm")"""
    return _native.Context.get().conv2d_symm(_as2d(src), gaussian_kernel(radius, amount))


def USM(src, radius, strength, amount, method="bessel"):
    """lib/utils.py:267-277: src + (src - blur(src, radius, strength)) * amount, fused on the device."""
    kern = {"bessel": kaiser_kernel, "gauss": gaussian_kernel}[method](radius, strength)
    return _native.Context.get().usm(_as2d(src), kern, float(amount))


# ---- I/O ------------------------------------------------------------------------------------------
def _write_tiff_rgb16(path, arr):
    """Minimal baseline TIFF writer: uncompressed, little-endian, 16 bits x 3 samples, chunky RGB.
    (The reference vendors tifffile for this, lib/utils.py:312; one strip is all `save` needs.)"""
    h, w, c = arr.shape
    assert c == 3 and arr.dtype == np.uint16
    data = np.ascontiguousarray(arr).astype("<u2").tobytes()
    n_entries = 10
    ifd_off = 8
    bps_off = ifd_off + 2 + n_entries * 12 + 4
    data_off = bps_off + 6
    def ent(tag, typ, count, value):
        return struct.pack("<HHII", tag, typ, count, value)
    ifd = struct.pack("<H", n_entries)
    ifd += ent(256, 4, 1, w) + ent(257, 4, 1, h) + ent(258, 3, 3, bps_off) + ent(259, 3, 1, 1)
    ifd += ent(262, 3, 1, 2) + ent(273, 4, 1, data_off) + ent(277, 3, 1, 3) + ent(278, 4, 1, h)
    ifd += ent(279, 4, 1, len(data)) + ent(284, 3, 1, 1)
    ifd += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_off))
        f.write(ifd)
        f.write(struct.pack("<HHH", 16, 16, 16))
        f.write(data)


def save(pic, name, dest_path):
    """lib/utils.py:303-312: 16-bit RGB TIFF `<dest_path>/<name>.tif`."""
    _write_tiff_rgb16(join(dest_path, name + ".tif"), np.asarray(pic).astype(np.uint16))


def convolve(a, b, domain):
    """lib/utils.py:420-447 (pyFFTW in the reference, unused by it): FFT convolution of two 2-D arrays,
    `domain` in {"same", "valid", "full"}; output = the first (Y, X) samples of the full result like the
    reference's irfft2(c_temp, (Y, X))."""
    MK, NK = b.shape[0], b.shape[1]
    M, N = a.shape[0], a.shape[1]
    if domain == "same":
        Y, X = M, N
    elif domain == "valid":
        Y, X = M - MK + 1, N - NK + 1
    elif domain == "full":
        Y, X = M + MK - 1, N + NK - 1
    else:
        raise SyntaxError
    s = (M + MK - 1, N + NK - 1)
    c_temp = np.fft.rfft2(a, s=s) * np.fft.rfft2(b, s=s)
    return np.fft.irfft2(c_temp, (Y, X))
