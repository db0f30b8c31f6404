# -*- coding: utf-8 -*-
"""Driver with the call surface of the reference's `deconvolve.py` (pad_image :24-37, build_pyramid
:40-60, deblur_module :65-368), feeding the GPU solver in `lib.deconvolution`.

What is kept from the reference: argument names/defaults of `deblur_module`, the validation errors
(:117-120, :145-148), gamma and bit-depth scaling (:97-103, :346-352), odd-size padding (:164-175), the
pyramid schedule, the mask-window arithmetic (:209-230) and the three call shapes into
`richardson_lucy_MM` (:277-313) including their positional arguments.

What differs, and why:
  * `skimage.transform.resize(order=3, mode="edge")` (:245-249) is an un-vendored dependency that is
    not installed in this image; `resize_bicubic` below runs its documented algorithm (Gaussian anti-aliasing
    when shrinking + cubic B-spline interpolation) on the GPU (csrc/ics_resize.hip), checked against
    scipy.ndimage in oracle/resize_oracle.py.  PARITY with skimage itself is UNPINNED for the pyramid levels
    below 1.0; with `pyramid=False` (one level, scale 1) no resize is involved.
  * matplotlib pop-ups (:331-336) only when `display=True` and matplotlib is importable.
  * the 16-bit TIFF is written by `lib.utils.save` (own minimal writer instead of the vendored tifffile).
"""
import numpy as np

from lib import utils
from lib import deconvolution as dc


def pad_image(image, pad, mode="edge"):
    """deconvolve.py:24-37 -- grow the two spatial axes of an H x W x 3 picture; `pad` is whatever numpy.pad takes for a 2-D array
    (a width, a (before, after) pair for both axes, or one pair per axis); the channel axis is never padded.  float32, C-contiguous."""
    widths = np.broadcast_to(np.asarray(pad, dtype=int), (2, 2))
    grown = np.pad(np.asarray(image), (tuple(widths[0]), tuple(widths[1]), (0, 0)), mode=mode)
    return np.ascontiguousarray(grown, dtype=np.float32)


def build_pyramid(psf_size, lambd):
    """deconvolve.py:40-60 -- the coarse-to-fine schedule: every level shrinks the picture by sqrt(2) and the PSF with it, the PSF size
    rounded up and then made odd (never below 3); the last level is the first whose PSF is 3 x 3.  Returns (scales, sizes), finest first.
    (`lambd` is accepted and unused, as in the reference.)"""
    root2 = np.sqrt(2)
    scales, sizes = [1.], [psf_size]
    while sizes[-1] > 3:
        k = int(np.ceil(sizes[-1] / root2))
        k -= 1 - k % 2                      # even -> the odd size below
        sizes.append(max(k, 3))
        scales.append(scales[-1] / root2)
    return scales, sizes


def resize_bicubic(img, shape):
    """skimage.transform.resize(img, shape, order=3, mode="edge", preserve_range=True) (deconvolve.py:245-249) on the GPU:
    Gaussian anti-aliasing when shrinking + cubic B-spline interpolation, `ics_resize_bicubic` (csrc/ics_resize.hip)."""
    img = np.asarray(img, dtype=np.float64)
    if img.shape[:2] == (int(shape[0]), int(shape[1])):   # scale 1: the interpolating spline reproduces the samples
        return img.copy()
    from lib import _native
    return _native.Context.get().resize_bicubic(img, shape)


def mask_window(i, top, bottom, left, right):
    """deconvolve.py:209-230 -- scale the mask box to pyramid level `i` and make it odd and square,
    with the reference's exact (and slightly odd) tie-breaking."""
    temp_top, temp_bottom = int(i * top), int(i * bottom)
    temp_left, temp_right = int(i * left), int(i * right)
    if int(temp_bottom - temp_top) % 2 == 0:
        if int(temp_bottom - temp_top) < int(temp_right - temp_left):
            temp_bottom += 1
        elif int(temp_bottom - temp_top) > int(temp_right - temp_left):
            temp_top += 1
        else:
            temp_top -= 1
    if int(temp_right - temp_left) % 2 == 0:
        if int(temp_bottom - temp_top) < int(temp_right - temp_left):
            temp_left += 1
        elif int(temp_bottom - temp_top) > int(temp_bottom - temp_top):   # (sic) never true, deconvolve.py:227
            temp_right += 1
        else:
            temp_right -= -1                                              # (sic) deconvolve.py:230
    return temp_top, temp_bottom, temp_left, temp_right


def _check_mask_size(mask_size):
    """The stop-test statistics of the GPU solver (residual autocorrelation, lib/deconvolution.pyx:623-638) run a P x P FFT with
    P <= 2048: windows up to 1024 px.  The reference has no such limit (its examples use 255 and 811, deconvolve.py:67,416);
    fail here with a clear message rather than from inside the first solver call."""
    if 2 * (2 * (int(mask_size) // 2)) - 1 > 2048:
        raise ValueError("mask_size = %d: the GPU path evaluates the stop test on windows of at most 1024 x 1024 px (mask_size <= 1025)" % mask_size)


@utils.timeit
def deblur_module(pic, filename, dest_path, blur_width, confidence=10, tolerance=1, quality="normal", bits=8,
                  mask=None, display=True, blur="static", preview=False, p=1, order=2, norm=1, priority=0, mask_size=255,
                  iterations=200, refocus=False, pyramid=True, solver=None, save=True, device_resident=None):
    """deconvolve.py:65-368.  Extra keyword arguments (not in the reference): `pyramid=False` runs the
    single scale-1 level only, `solver` replaces `dc.richardson_lucy_MM` (tests record the calls),
    `save=False` returns the float image instead of writing the TIFF, `device_resident=True` keeps every frame in HBM
    from the first upload to the final download (`_deblur_device`; same arithmetic, same calls into the solver; the two
    paths differ by float32 `powf` of the gamma steps, numpy vs device: <= 2e-5 of the 16-bit range, tests/test_driver.py).
    Default (None): resident unless a `solver` is given or `display` asks for the matplotlib pop-up of the host frames --
    2048^2, 15-px blur, 20 iterations: 0.085 s resident, 0.6-0.9 s with the frames on the host between the solver calls."""
    if device_resident is None:
        device_resident = solver is None and not display
    if device_resident:
        if solver is not None:
            raise ValueError("device_resident=True runs the GPU solver; `solver` cannot be replaced")
        return _deblur_device(pic, filename, dest_path, blur_width, confidence, tolerance, quality, bits, mask, display, blur, preview, p,
                              order, norm, priority, mask_size, iterations, refocus, pyramid, save)
    rl = solver if solver is not None else dc.richardson_lucy_MM
    pic = np.ascontiguousarray(pic, dtype=np.float32)
    pic = pad_image(pic, (1, 1)).astype(np.float32)                       # :94
    samples = 2 ** bits - 1                                               # :97
    pic = pic / samples
    pic = pic ** (1 / 2.2)                                                # :103
    step = {"normal": 1e-3, "high": 5e-4, "veryhigh": 1e-4, "low": 5e-3}[quality]   # :106-113
    if blur_width < 3:
        raise ValueError("The blur width should be at least 3 pixels.")
    elif blur_width % 2 == 0:
        raise ValueError("The blur width should be odd. You can use %i." % (blur_width + 1))
    M, N = pic.shape[0], pic.shape[1]
    if mask is None:
        mask = [M // 2, N // 2]
    _check_mask_size(mask_size)
    top, bottom = mask[0] - mask_size // 2, mask[0] + mask_size // 2       # :138-141
    left, right = mask[1] - mask_size // 2, mask[1] + mask_size // 2
    print("Mask size :", (bottom - top + 1), "×", (right - left + 1))
    if not (top > 0 and bottom < M and left > 0 and right < N):
        raise ValueError("The mask is outside the picture boundaries. Move its center inside or reduce the blur size.")
    correlation = {"static": False, "motion": True}[blur]                 # :154-157
    tolerance /= 100.
    odd_vert = odd_hor = False
    if pic.shape[0] % 2 == 0:                                             # :167-175
        pic = pad_image(pic, ((1, 0), (0, 0))).astype(np.float32)
        odd_vert = True
        print("Padded vertically")
    if pic.shape[1] % 2 == 0:
        pic = pad_image(pic, ((0, 0), (1, 0))).astype(np.float32)
        odd_hor = True
        print("Padded horizontally")
    M, N = pic.shape[0], pic.shape[1]
    psf = utils.uniform_kernel(blur_width)                                # :178-179
    psf = np.dstack((psf, psf, psf))
    images, kernels = build_pyramid(blur_width, confidence)               # :182
    if not pyramid:
        images, kernels = images[:1], kernels[:1]
    deblured_image = pic
    try:
        for case in ["blind", "non-blind"]:                               # :193
            print("\n===== %s DECONVOLUTION =====" % case)
            deblured_image = pic.copy()
            lambd = confidence * 1000                                     # :200
            for i, k in zip(reversed(images), reversed(kernels)):        # :204
                print("======== Pyramid step %1.3f ========" % i)
                temp_top, temp_bottom, temp_left, temp_right = mask_window(i, top, bottom, left, right)
                temp_width, temp_height = int(np.floor(i * N)), int(np.floor(i * M))
                if temp_width % 2 == 0:
                    temp_width += 1
                if temp_height % 2 == 0:
                    temp_height += 1
                shape = (temp_height, temp_width, 3)
                temp_blurry_image = resize_bicubic(pic, shape).astype(np.float32)              # :245
                deblured_image = resize_bicubic(deblured_image, shape).astype(np.float32)      # :246
                if case == "blind":
                    psf_copy = np.ascontiguousarray(resize_bicubic(psf, (k, k, 3)).astype(np.float32))   # :249
                    dc.normalize_kernel(psf_copy, k)
                else:
                    psf_copy = np.ascontiguousarray(psf, dtype=np.float32).copy()
                    k = kernels[0]
                temp_blurry_image = pad_image(temp_blurry_image, (1, 1)).astype(np.float32)    # :256-257
                deblured_image = pad_image(deblured_image, (1, 1)).astype(np.float32)
                pad = int(np.floor(k / 2))
                print("Image size", temp_blurry_image.shape)
                print("u size", deblured_image.shape)
                print("Mask size", (temp_bottom - temp_top), (temp_right - temp_left))
                print("PSF size", psf_copy.shape)
                tolerance_temp = tolerance if i == 1. else 0
                win = (pad + 1, temp_bottom - temp_top - pad - 1, pad + 1, temp_bottom - temp_top - pad - 1)
                if case == "blind":                                       # :277-288
                    deblured_image[temp_top - 1:temp_bottom + 1, temp_left - 1:temp_right + 1, ...] = rl(
                        temp_blurry_image[temp_top - 1:temp_bottom + 1, temp_left - 1:temp_right + 1, ...],
                        deblured_image[temp_top - pad - 1:temp_bottom + pad + 1, temp_left - pad - 1:temp_right + pad + 1, ...],
                        psf_copy, *win, 0, temp_bottom - temp_top + 2, temp_right - temp_left + 2, 3,
                        k, iterations, step, lambd, blind=True, p=p, correlation=correlation, order=order, norm=2,
                        priority=0, refocus=refocus)
                    psf = psf_copy.copy()
                elif preview:                                             # :290-300
                    deblured_image[temp_top - 1:temp_bottom + 1, temp_left - 1:temp_right + 1, ...] = rl(
                        temp_blurry_image[temp_top - 1:temp_bottom + 1, temp_left - 1:temp_right + 1, ...],
                        deblured_image[temp_top - pad - 1:temp_bottom + pad + 1, temp_left - pad - 1:temp_right + pad + 1, ...],
                        psf_copy, *win, tolerance_temp, temp_bottom - temp_top + 2, temp_right - temp_left + 2, 3,
                        k, iterations, step, lambd, blind=False, p=p, order=order, norm=2, priority=priority, refocus=refocus)
                else:                                                     # :301-316
                    deblured_image = pad_image(deblured_image, (pad, pad)).astype(np.float32)
                    deblured_image[pad:-pad, pad:-pad, ...] = rl(
                        temp_blurry_image, deblured_image, psf_copy, *win, tolerance_temp,
                        temp_height + 2, temp_width + 2, 3,
                        k, iterations, step, lambd, blind=False, p=p, order=order, norm=2, priority=priority, refocus=refocus)
                    deblured_image = deblured_image[pad:-pad, pad:-pad, ...]
                temp_blurry_image = temp_blurry_image[1:-1, 1:-1, ...]    # :322-323
                deblured_image = deblured_image[1:-1, 1:-1, ...]
            if display and case == "blind":                               # :331-336
                try:
                    import matplotlib.pyplot as plt
                    psf_check = (psf - np.amin(psf)) / (np.amax(psf) - np.amin(psf))
                    plt.imshow(psf_check, interpolation="lanczos", aspect="equal", vmin=0, vmax=1)
                    plt.show()
                except ImportError:
                    pass
    except KeyboardInterrupt:                                             # :338-342
        pass
    deblured_image = np.clip(deblured_image, 0., 1.)                      # :346
    deblured_image = deblured_image ** 2.2
    deblured_image = deblured_image * (2 ** 16 - 1)
    if preview:
        filename = filename + "-preview"
        deblured_image = deblured_image[top:bottom, left:right, ...]
    else:
        if odd_hor:
            deblured_image = deblured_image[:, 1:, ...]
        if odd_vert:
            deblured_image = deblured_image[1:, :, ...]
        deblured_image = deblured_image[1:-1, 1:-1, ...]
    if save:
        utils.save(deblured_image, filename, dest_path)
    return deblured_image, psf


def _level_shape(i, M, N):
    """deconvolve.py:232-243 -- odd size of pyramid level `i`"""
    temp_width, temp_height = int(np.floor(i * N)), int(np.floor(i * M))
    if temp_width % 2 == 0:
        temp_width += 1
    if temp_height % 2 == 0:
        temp_height += 1
    return temp_height, temp_width


def _deblur_device(pic, filename, dest_path, blur_width, confidence, tolerance, quality, bits, mask, display, blur, preview, p, order, norm,
                   priority, mask_size, iterations, refocus, pyramid, save):
    """`deblur_module` (deconvolve.py:65-368) with every frame resident in HBM: one upload of the picture,
    one download of the result; pad_image, gamma, the window views, the resize between pyramid levels and the solver all
    work on `lib._native.DeviceImage`s.  Line references as in `deblur_module` above."""
    from lib._native import DeviceImage
    if blur_width < 3:
        raise ValueError("The blur width should be at least 3 pixels.")
    elif blur_width % 2 == 0:
        raise ValueError("The blur width should be odd. You can use %i." % (blur_width + 1))
    raw = DeviceImage.from_host(pic)              # (uint8 / uint16 pictures cross PCIe as they are and become float32 on the device)
    pic_d = raw.pad_edge(1, 1, 1, 1)                                        # :94
    raw.close()
    pic_d.gamma(2 ** bits - 1, 1 / 2.2)                                     # :97-103
    step = {"normal": 1e-3, "high": 5e-4, "veryhigh": 1e-4, "low": 5e-3}[quality]
    M, N, _ = pic_d.shape
    if mask is None:
        mask = [M // 2, N // 2]
    _check_mask_size(mask_size)
    top, bottom = mask[0] - mask_size // 2, mask[0] + mask_size // 2
    left, right = mask[1] - mask_size // 2, mask[1] + mask_size // 2
    print("Mask size :", (bottom - top + 1), "×", (right - left + 1))
    if not (top > 0 and bottom < M and left > 0 and right < N):
        raise ValueError("The mask is outside the picture boundaries. Move its center inside or reduce the blur size.")
    correlation = {"static": False, "motion": True}[blur]
    tolerance /= 100.
    odd_vert = odd_hor = False
    if M % 2 == 0:                                                          # :167-175
        pic_d, old = pic_d.pad_edge(1, 0, 0, 0), pic_d
        old.close()
        odd_vert = True
        print("Padded vertically")
    if N % 2 == 0:
        pic_d, old = pic_d.pad_edge(0, 0, 1, 0), pic_d
        old.close()
        odd_hor = True
        print("Padded horizontally")
    M, N, _ = pic_d.shape
    psf = utils.uniform_kernel(blur_width)
    psf = np.dstack((psf, psf, psf))
    images, kernels = build_pyramid(blur_width, confidence)
    if not pyramid:
        images, kernels = images[:1], kernels[:1]
    deb = pic_d.copy()
    import time
    from lib import _native
    phases = deblur_module.last_phase_seconds = {}      # wall time per phase of this call, device drained at the phase boundaries (bench.py reports it)
    try:
        for case in ["blind", "non-blind"]:
            _native.Context.get().synchronize()
            t_case = time.perf_counter()
            print("\n===== %s DECONVOLUTION =====" % case)
            deb.close()
            deb = pic_d.copy()
            lambd = confidence * 1000
            for i, k in zip(reversed(images), reversed(kernels)):
                print("======== Pyramid step %1.3f ========" % i)
                tt, tb, tl, tr = mask_window(i, top, bottom, left, right)
                th, tw = _level_shape(i, M, N)
                blurry = pic_d.resize(th, tw)                                # :245
                deb, old = deb.resize(th, tw), deb                           # :246
                old.close()
                if case == "blind":
                    psf_copy = np.ascontiguousarray(resize_bicubic(psf, (k, k, 3)).astype(np.float32))   # :249
                    dc.normalize_kernel(psf_copy, k)
                else:
                    psf_copy = np.ascontiguousarray(psf, dtype=np.float32).copy()
                    k = kernels[0]
                blurry, old = blurry.pad_edge(1, 1, 1, 1), blurry            # :256-257
                old.close()
                deb, old = deb.pad_edge(1, 1, 1, 1), deb
                old.close()
                pad = int(np.floor(k / 2))
                print("Image size", blurry.shape)
                print("u size", deb.shape)
                print("Mask size", (tb - tt), (tr - tl))
                print("PSF size", psf_copy.shape)
                tolerance_temp = tolerance if i == 1. else 0
                win = (pad + 1, tb - tt - pad - 1, pad + 1, tb - tt - pad - 1)
                if case == "blind" or preview:                               # :277-300: windows of both frames
                    dc.richardson_lucy_MM_device(
                        blurry, (tt - 1, tl - 1), deb, (tt - pad - 1, tl - pad - 1), psf_copy, *win,
                        0 if case == "blind" else tolerance_temp, tb - tt + 2, tr - tl + 2, 3, k, iterations, step, lambd,
                        blind=(case == "blind"), p=p, correlation=(correlation if case == "blind" else False), order=order, norm=2,
                        priority=(0 if case == "blind" else priority), refocus=refocus)
                    if case == "blind":
                        psf = psf_copy.copy()
                else:                                                        # :301-316: the whole frame
                    big = deb.pad_edge(pad, pad, pad, pad)
                    dc.richardson_lucy_MM_device(blurry, (0, 0), big, (0, 0), psf_copy, *win, tolerance_temp, th + 2, tw + 2, 3, k,
                                                 iterations, step, lambd, blind=False, p=p, order=order, norm=2, priority=priority,
                                                 refocus=refocus)
                    deb.close()
                    H2, W2, _ = big.shape
                    deb = big.crop(pad, H2 - pad, pad, W2 - pad)
                    big.close()
                blurry.close()
                H1, W1, _ = deb.shape
                deb, old = deb.crop(1, H1 - 1, 1, W1 - 1), deb               # :322-323
                old.close()
            _native.Context.get().synchronize()
            phases[case] = time.perf_counter() - t_case
    except KeyboardInterrupt:
        pass
    deb.gamma(1.0, 2.2, 2 ** 16 - 1, clip01=True)                           # :346-352
    out = deb.to_host()
    deb.close()
    pic_d.close()
    if preview:
        filename = filename + "-preview"
        out = out[top:bottom, left:right, ...]
    else:
        if odd_hor:
            out = out[:, 1:, ...]
        if odd_vert:
            out = out[1:, :, ...]
        out = out[1:-1, 1:-1, ...]
    if save:
        utils.save(out, filename, dest_path)
    return out, psf


deblur_module.last_phase_seconds = {}


if __name__ == '__main__':
    import sys
    from PIL import Image
    if len(sys.argv) < 4:
        raise SystemExit("usage: python deconvolve.py <picture> <dest_dir> <blur_width> [mask_y mask_x]")
    with Image.open(sys.argv[1]) as pic:
        mask = [int(sys.argv[4]), int(sys.argv[5])] if len(sys.argv) >= 6 else None
        deblur_module(pic, sys.argv[1].rsplit("/", 1)[-1] + "-ics", sys.argv[2], int(sys.argv[3]), mask=mask, display=False)
