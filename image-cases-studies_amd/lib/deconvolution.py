"""Drop-in for the reference's compiled module `lib.deconvolution` (lib/deconvolution.pyx).

Same public names, argument order and side effects as the reference:

    richardson_lucy_MM(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations,
                       step_factor, lambd, blind=True, correlation=False, p=1., norm=1, order=2,
                       priority=0, refocus=0)            -> lib/deconvolution.pyx:341-342
    normalize_kernel(kern, MK)                          -> lib/deconvolution.pyx:73-75
    DTYPE = numpy.float32                                -> lib/deconvolution.pyx:31

but the loop runs on an MI355X through libics_hip.so (hand-written gfx950 kernels, C ABI in
include/ics_hip.h, bound with ctypes in lib/_native.py).  There is no CPU path in this module.

Behaviour kept from the reference:
  * `image`, `u`, `psf` must be float32 ndarrays with ndim 3 (any strides); wrong dtype/ndim raise the
    same ValueError texts Cython's buffer check produces.
  * `u` is updated IN PLACE and the return value is a VIEW of the caller's `u` (pyx:675).
  * `psf` is updated in place when `blind`; under `correlation=True` the caller's array only
    receives the first gradient step because pyx:585 rebinds the local name.
  * `p, norm, order, priority, refocus` are accepted and ignored; `iterations` counts OUTER
    iterations of 5 inner ones (pyx:375).
  * The reference's progress lines are printed as the run proceeds (one callback per outer iteration), any number of
    outer iterations.
Extras: `richardson_lucy_MM.last` holds the `RLStats` of the most recent call.
"""
from __future__ import annotations

import numpy as np

from . import _native

DTYPE = np.float32

_job_cache = {}


def _check_buffer(name, arr):
    if not isinstance(arr, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)" % (name, type(arr).__name__))
    if arr.ndim != 3:
        raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % arr.ndim)
    if arr.dtype != np.float32:
        cname = {"float64": "double", "int64": "long", "int32": "int", "uint8": "unsigned char",
                 "float16": "short"}.get(arr.dtype.name, arr.dtype.name)
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % cname)


# deblur_module calls the solver at 2 x (pyramid levels) problem sizes per picture -- 10 for a 15-px blur -- and batch runs repeat them per
# picture: with 4 cached jobs every call built a new one (24 buffers, a pinned mirror, events; ~1 ms).  Least recently used out, by count
# and by device bytes (a job holds 7-8 frames of its size).
_JOB_CACHE_MAX = 16
_JOB_CACHE_MAX_BYTES = 32 << 30      # (upper bound; the effective limit is a quarter of the device's memory, _cache_limit)


def _cache_limit():
    """bytes of device frames the cache may hold: a quarter of the device's memory (ics_ctx_info), at most _JOB_CACHE_MAX_BYTES --
    a card smaller than an MI355X, or two processes on one, must not find the cache in the way of an allocation"""
    try:
        return min(_JOB_CACHE_MAX_BYTES, _native.Context.get(_native.default_device()).hbm_bytes // 4)
    except Exception:
        return _JOB_CACHE_MAX_BYTES


_FRAME_LIMIT = _native.FRAME_LIMIT_BYTES     # (a test lowers it to send small frames through the same routing)


def _bands_needed(M, N, MK, limit=None):
    """1 when an M x N job's frames stay below the library's 2 GiB frame limit, else the smallest number of row bands whose jobs (a band's
    rows plus MK // 2 halo rows either side) do; ValueError when no split helps (a single row of pixels too wide, bands thinner than the halo)."""
    limit = _FRAME_LIMIT if limit is None else limit
    if _native.frame_bytes(M, N, MK) < limit:
        return 1
    pad = MK // 2
    for k in range(2, M // max(2 * pad, 1) + 1):
        rows = -(-M // k) + 2 * pad                # the tallest band with its halos
        if _native.frame_bytes(min(rows, M), N, MK) < limit:
            return k
    raise ValueError("a %d x %d image with a %d x %d PSF cannot be cut into row bands below the %d-byte frame limit" % (M, N, MK, MK, limit))


def _get_job(M, N, MK):
    """Keep the device frames of the most recent problem sizes alive (least recently used out): deblur_module calls the
    solver once or twice per pyramid level, and the reference's per-call allocation of 13 scratch frames (pyx:378-390) is
    what this replaces."""
    key = (int(M), int(N), int(MK), _native.default_device())
    job = _job_cache.pop(key, None)
    if job is None:
        need, limit = _job_bytes(key), _cache_limit()
        while _job_cache and (len(_job_cache) >= _JOB_CACHE_MAX or need + sum(_job_bytes(k) for k in _job_cache) > limit):
            _job_cache.pop(next(iter(_job_cache))).close()
        try:
            job = _native.RLJob(M, N, MK)
        except _native.NativeError as exc:
            if exc.code != _native.ICS_ENOMEM or not _job_cache:
                raise
            _drop_jobs()                       # out of device memory with cached jobs alive: release them and try once more
            job = _native.RLJob(M, N, MK)
    _job_cache[key] = job            # (re)inserted last = most recently used
    return job


def _job_bytes(key):
    """device bytes a job of this shape can come to hold: 6 frames at creation, the ping-pong residual frame and the image's
    accumulator-order copies of the overlapped / matrix-core runs (3 more), the channel-planar mirrors of the FFT-tile pipeline
    (7, and the TV frame's for the PAM kinds; any PSF size on a large frame since round 6) with the image
    windows' spectra of its mode 2 (1.7 frames, PSF sizes up to 25), the scratch frames of the tap-block path above 49"""
    M, N, MK = key[:3]
    frames = 9 + 8 + (2 if MK <= 25 else 0) + (2 if MK >= 51 else 0)
    return frames * (M + 2 * MK + 128) * (N + 2 * MK + 128) * 12


def _drop_jobs():
    """Release every cached job (device frames)."""
    while _job_cache:
        _job_cache.pop(next(iter(_job_cache))).close()


def normalize_kernel(kern, MK):
    """lib/deconvolution.pyx:73-75 -- clamp negatives to 0 and normalise every channel to sum 1, in place."""
    _check_buffer("kern", kern)
    MK = int(MK)
    work = np.ascontiguousarray(kern[:MK, :MK, :3], dtype=np.float32)
    _native.Context.get().normalize_kernel(work, MK)
    kern[:MK, :MK, :3] = work


def _progress(it, stopped, dof_min, dof_max, M_r, Hu, varu):
    """the reference's per-outer-iteration lines (pyx:593,648,658-659), printed live from ics_rl_run's callback"""
    print("DoF : min = %f | max = %f" % (dof_min, dof_max))
    if stopped:
        print("white autocorellation condition met")
    if it % 50 == 0:
        print("%i iterations completed" % it)


def _run_interruptible(job, params):
    """job.run with the reference's behaviour under Ctrl-C: the reference updates the caller's `u` / `psf` in place as it goes, so
    a KeyboardInterrupt leaves the partial result in them and deconvolve.py:338-342 keeps it.  Here the frames live on the device:
    an exception raised while a progress line is printed stops the device loop after that outer iteration (the callback's abort
    channel, include/ics_hip.h ics_rl_progress_fn), the caller writes the partial `u` / `psf` back as usual and re-raises.
    Returns (stats, exception or None)."""
    try:
        return job.run(params, progress=_progress), None
    except BaseException as e:      # noqa: BLE001
        st = getattr(e, "ics_stats", None)
        if st is None:
            raise
        return st, e


def _report(st, top, bottom, left, right, lambd):
    """the reference's closing lines (pyx:661-672)"""
    if st.stopped:
        print("Convergence after %i iterations." % st.iterations_done)
    else:
        print("Did not converge after %i iterations. Don't use the result." % st.iterations_done)
    print("Stats : autocovariance = %.6f | lamdba = %.0f | residual = %.6f | variance/noise = %.6f" % (
        1000 * st.M_r / ((bottom - top) * (right - left) * 3), np.float32(lambd), st.Hu, st.varu))
    if st.has_nan:
        print("has NaN after DoF correction")


class DeviceWindow:
    """What `richardson_lucy_MM_device` returns: the counterpart of the reference's returned view u[pad:pad+M, pad:pad+N]
    (pyx:675) for a device-resident frame -- the image and the rectangle, nothing is copied until `.to_host()` / `.crop()`."""

    def __init__(self, image, y0, x0, H, W):
        self.image, self.y0, self.x0, self.H, self.W = image, int(y0), int(x0), int(H), int(W)
        self.shape = (self.H, self.W, 3)

    def crop(self):
        return self.image.crop(self.y0, self.y0 + self.H, self.x0, self.x0 + self.W)

    def to_host(self):
        c = self.crop()
        try:
            return c.to_host()
        finally:
            c.close()


def richardson_lucy_MM_device(image, image_origin, u, u_origin, psf, top, bottom, left, right, tau, M, N, C, MK, iterations,
                              step_factor, lambd, blind=True, correlation=False, p=1., norm=1, order=2, priority=0, refocus=0, *,
                              tv_mode=0, conv=0, flags=0):
    """`richardson_lucy_MM` on device-resident frames (not in the reference): `image` and `u` are
    `_native.DeviceImage`s, the solver works on the windows image[iy:iy+M, ix:ix+N] and u[uy:uy+uM, ux:ux+uN]
    (`image_origin = (iy, ix)`, `u_origin = (uy, ux)`) -- the views deconvolve.py:277-313 passes -- and the whole `u`
    window is updated in place like the reference's `u` argument.  `psf` stays a host array, updated in place when
    `blind`.  Nothing crosses PCIe but the PSF and the statistics."""
    _check_buffer("psf", psf)
    M, N, MK = int(M), int(N), int(MK)
    if tv_mode == 1:
        raise ValueError("tv_mode 1 modifies `image`; use richardson_lucy_MM for it")
    job = _get_job(M, N, MK)
    job.upload_img(image, image_origin, u, u_origin, psf)
    params = job.params(top, bottom, left, right, tau, iterations, step_factor, lambd, blind, correlation, channels=C,
                        tv_mode=tv_mode, conv=conv, flags=flags)
    st, interrupt = _run_interruptible(job, params)
    job.download_img(u, u_origin)
    if blind:
        psf[...] = job.download_psf_caller()
    richardson_lucy_MM.last = st
    if interrupt is not None:
        raise interrupt
    _report(st, top, bottom, left, right, lambd)
    richardson_lucy_MM.last = st
    pad = MK // 2
    return DeviceWindow(u, u_origin[0] + pad, u_origin[1] + pad, M, N)                       # pyx:675


def richardson_lucy_MM(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations, step_factor, lambd,
                       blind=True, correlation=False, p=1., norm=1, order=2, priority=0, refocus=0, *, tv_mode=0, conv=0, flags=0):
    """Richardson-Lucy blind / non-blind deconvolution by majorisation-minimisation
    (lib/deconvolution.pyx:341-675), executed on the GPU.  See the module docstring.

    `tv_mode` (keyword-only, not in the reference): 0 = the shipped behaviour (TV term dead); 1 = the
    build-defined active MM-TV mode (include/ics_hip.h ICS_TV_MM_ACTIVE, oracle/rl_ext_oracle.py; parity
    unpinned), in which `image` is also updated in place as pyx:549 intends.

    `conv` (keyword-only, not in the reference): include/ics_hip.h ICS_CONV_*: 0 = auto (matrix-core kernels with
    fp16-split operands at every PSF size -- whole PSF to 49 x 49, tap blocks above; `_native.RLJob.describe(params)` reports what a run
    will use), 1 = fp32 products everywhere, 2 = force the matrix-core kernels.

    `flags` (keyword-only, not in the reference): include/ics_hip.h ICS_FLAG_* bits (1 = run A11 and A13 as two kernels
    instead of the fused one)."""
    _check_buffer("image", image)
    _check_buffer("u", u)
    _check_buffer("psf", psf)
    M, N, MK = int(M), int(N), int(MK)
    u_M, u_N = u.shape[0], u.shape[1]
    pad = (u_M - M) // 2                                                       # pyx:376
    if u.shape != (M + 2 * (MK // 2), N + 2 * (MK // 2), 3) or image.shape != (M, N, 3) or psf.shape != (MK, MK, 3):
        # the reference does not validate shapes (it would read out of bounds); the GPU path cannot do that
        raise ValueError("expected image (%d,%d,3), u (%d,%d,3), psf (%d,%d,3); got %s, %s, %s" %
                         (M, N, M + 2 * (MK // 2), N + 2 * (MK // 2), MK, MK, image.shape, u.shape, psf.shape))
    nbands = _bands_needed(M, N, MK)
    if nbands > 1:
        # a frame of 2 GiB and more (about 13 000 x 13 000 px): the kernels address a frame with 32-bit offsets.  The reference has no such
        # limit (pyx:341), so the call goes through the row bands of lib/banded.py -- all of them on this GPU -- instead of failing.
        if tv_mode != 0 or flags != 0:
            raise ValueError("a %d x %d image needs row bands (frames of 2 GiB and more), which are built for the shipped loop (tv_mode 0, flags 0)" % (M, N))
        from .banded import richardson_lucy_MM_banded
        _drop_jobs()
        out = richardson_lucy_MM_banded(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations, step_factor, lambd, blind, correlation,
                                        bands=nbands, devices=[_native.default_device()] * nbands, conv=conv)
        richardson_lucy_MM.last = richardson_lucy_MM_banded.last
        return out
    job = _get_job(M, N, MK)
    job.upload(image, u, psf)
    params = job.params(top, bottom, left, right, tau, iterations, step_factor, lambd, blind, correlation, channels=C,
                        tv_mode=tv_mode, conv=conv, flags=flags)
    st, interrupt = _run_interruptible(job, params)
    u_new, _psf_local, psf_caller = job.download(u)                            # straight into the caller's u when it is contiguous ...
    if u_new is not u:
        u[...] = u_new                                                         # ... else in place through a copy, any strides
    if blind:
        psf[...] = psf_caller
    if tv_mode == 1:
        image[...] = job.read(_native.BUF_IMAGE)                               # pyx:549 (live in this mode only; the PAM kinds never write it)
    richardson_lucy_MM.last = st
    if interrupt is not None:
        raise interrupt
    _report(st, top, bottom, left, right, lambd)
    richardson_lucy_MM.last = st
    return u[pad:pad + M, pad:pad + N, ...]                                    # pyx:675 (view)


richardson_lucy_MM.last = None
