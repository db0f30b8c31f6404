"""CPU oracle: numpy restatement of the reference's Richardson-Lucy / MM deconvolution loop.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is imported by the product path
(`image-cases-studies_amd/`): only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may use it, and only as the checker.

What is restated (reference = /root/reference, file:line):
  * `richardson_lucy_MM`            lib/deconvolution.pyx:341-675
  * `normalize_kernel`              lib/deconvolution.pyx:47-75
  * `rotate_180`                    lib/deconvolution.pyx:242-252
  * `TV` (order 1/2, norm 1/2)      lib/deconvolution.pyx:137-239   (dead in the shipped loop, see below)
  * `gaussian_weight`               lib/deconvolution.pyx:35-36

Semantics that are easy to get wrong, all reproduced here:
  1. The TV regulariser is arithmetically dead: TV_ut_L1/L2 stay zero (pyx:386-387, writers
     commented out at :464-465) so the else-branches at :519 and :545 always run.
  2. `1/(u_M*u_N)` (pyx:524,548,574) is integer division == 0 (Cython 0.28 / language_level 2).
  3. Convolutions are `scipy.signal.convolve(..., method="auto")` (pyx:13, call sites :478,:491,
     :558,:571,:632).  scipy is a third-party dependency of the reference that is neither vendored
     nor version-pinned by it (no requirements file, README.md:184-192); parity is pinned against
     scipy 1.15.3 / numpy 2.2.6, the versions in this image.  `conv="scipy"` calls the very same
     entry point (bit-exact against the compiled reference in this container, see
     oracle/check_oracle_vs_reference.py); `conv="direct"` evaluates the same sums directly in
     float64 (the "noise floor" trajectory of SURVEY.md section 8c).
  4. Scalar arithmetic follows numpy-2 promotion (NEP 50): Python floats are weak, so
     `step_factor * np.amax(u[..., k])` etc. are float32 operations (pyx:524,574).
  5. `correlation=True` rebinds the local `psf` (pyx:585): the caller's array only ever receives
     the first gradient step (pyx:577-581), un-normalised.

Parity status: PINNED for `richardson_lucy_MM` / `normalize_kernel` by goldens generated from the
compiled reference (tests/golden/rl_*.npz, generator oracle/make_golden.py).  The extended modes
(`tv_mode` != "shipped") have no reference implementation: PARITY UNPINNED for those.
"""
from __future__ import annotations

import io
import math
from dataclasses import dataclass, field

import numpy as np

DTYPE = np.float32  # pyx:31
INNER_ITER = 5      # pyx:375

F32 = np.float32


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def normalize_kernel(kern: np.ndarray, MK: int) -> None:
    """pyx:47-75 -- clamp negatives to 0, divide each channel by its (sequential float32) sum."""
    assert kern.dtype == np.float32 and kern.ndim == 3
    temp = [F32(0.0), F32(0.0), F32(0.0)]
    for k in range(3):
        for i in range(MK):
            for j in range(MK):
                if kern[i, j, k] < 0:
                    kern[i, j, k] = 0
                temp[k] = F32(temp[k] + kern[i, j, k])
    with np.errstate(divide="ignore", invalid="ignore"):
        for k in range(3):
            kern[:MK, :MK, k] /= temp[k]


def rotate_180(a: np.ndarray) -> np.ndarray:
    """pyx:242-252 -- out[i, N-1-j, k] = in[M-1-i, j, k]."""
    return np.ascontiguousarray(a[::-1, ::-1, :])


def gaussian_weight(source, target, sigma):
    """pyx:35-36, float32 libm arithmetic (expf/powf)."""
    source = np.asarray(source, dtype=np.float32)
    PI = F32(3.141592653589793)
    num = -np.power(source - F32(target), F32(2)) / (F32(2) * np.power(F32(sigma), F32(2)))
    return (np.exp(num.astype(np.float32)) / (F32(sigma) * np.power(F32(2) * PI, F32(0.5)))).astype(np.float32)


def stop_weights(top, bottom, left, right):
    """pyx:393-404 -- Gaussian window for the residual-whiteness metric."""
    width = np.linspace(-1.0, 1.0, num=(bottom - top), dtype=DTYPE)
    height = np.linspace(-1.0, 1.0, num=(right - left), dtype=DTYPE)
    width = gaussian_weight(width, 0.0, 1.0)
    height = gaussian_weight(height, 0.0, 1.0)
    weights = np.sqrt(np.outer(width, height))
    weights /= np.sum(weights)
    return weights.astype(np.float32)


def _conv_scipy(a, b, mode):
    from scipy.signal import convolve  # pyx:13
    return convolve(a, b, mode=mode)


def _conv_direct(a, b, mode):
    """Direct float64 evaluation of scipy.signal.convolve(a, b, mode) for 2-D inputs."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if mode == "valid" and (b.shape[0] > a.shape[0]):
        a, b = b, a
    Ma, Na = a.shape
    Mb, Nb = b.shape
    if mode == "valid":
        out = np.zeros((Ma - Mb + 1, Na - Nb + 1))
        # out[i,j] = sum_{p,q} b[p,q] a[i+Mb-1-p, j+Nb-1-q]   (SURVEY.md 8a "exact index forms", A1)
        if Mb * Nb <= out.size:
            for p in range(Mb):
                for q in range(Nb):
                    out += b[p, q] * a[Mb - 1 - p:Mb - 1 - p + out.shape[0], Nb - 1 - q:Nb - 1 - q + out.shape[1]]
        else:  # few outputs, long sums (A13)
            br = b[::-1, ::-1]
            for i in range(out.shape[0]):
                for j in range(out.shape[1]):
                    out[i, j] = np.sum(a[i:i + Mb, j:j + Nb] * br)
        return out
    full = np.zeros((Ma + Mb - 1, Na + Nb - 1))
    if Mb * Nb <= Ma * Na:
        for p in range(Mb):
            for q in range(Nb):
                full[p:p + Ma, q:q + Na] += b[p, q] * a
    else:
        for p in range(Ma):
            for q in range(Na):
                full[p:p + Mb, q:q + Nb] += a[p, q] * b
    if mode == "full":
        return full
    if mode == "same":  # centred on `a`
        oy, ox = (Mb - 1) // 2, (Nb - 1) // 2
        return full[oy:oy + Ma, ox:ox + Na]
    raise ValueError(mode)


_libm_powf = None


def _powf(x, y):
    """C `powf` of the host libm, element by element (the reference calls it for every square and square root, pyx:129-131;
    glibc's powf is within 0.82 ulp, i.e. not always the correctly rounded value numpy's float32 power returns).  Arrays
    beyond 2e5 elements fall back to the correctly rounded float64 evaluation (<= 1 ulp from libm)."""
    global _libm_powf
    x = np.asarray(x, dtype=np.float32)
    if x.size > 200000:
        return np.power(x.astype(np.float64), float(y)).astype(np.float32)
    if _libm_powf is None:
        import ctypes
        import ctypes.util
        lib = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        lib.powf.restype = ctypes.c_float
        lib.powf.argtypes = [ctypes.c_float, ctypes.c_float]
        _libm_powf = np.frompyfunc(lambda a, b: lib.powf(a, b), 2, 1)
    return np.asarray(_libm_powf(x, np.float32(y)), dtype=np.float32)


def TV(u, M, N, epsilon, order, norm):
    """pyx:137-239.  Returns (out, div); borders untouched (zero), as `:239` says.

    Pinned against the compiled reference (oracle/make_golden_tv.py, tests/golden/tv.npz), which settled two points the
    source does not show: Cython writes the integer literal of `-2 * u[i, j, k]` as the C double `-2.0`, so the
    second-order stencil `-2*u + a + b` is evaluated in DOUBLE and rounded once (lib/deconvolution.c:4176); and
    `adjust = 4. * (1 + 1/dxdy)` is likewise a double expression stored into a C float (:4031).  The first-order
    differences `u - a`, `-u + b` involve no literal and stay float.  gcc folds `powf(x, 2)` into `x * x` (exactly rounded),
    `powf(s, 0.5)` stays a libm call."""
    u = np.asarray(u, dtype=np.float32)
    out = np.zeros_like(u)
    div = np.zeros_like(u)
    dxdy = F32(_powf(np.array([2.0], np.float32), 0.5)[0])                       # powf(2.0, 0.5)
    if norm == 1:
        adjust = F32(4.0 * (1.0 + 1.0 / np.float64(dxdy)))
    else:
        adjust = F32(2.0 * (1.0 + np.float64(dxdy)))
    eps = F32(epsilon)
    c = u[1:M - 1, 1:N - 1]

    def n1(x, y):                                                                 # fabsf(x) + fabsf(y) + epsilon
        return np.abs(x) + np.abs(y) + eps

    def n2(x, y):                                                                 # powf(powf(x,2) + powf(y,2) + powf(eps,2), 0.5)
        return _powf((x * x + y * y) + eps * eps, 0.5)

    nrm = n1 if norm == 1 else n2
    up, dn = u[0:M - 2, 1:N - 1], u[2:M, 1:N - 1]
    lf, rt = u[1:M - 1, 0:N - 2], u[1:M - 1, 2:N]
    ul, dr = u[0:M - 2, 0:N - 2], u[2:M, 2:N]
    ur, dl = u[0:M - 2, 2:N], u[2:M, 0:N - 2]
    if order == 2:
        c64 = c.astype(np.float64)
        udx = ((-2.0 * c64 + up) + dn).astype(np.float32)
        udy = ((-2.0 * c64 + lf) + rt).astype(np.float32)
        # (the numerator of the diagonal terms lands in a float temporary -- Cython's division-by-zero check -- before the
        #  float division by dxdy: double sum, rounded, then divided in float)
        udxdy = ((-2.0 * c64 + ul) + dr).astype(np.float32) / dxdy
        udydx = ((-2.0 * c64 + ur) + dl).astype(np.float32) / dxdy
        d = (-udx - udy - udxdy - udydx) / adjust
        o = (nrm(udx, udy) + nrm(udxdy, udydx)) / adjust
    else:
        udx_b, udy_b = c - up, c - lf
        udx_f, udy_f = -c + dn, -c + rt
        udxdy_b, udydx_b = (c - ul) / dxdy, (c - ur) / dxdy
        udydx_f, udxdy_f = (-c + dl) / dxdy, (-c + dr) / dxdy
        d = (udx_b + udy_b - udx_f - udy_f + udxdy_b + udydx_b - udxdy_f - udydx_f) / adjust
        o = (nrm(udx_b, udy_b) + nrm(udx_f, udy_f) + nrm(udxdy_b, udydx_b) + nrm(udxdy_f, udydx_f)) / adjust
    out[1:M - 1, 1:N - 1] = o
    div[1:M - 1, 1:N - 1] = d
    return out, div


# --------------------------------------------------------------------------------------------
# trace object (the reference only prints; tests want numbers)
# --------------------------------------------------------------------------------------------
@dataclass
class Trace:
    """Per-outer-iteration scalars the reference prints but never returns."""
    M_r: list = field(default_factory=list)
    Hu: list = field(default_factory=list)
    varu: list = field(default_factory=list)
    dof_min: list = field(default_factory=list)
    dof_max: list = field(default_factory=list)
    dt: list = field(default_factory=list)       # per inner iteration, 3 floats
    dtpsf: list = field(default_factory=list)    # per inner iteration (blind)
    iterations: int = 0
    stopped: bool = False
    psf_final: np.ndarray | None = None          # the *local* psf (differs from caller's under correlation)
    snapshots: dict = field(default_factory=dict)  # outer-iteration -> copy of u (full, padded)
    log: io.StringIO = field(default_factory=io.StringIO)


def residual_whiteness(err_win, weights, conv):
    """pyx:627-638 -- returns M_r (float32)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        test = (err_win - np.mean(err_win)) / np.std(err_win)
        test = test / np.amax(np.abs(test))
        test = np.ascontiguousarray(test, dtype=np.float32)
        for k in range(3):
            ac = conv(test[..., k], np.rot90(test[..., k], 2), "same")
            test[..., k] = ac
            test[..., k] = test[..., k] ** 2 * weights
        return F32(np.mean(test))


# --------------------------------------------------------------------------------------------
# the loop
# --------------------------------------------------------------------------------------------
def dof_ratio(g, f, zero_rule=1.0):
    """(g - f)/(g + f) of pyx:499 in float32, IEEE except where g == f == 0 exactly -> `zero_rule` (NaN = plain IEEE).
    See richardson_lucy_MM's docstring for why the value is 1 wherever the convolutions are exact."""
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (g - f) / (g + f)
    if zero_rule == zero_rule:
        r = np.where((g == 0) & (f == 0), F32(zero_rule), r).astype(np.float32)
    return r


def richardson_lucy_MM(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations,
                       step_factor, lambd, blind=True, correlation=False, p=1., norm=1, order=2,
                       priority=0, refocus=0, *, conv="scipy", trace: Trace | None = None,
                       snapshot_at=(), quiet=False, tv_mode="shipped", dof_zero_rule=None):
    """Restatement of lib/deconvolution.pyx:341-675.  Mutates `u` (always) and `psf` (blind) in
    place and returns a view of `u`, exactly like the reference.

    Extra keyword-only arguments (not in the reference): `conv` ("scipy" | "direct"), `trace`
    (collects the printed scalars), `snapshot_at` (outer-iteration counts at which to copy u),
    `quiet` (suppress prints), `tv_mode` ("shipped" = TV term dead, as the reference behaves),
    `dof_zero_rule`: value of the ratio (g - f)/(g + f) of pyx:499 where g == f == 0 EXACTLY.  None -> IEEE (0/0 = NaN)
    with conv="scipy" -- the reference bit for bit, whatever its FFT's rounding makes of such a pixel -- and 1 with
    conv="direct".  Why 1: in a region where image and u are exactly 0 the reference's g is the rounding noise of a complex64
    FFT (~1e-10, either sign), f is 0, and (g - 0)/(g + 0) = 1 for EVERY non-zero g; exact arithmetic (this branch, and the
    device kernels) produces g = 0 there and IEEE would turn the whole frame into NaN through the next convolution.  The
    rule touches nothing else: g + f == 0 with g != 0 stays +-inf as in the reference (include/ics_hip.h, "DoF ratio").
    """
    for name, arr in (("image", image), ("u", u), ("psf", psf)):
        if not isinstance(arr, np.ndarray):
            raise TypeError("Argument '%s' has incorrect type" % name)
        if arr.ndim != 3:
            raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % arr.ndim)
        if arr.dtype != np.float32:
            raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" %
                             {"float64": "double"}.get(arr.dtype.name, arr.dtype.name))
    if tv_mode != "shipped":
        raise NotImplementedError("extended TV modes live in oracle/rl_ext_oracle.py")
    cv = _conv_scipy if conv == "scipy" else _conv_direct
    if dof_zero_rule is None:
        dof_zero_rule = 1.0 if conv == "direct" else float("nan")
    tr = trace if trace is not None else Trace()

    def say(s):
        tr.log.write(s + "\n")
        if not quiet:
            print(s)

    step_factor = F32(step_factor)
    lambd = F32(lambd)
    tau = F32(tau)
    u_M, u_N = u.shape[0], u.shape[1]
    pad = (u_M - M) // 2                                                  # pyx:376
    gradk = np.zeros((MK, MK, 3), dtype=DTYPE)
    ut = np.zeros((u_M, u_N, 3), dtype=DTYPE)
    gradu = np.zeros((u_M, u_N, 3), dtype=DTYPE)
    synth = np.zeros((M, N, 3), dtype=DTYPE)
    error = np.zeros((M, N, 3), dtype=DTYPE)
    DoF = np.zeros((M, N, 3), dtype=DTYPE)
    weights = stop_weights(top, bottom, left, right)                      # pyx:393-404
    psf_rotated = rotate_180(psf)                                         # pyx:441
    caller_psf = psf
    it = 0
    stop_flag = False
    M_r = M_r_prev = F32(0)
    Hu = varu = F32(0)
    interior = (slice(pad, u_M - pad), slice(pad, u_N - pad))            # [pad:-pad]

    while it < iterations and not stop_flag:                              # pyx:460
        ut[:] = u                                                         # pyx:462
        for _itt in range(INNER_ITER):                                    # pyx:473
            for ch in range(3):                                           # pyx:477-478  (A1)
                synth[..., ch] = cv(u[..., ch], psf[..., ch], "valid")
            error[:] = synth - image                                      # pyx:482-488  (A2)
            for k in range(3):                                            # pyx:490-491  (A3)
                gradu[..., k] = cv(error[..., k], psf_rotated[..., k], "full")
            # pyx:495-496 TV(u) x2: outputs unused in the shipped code (A4) -> skipped
            with np.errstate(divide="ignore", invalid="ignore"):
                gi = gradu[interior]
                DoF = dof_ratio(gi, image, dof_zero_rule) ** 2            # pyx:499       (A5)
                if not blind:
                    DoF = DoF / lambd                                     # pyx:501-502
            # pyx:512-519 else-branch (A6): float product + double (u-ut)/2., stored as float
            a = (lambd * gradu).astype(np.float64)
            b = (u - ut).astype(np.float64) / 2.0
            gradu[:] = (a + b).astype(np.float32)
            dt = np.zeros(3, dtype=np.float32)
            for k in range(3):                                            # pyx:523-524  (A7)
                dt[k] = F32(step_factor * F32(np.amax(u[..., k]) + 0)) / F32(np.amax(np.abs(gradu[..., k])) + F32(1e-15))
            tr.dt.append(dt.copy())
            for k in range(3):                                            # pyx:527-531  (A8)
                u[..., k] -= dt[k] * gradu[..., k]
            # pyx:534-549 (A9): gradu = 0; image -= dt*0/lambd  -> image unchanged
            gradu[:] = 0
            u[interior] = (F32(1.0) - DoF) * u[interior] + DoF * image    # pyx:552       (A10)
            if blind and not stop_flag:                                   # pyx:555
                for ch in range(C):                                       # pyx:557-558  (A11)
                    error[..., ch] = cv(u[..., ch], psf[..., ch], "valid")
                error -= image                                            # pyx:561-565
                u_rot = rotate_180(u)                                     # pyx:567       (A12)
                for ch in range(C):                                       # pyx:570-571  (A13)
                    gradk[..., ch] = cv(u_rot[..., ch], error[..., ch], "valid")
                # pyx:574 (A14): step_factor / MK is a C float division
                dtpsf = F32(F32(step_factor / F32(MK)) * F32(np.amax(psf) + 0)) / F32(np.amax(np.abs(gradk)) + F32(1e-15))
                tr.dtpsf.append(dtpsf)
                psf -= dtpsf * gradk                                      # pyx:577-581
                if correlation:                                           # pyx:584-585  (A15, rebinding quirk)
                    m = np.mean(psf, axis=2)
                    psf = np.dstack((m, m, m))
                normalize_kernel(psf, MK)                                 # pyx:587       (A16)
                psf_rotated = rotate_180(psf)                             # pyx:589       (A17)
        with np.errstate(invalid="ignore"):
            say("DoF : min = %f | max = %f" % (np.amin(DoF), np.amax(DoF)))   # pyx:593
            tr.dof_min.append(F32(np.amin(DoF)))
            tr.dof_max.append(F32(np.amax(DoF)))
            varu = F32(np.std(u[top + pad:bottom - pad, left + pad:right - pad, ...]) ** 2)      # pyx:600
            Hu = F32(np.linalg.norm(error[top:bottom, left:right, ...]) ** 2 / ((bottom - top) * (right - left) * 3))  # pyx:601
        if it > 0:
            M_r_prev = M_r                                                # pyx:623-624
        M_r = residual_whiteness(error[top:bottom, left:right, ...], weights, cv)  # pyx:627-638
        tr.M_r.append(M_r)
        tr.Hu.append(Hu)
        tr.varu.append(varu)
        if it > 1:                                                        # pyx:643-654
            if blind:
                if M_r > M_r_prev:
                    stop_flag = True
                    say("white autocorellation condition met")
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    if F32(M_r - M_r_prev) / F32(M_r + M_r_prev) > tau:
                        stop_flag = True
                        say("white autocorellation condition met")
        it += 1
        if it in snapshot_at:
            tr.snapshots[it] = (u.copy(), psf.copy())
        if it % 50 == 0:
            say("%i iterations completed" % it)

    if stop_flag:
        say("Convergence after %i iterations." % it)
    else:
        say("Did not converge after %i iterations. Don't use the result." % it)
    say("Stats : autocovariance = %.6f | lamdba = %.0f | residual = %.6f | variance/noise = %.6f" % (
        1000 * M_r / ((bottom - top) * (right - left) * 3), lambd, Hu, varu))
    if np.any(np.isnan(u)):
        say("has NaN after DoF correction")
    tr.iterations = it
    tr.stopped = bool(stop_flag)
    tr.psf_final = psf.copy()
    del caller_psf
    return u[pad:pad + M, pad:pad + N, ...]                               # pyx:675


# --------------------------------------------------------------------------------------------
# synthetic inputs -- shared by tests, goldens and bench
# --------------------------------------------------------------------------------------------
def gaussian_psf(MK, sigma=None):
    """utils.gaussian_kernel(MK, MK/6) (lib/utils.py:152-156) stacked on 3 channels, float32."""
    sigma = MK / 6.0 if sigma is None else sigma
    n = np.arange(MK) - (MK - 1) / 2.0
    w = np.exp(-0.5 * (n / sigma) ** 2)
    k = np.outer(w, w)
    k = k / k.sum()
    return np.ascontiguousarray(np.dstack((k, k, k)), dtype=np.float32)


def uniform_psf(MK):
    """utils.uniform_kernel (lib/utils.py:146-149) stacked on 3 channels (deconvolve.py:178-179)."""
    k = np.ones((MK, MK)) / (MK * MK)
    return np.ascontiguousarray(np.dstack((k, k, k)), dtype=np.float32)


def _smooth7(x):
    """7x7 Gaussian sigma=1.5, separable, 'same' with edge replication (per channel)."""
    n = np.arange(7) - 3.0
    w = np.exp(-0.5 * (n / 1.5) ** 2)
    w /= w.sum()
    xp = np.pad(x, ((3, 3), (3, 3), (0, 0)), mode="edge")
    t = sum(w[i] * xp[i:i + x.shape[0]] for i in range(7))
    t = sum(w[i] * t[:, i:i + x.shape[1]] for i in range(7))
    return t


def synth_case(M, N, MK, seed=0, blind=False, noise=1e-3, per_channel_psf=False):
    """Seeded synthetic deconvolution problem.  Returns dict(image, u0, psf0, psf_true, pad).

    sharp = smoothed uniform noise *0.8+0.1 on the padded frame; image = valid conv with the true
    Gaussian PSF + N(0, noise); u0 = image edge-padded by pad (mirrors deconvolve.py:303);
    psf0 = true PSF (non-blind) or uniform (blind, deconvolve.py:178).
    """
    rng = np.random.default_rng(seed)
    pad = MK // 2
    sharp = rng.random((M + 2 * pad, N + 2 * pad, 3), dtype=np.float32).astype(np.float64)
    sharp = _smooth7(sharp) * 0.8 + 0.1
    psf_true = gaussian_psf(MK).astype(np.float64)
    if per_channel_psf:
        for c in range(3):
            k = gaussian_psf(MK, MK / 6.0 * (1.0 + 0.15 * c))[..., 0].astype(np.float64)
            psf_true[..., c] = k
    image = np.stack([_conv_direct(sharp[..., c], psf_true[..., c], "valid") for c in range(3)], axis=-1)
    image = image + noise * rng.standard_normal(image.shape)
    image = np.ascontiguousarray(image, dtype=np.float32)
    u0 = np.ascontiguousarray(np.pad(image, ((pad, pad), (pad, pad), (0, 0)), mode="edge"), dtype=np.float32)
    psf0 = uniform_psf(MK) if blind else np.ascontiguousarray(psf_true, dtype=np.float32)
    return dict(image=image, u0=u0, psf0=psf0, psf_true=psf_true.astype(np.float32), pad=pad)


def synth_case_large(M, N, MK, seed=0, blind=False, noise=1e-3):
    """synth_case for BASELINE-size frames (2048^2 ... 6144^2): the same recipe with the true Gaussian PSF applied as a row
    pass and a column pass (it is an outer product), elementwise float64 operations only -- seconds instead of minutes, and
    bit-reproducible wherever numpy is.  Used by oracle/make_golden_baseline.py and the tests that read its fixtures."""
    rng = np.random.default_rng(seed)
    pad = MK // 2
    sharp = rng.random((M + 2 * pad, N + 2 * pad, 3), dtype=np.float32).astype(np.float64)
    sharp = _smooth7(sharp) * 0.8 + 0.1
    n = np.arange(MK) - (MK - 1) / 2.0
    w = np.exp(-0.5 * (n / (MK / 6.0)) ** 2)
    w /= w.sum()
    t = np.zeros((M, N + 2 * pad, 3))
    for i in range(MK):
        t += w[i] * sharp[i:i + M]
    image = np.zeros((M, N, 3))
    for i in range(MK):
        image += w[i] * t[:, i:i + N]
    del t, sharp
    image += noise * rng.standard_normal(image.shape)
    image = np.ascontiguousarray(image, dtype=np.float32)
    u0 = np.ascontiguousarray(np.pad(image, ((pad, pad), (pad, pad), (0, 0)), mode="edge"), dtype=np.float32)
    psf_true = gaussian_psf(MK)
    psf0 = uniform_psf(MK) if blind else psf_true.copy()
    return dict(image=image, u0=u0, psf0=psf0, psf_true=psf_true, pad=pad)


BLACK_KINDS = ("band_mid", "band_top", "band_bot", "letterbox", "cols_mid", "rect")


def black_case(M, N, MK, kind, seed=0, blind=False):
    """synth_case with an exactly-black region in the image AND in u0 (clipped shadows, letterbox bars, zero borders:
    deconvolve.py:100-103 maps 0 -> 0).  The region is at least 2 MK + 8 px deep, so that it holds pixels whose whole
    (2 MK - 1)^2 dependency window is black: there the back-projection of an exact convolution is exactly 0 and pyx:499 is 0/0.
      band_mid / band_top / band_bot : full-width rows (top / bottom: the edge-padded border of u is black too);
      letterbox : black bars top and bottom plus a saturated (1.0) 8 x 8 patch in the picture;
      cols_mid  : full-height columns;   rect : a black rectangle inside the picture.
    Returns synth_case's dict with `image` / `u0` replaced and `black` = boolean mask of the image pixels set to 0."""
    case = synth_case(M, N, MK, seed=seed, blind=blind)
    img = case["image"].copy()
    d = 2 * MK + 8
    m = np.zeros((M, N), bool)
    if kind == "band_mid":
        m[(M - d) // 2:(M - d) // 2 + d] = True
    elif kind == "band_top":
        m[:d] = True
    elif kind == "band_bot":
        m[M - d:] = True
    elif kind == "letterbox":
        m[:d] = True
        m[M - d:] = True
        img[M // 2 - 4:M // 2 + 4, N // 2 - 4:N // 2 + 4] = 1.0
    elif kind == "cols_mid":
        m[:, (N - d) // 2:(N - d) // 2 + d] = True
    elif kind == "rect":
        m[(M - d) // 2:(M - d) // 2 + d, (N - d) // 2:(N - d) // 2 + d] = True
    else:
        raise ValueError(kind)
    img[m] = 0.0
    pad = MK // 2
    case["image"] = img
    case["u0"] = np.ascontiguousarray(np.pad(img, ((pad, pad), (pad, pad), (0, 0)), mode="edge"), dtype=np.float32)
    case["black"] = m
    return case


def default_window(M, N, MK, size=255):
    """Stats window as the driver passes it (deconvolve.py:281): (pad+1, size-pad-1) twice, clipped
    so that it stays inside small test frames."""
    pad = MK // 2
    size = min(size, min(M, N) - 2)
    if size % 2 == 0:
        size -= 1
    lo, hi = pad + 1, size - pad - 1
    return lo, hi, lo, hi
