"""CPU oracle of the BUILD-DEFINED extended mode `tv_mode = 1` (active MM-TV).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference contains this arithmetic only as dead code.  `TV_ut_L1/TV_ut_L2` are
allocated as zeros and never written (lib/deconvolution.pyx:386-387; the candidate writers at :464-465 are
commented out and would not write them anyway), so the `if` branches at :517 and :543 never execute
(SURVEY.md section 0.1, 8c "Extended modes with no oracle").  This file defines what "active" means for
this build, following the reference's own formulas wherever they exist:

  per outer iteration (right after ut = u.copy(), :462):
      TV_ut_L1 = TV(ut, order=2, norm=1),  TV_ut_L2 = TV(ut, order=2, norm=2)              (:137-189)
  per inner iteration, before the update (:495-496):
      TV_u_L1 = TV(u, 2, 1);  TV_u_L2, div = TV(u, 2, 2)      (`div` keeps the norm-2 scaling: the second
                                                               call overwrites it, as in the reference)
      T = float32( div/TV_u_L1/TV_ut_L1/2. + div/TV_u_L2/TV_ut_L2/2. )   where both TV_*_L1 != 0, else 0
                                                               (:517/:543; the two double terms are summed
                                                               and STAGED AS FLOAT32 -- a build definition)
      gradu = float32( T + lambd*gradu + (u-ut)/4. )           where T is defined (:517), else :519
      dt_k, u -= dt_k*gradu                                    (:523-531, unchanged)
      gradu2 = T;  dt2_k = step*(max image_k + 0)/(max|gradu2_k| + 1e-15)
      image[..., k] -= dt2_k * gradu2[pad:-pad, pad:-pad, k] / lambd                         (:547-549)
      u[interior] = (1-DoF)*u[interior] + DoF*image            (:552, with the UPDATED image)
  epsilon = 1e-2 blind / 1e-6 non-blind (:434-437).  Everything else is rl_mm_oracle.richardson_lucy_MM.
"""
from __future__ import annotations

import numpy as np

import rl_mm_oracle as base
from rl_mm_oracle import F32, INNER_ITER, TV, Trace, dof_ratio, normalize_kernel, rotate_180, stop_weights, residual_whiteness


def tv_term(u, ut, epsilon):
    """T (float32, zero on the 1-px border) from the 3x3 neighbourhoods of u and ut."""
    M, N = u.shape[:2]
    tu1, _ = TV(u, M, N, epsilon, 2, 1)
    tu2, div = TV(u, M, N, epsilon, 2, 2)
    tt1, _ = TV(ut, M, N, epsilon, 2, 1)
    tt2, _ = TV(ut, M, N, epsilon, 2, 2)
    T = np.zeros_like(u)
    act = (tt1 != 0) & (tu1 != 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = ((div / tu1) / tt1).astype(np.float64) / 2.0
        b = ((div / tu2) / tt2).astype(np.float64) / 2.0
        T[act] = (a + b).astype(np.float32)[act]
    return T, act


def richardson_lucy_MM_tv(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations, step_factor, lambd,
                          blind=True, correlation=False, *, conv="direct", trace: Trace | None = None, quiet=True):
    """tv_mode = 1.  Mutates image (!), u and psf in place; returns the view of u like the reference."""
    cv = base._conv_scipy if conv == "scipy" else base._conv_direct
    tr = trace if trace is not None else Trace()
    step_factor, lambd, tau = F32(step_factor), F32(lambd), F32(tau)
    u_M, u_N = u.shape[:2]
    pad = (u_M - M) // 2
    epsilon = 1e-2 if blind else 1e-6
    gradk = np.zeros((MK, MK, 3), np.float32)
    ut = np.zeros_like(u)
    gradu = np.zeros_like(u)
    error = np.zeros((M, N, 3), np.float32)
    weights = stop_weights(top, bottom, left, right)
    psf_rotated = rotate_180(psf)
    interior = (slice(pad, u_M - pad), slice(pad, u_N - pad))
    it, stop_flag = 0, False
    M_r = M_r_prev = F32(0)
    while it < iterations and not stop_flag:
        ut[:] = u
        for _ in range(INNER_ITER):
            synth = np.stack([cv(u[..., c], psf[..., c], "valid") for c in range(3)], axis=-1).astype(np.float32)
            error[:] = synth - image
            for k in range(3):
                gradu[..., k] = cv(error[..., k], psf_rotated[..., k], "full")
            T, act = tv_term(u, ut, epsilon)
            with np.errstate(divide="ignore", invalid="ignore"):
                gi = gradu[interior]
                DoF = dof_ratio(gi, image) ** 2      # build-defined modes: 0/0 -> 1 always (rl_mm_oracle.dof_ratio)
                if not blind:
                    DoF = DoF / lambd
            lg = (lambd * gradu).astype(np.float64)
            d = (u - ut).astype(np.float64)
            gradu[:] = np.where(act, T.astype(np.float64) + lg + d / 4.0, lg + d / 2.0).astype(np.float32)
            dt = np.zeros(3, np.float32)
            for k in range(3):
                dt[k] = F32(step_factor * F32(np.amax(u[..., k]))) / F32(np.amax(np.abs(gradu[..., k])) + F32(1e-15))
            tr.dt.append(dt.copy())
            for k in range(3):
                u[..., k] -= dt[k] * gradu[..., k]
            for k in range(3):                                                     # :547-549
                dt2 = F32(step_factor * F32(np.amax(image[..., k]))) / F32(np.amax(np.abs(T[..., k])) + F32(1e-15))
                image[..., k] -= (dt2 * T[interior][..., k]) / lambd
            u[interior] = (F32(1.0) - DoF) * u[interior] + DoF * image
            if blind:
                for c in range(C):
                    error[..., c] = cv(u[..., c], psf[..., c], "valid")
                error -= image
                u_rot = rotate_180(u)
                for c in range(C):
                    gradk[..., c] = cv(u_rot[..., c], error[..., c], "valid")
                dtpsf = F32(F32(step_factor / F32(MK)) * F32(np.amax(psf))) / F32(np.amax(np.abs(gradk)) + F32(1e-15))
                psf -= dtpsf * gradk
                if correlation:
                    m = np.mean(psf, axis=2)
                    psf = np.dstack((m, m, m))
                normalize_kernel(psf, MK)
                psf_rotated = rotate_180(psf)
        if it > 0:
            M_r_prev = M_r
        M_r = residual_whiteness(error[top:bottom, left:right, ...], weights, base._conv_scipy)
        tr.M_r.append(M_r)
        if it > 1:
            if blind:
                stop_flag = bool(M_r > M_r_prev)
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    stop_flag = bool(F32(M_r - M_r_prev) / F32(M_r + M_r_prev) > tau)
        it += 1
    tr.iterations, tr.stopped, tr.psf_final = it, stop_flag, psf.copy()
    return u[pad:pad + M, pad:pad + N, ...]


# =================================================================================================
# tv_mode = 2 / 3: PAM (projected alternating minimisation, Perrone & Favaro 2014; README.md:42,106 of the
# reference describe it in prose only) with an isotropic (2) or collaborative L-inf,1,1 (3; Duran, Moeller,
# Sbert, Cremers, IPOL 2016, README.md:113-114) total-variation gradient.  BUILD-DEFINED, PARITY UNPINNED.
#
#   u-step :  G = lambd * k_ (*) (k * u - f)  -  div(p),      u <- u - dt_k * G,
#             dt_k = step * max(u_k) / (max|G_k| + 1e-15)     (the reference's max-normalised step, pyx:524)
#             no majoriser term, no DoF blend, image untouched
#   p      :  forward differences  dx u = u[i+1,j] - u[i,j],  dy u = u[i,j+1] - u[i,j]   (float32)
#             isotropic     p_d,c = d_d u_c / sqrt(dx u_c^2 + dy u_c^2 + eps^2)
#             collaborative p_d,c = [c == argmax_c' |d_d u_c'|] * d_d u_c / sqrt(d_d u_c^2 + eps^2)
#                           (first maximal channel wins ties), i.e. the (sub)gradient of sum_px sum_d max_c |d_d u_c|
#   div    :  backward differences  (p_x[i,j] - p_x[i-1,j]) + (p_y[i,j] - p_y[i,j-1]);  the stored term is
#             T = -div(p) on the interior of the u-frame and 0 on its 1-px border, G = float32(T + lambd*gradu)
#   PSF    :  gradient step, clamp, normalise exactly as lib/deconvolution.pyx:555-589
# =================================================================================================
def pam_tv_term(u, epsilon, collaborative):
    u = np.asarray(u, np.float32)
    M, N = u.shape[:2]
    eps = F32(epsilon)
    px = np.zeros_like(u)
    py = np.zeros_like(u)
    dx = np.zeros_like(u); dx[:-1] = u[1:] - u[:-1]
    dy = np.zeros_like(u); dy[:, :-1] = u[:, 1:] - u[:, :-1]
    if not collaborative:
        nrm = np.sqrt(dx * dx + dy * dy + eps * eps).astype(np.float32)
        px, py = dx / nrm, dy / nrm
    else:
        for d, p in ((dx, px), (dy, py)):
            sel = np.argmax(np.abs(d), axis=2)                      # first maximal channel
            mask = np.zeros(d.shape, bool)
            np.put_along_axis(mask, sel[..., None], True, axis=2)
            p[...] = np.where(mask, d / np.sqrt(d * d + eps * eps).astype(np.float32), F32(0))
    div = np.zeros_like(u)
    div[1:] += px[1:] - px[:-1]
    div[:, 1:] += py[:, 1:] - py[:, :-1]
    T = np.zeros_like(u)
    T[1:M - 1, 1:N - 1] = -div[1:M - 1, 1:N - 1]
    return T


def argmax_margin(u):
    """Per pixel: the smallest gap, over the two directions, between the largest and the second largest |forward difference| of the
    three channels -- how close the collaborative term's arg-max channel (pam_tv_term) is to flipping.  A pixel whose margin is below
    the rounding differences of two implementations (~1e-7 x the data range) may legitimately pick different channels."""
    u = np.asarray(u, np.float32)
    dx = np.zeros_like(u); dx[:-1] = u[1:] - u[:-1]
    dy = np.zeros_like(u); dy[:, :-1] = u[:, 1:] - u[:, :-1]
    out = np.full(u.shape[:2], np.inf, np.float32)
    for d in (dx, dy):
        a = np.sort(np.abs(d), axis=2)
        out = np.minimum(out, a[..., 2] - a[..., 1])
    return out


def richardson_lucy_PAM(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations, step_factor, lambd,
                        blind=True, correlation=False, *, collaborative=False, conv="direct", trace: Trace | None = None,
                        margins: list | None = None):
    cv = base._conv_scipy if conv == "scipy" else base._conv_direct
    tr = trace if trace is not None else Trace()
    step_factor, lambd, tau = F32(step_factor), F32(lambd), F32(tau)
    u_M, u_N = u.shape[:2]
    pad = (u_M - M) // 2
    epsilon = 1e-2 if blind else 1e-6
    gradk = np.zeros((MK, MK, 3), np.float32)
    gradu = np.zeros_like(u)
    error = np.zeros((M, N, 3), np.float32)
    weights = stop_weights(top, bottom, left, right)
    psf_rotated = rotate_180(psf)
    it, stop_flag = 0, False
    M_r = M_r_prev = F32(0)
    while it < iterations and not stop_flag:
        for _ in range(INNER_ITER):
            synth = np.stack([cv(u[..., c], psf[..., c], "valid") for c in range(3)], axis=-1).astype(np.float32)
            error[:] = synth - image
            for k in range(3):
                gradu[..., k] = cv(error[..., k], psf_rotated[..., k], "full")
            if margins is not None:
                margins.append(argmax_margin(u))          # (tests: which pixels are near an arg-max tie in this inner iteration)
            T = pam_tv_term(u, epsilon, collaborative)
            gradu[:] = (T.astype(np.float64) + (lambd * gradu).astype(np.float64)).astype(np.float32)
            for k in range(3):
                dt = F32(step_factor * F32(np.amax(u[..., k]))) / F32(np.amax(np.abs(gradu[..., k])) + F32(1e-15))
                u[..., k] -= dt * gradu[..., k]
            if blind:
                for c in range(C):
                    error[..., c] = cv(u[..., c], psf[..., c], "valid")
                error -= image
                u_rot = rotate_180(u)
                for c in range(C):
                    gradk[..., c] = cv(u_rot[..., c], error[..., c], "valid")
                dtpsf = F32(F32(step_factor / F32(MK)) * F32(np.amax(psf))) / F32(np.amax(np.abs(gradk)) + F32(1e-15))
                psf -= dtpsf * gradk
                if correlation:
                    m = np.mean(psf, axis=2)
                    psf = np.dstack((m, m, m))
                normalize_kernel(psf, MK)
                psf_rotated = rotate_180(psf)
        if it > 0:
            M_r_prev = M_r
        M_r = residual_whiteness(error[top:bottom, left:right, ...], weights, base._conv_scipy)
        tr.M_r.append(M_r)
        if it > 1:
            if blind:
                stop_flag = bool(M_r > M_r_prev)
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    stop_flag = bool(F32(M_r - M_r_prev) / F32(M_r + M_r_prev) > tau)
        it += 1
    tr.iterations, tr.stopped, tr.psf_final = it, stop_flag, psf.copy()
    return u[pad:pad + M, pad:pad + N, ...]
