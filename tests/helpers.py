"""Shared helpers for the parity tests (numpy only; the oracle lives in oracle/)."""
import json
import os

import numpy as np

import rl_mm_oracle as orc


def load_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "rl_%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    return z, meta


def rel_err(a, b):
    """max |a-b| / max |b| (the '1e-4 relative' of the north star is on this norm)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def conv_valid64(u, psf):
    return np.stack([orc._conv_direct(u[..., c], psf[..., c], "valid") for c in range(3)], axis=-1)


def corr_full64(e, psf):
    rot = psf[::-1, ::-1]
    return np.stack([orc._conv_direct(e[..., c], rot[..., c], "full") for c in range(3)], axis=-1)


def gradk64(u, e):
    urot = u[::-1, ::-1]
    return np.stack([orc._conv_direct(urot[..., c], e[..., c], "valid") for c in range(3)], axis=-1)


def update_f32(u, ut, g_raw, image, step, lambd, blind, pad):
    """A5-A10 in numpy float32 with the reference's rounding (lib/deconvolution.pyx:499-552)."""
    F = np.float32
    step, lambd = F(step), F(lambd)
    M, N = image.shape[:2]
    inter = (slice(pad, pad + M), slice(pad, pad + N))
    with np.errstate(divide="ignore", invalid="ignore"):
        gi = g_raw[inter]
        DoF = orc.dof_ratio(gi, image) ** 2      # 0/0 -> 1: include/ics_hip.h "DoF ratio"
        if not blind:
            DoF = DoF / lambd
        g = ((lambd * g_raw).astype(np.float64) + (u - ut).astype(np.float64) / 2.0).astype(np.float32)
        dt = np.zeros(3, np.float32)
        for k in range(3):
            dt[k] = F(step * np.amax(u[..., k])) / F(np.amax(np.abs(g[..., k])) + F(1e-15))
        un = u.copy()
        for k in range(3):
            un[..., k] -= dt[k] * g[..., k]
        un[inter] = (F(1.0) - DoF) * un[inter] + DoF * image
    return un, dt, DoF


def psf_step_f32(psf, gradk, step, MK, correlation):
    """A14-A17 in numpy float32 (lib/deconvolution.pyx:574-589).  Returns (local psf, caller's psf)."""
    F = np.float32
    dtpsf = F(F(F(step) / F(MK)) * np.amax(psf)) / F(np.amax(np.abs(gradk)) + F(1e-15))
    p = psf - dtpsf * gradk
    caller = p.copy()
    if correlation:
        m = np.mean(p, axis=2)
        p = np.ascontiguousarray(np.dstack((m, m, m)), dtype=np.float32)
        orc.normalize_kernel(p, MK)
        return p, caller, dtpsf
    orc.normalize_kernel(p, MK)
    return p, p.copy(), dtpsf
