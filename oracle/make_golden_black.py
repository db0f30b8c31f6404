#!/usr/bin/env python3
"""Reference results on frames with exactly-black regions (build container only; TEST INFRASTRUCTURE, data only).

Why these fixtures exist: pyx:499 computes ((gradu - image)/(gradu + image))^2.  Where image and u are exactly 0 the
reference's gradu is the rounding noise of scipy's complex64 FFT, and the ratio is 1 for every non-zero noise value: the
COMPILED REFERENCE returns a finite picture on black row bands and letterbox bars.  A convolution that is evaluated exactly
(the device kernels; the oracle with conv="direct") returns gradu = 0 there, IEEE 0/0 = NaN, and one NaN turns the whole frame
into NaN through the next convolution.  The library therefore defines the ratio as 1 where gradu == image == 0 exactly
(include/ics_hip.h "DoF ratio"), and these fixtures pin that rule to the reference itself.

Cases (inputs: orc.black_case(M, N, MK, kind, seed, blind); 2 outer iterations = 10 inner; blind and non-blind; MK = 9, 15, 31):
  * band_mid, band_top, band_bot, letterbox (with a saturated patch): black ROWS.  The reference's noise there is fine-grained
    (the last inverse transform runs along rows of pure noise) and an exact zero is rare: it returned a finite picture in 22 of
    these 24 cases (and all-NaN in letterbox_bl_k15 at 109 x 89 and letterbox_bl_k31: one exact zero among ~10^5 noise values
    is enough).  Where it
    is finite the generator asserts that the numpy oracle (scipy convolutions, IEEE ratio) equals it bit for bit and that the
    float64-direct oracle WITH the rule stays within 5e-5 (u; measured 1e-7 ... 9e-6) and 1e-6 (psf) of it; u, psf, M_r and the
    log are stored.
  * cols_mid, rect: black COLUMNS leave noise that is quantised by the cancellation of O(1) terms in the last (row-wise)
    inverse transform, so exact zeros -- and with them 0/0 -- are common: the reference returns an all-NaN frame at most sizes.
    Only the reference's NaN fraction and log are stored.
Every case records `ref_nan`; the GPU tests compare EVERY case with the float64-direct oracle (rule included, evaluated in the
test) and the finite ones with the reference's arrays as well, and print the reference's outcome beside the others.
Usage: python oracle/make_golden_black.py"""
import json
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_reference  # noqa: E402
import make_golden as mg  # noqa: E402
import rl_mm_oracle as orc  # noqa: E402

SIZES = {9: (65, 57), 15: (97, 89), 31: (129, 97)}
FINITE = ("band_mid", "band_top", "band_bot", "letterbox")
CHANCE = ("cols_mid", "rect")
ITERS = 2


def cases():
    for MK in (9, 15, 31):
        for kind in FINITE + CHANCE:
            for blind in (0, 1):
                M, N = SIZES[MK]
                if kind == "letterbox":
                    M = max(M, 2 * (2 * MK + 8) + 33)       # two bars of black_case's depth and a 33-row picture between them
                yield dict(name="%s_%s_k%d" % (kind, "bl" if blind else "nb", MK), kind=kind, M=M, N=N, MK=MK, blind=blind, corr=0,
                           step=1e-3, lambd=10000.0, tau=1e9, seed=100 * MK + blind, window=list(orc.default_window(M, N, MK)))


def logs_agree(a, b):
    """same lines, numbers within one unit of the last printed digit (with a black band inside the stop-test window the mean of
    the whiteness measure, pyx:638, differs between numpy builds' summation orders by 1e-6 relative: seen once in 36 cases)"""
    import re
    num = re.compile(r"-?\d+\.\d+|nan|inf")
    la, lb = a.splitlines(), b.splitlines()
    if len(la) != len(lb):
        return False
    for x, y in zip(la, lb):
        if num.sub("#", x) != num.sub("#", y):
            return False
        for p, q in zip(num.findall(x), num.findall(y)):
            if p != q and not (abs(float(p) - float(q)) <= 1.5e-6):
                return False
    return True


def main():
    ref = build_reference.load()
    out, metas = {}, {}
    for c in cases():
        case = orc.black_case(c["M"], c["N"], c["MK"], c["kind"], seed=c["seed"], blind=bool(c["blind"]))
        _, u_r, psf_r, log_r = mg.run_ref(ref, case, c, ITERS)
        _, u_o, psf_o, tr = mg.run_orc(case, c, ITERS)                       # scipy convolutions, IEEE ratio: the reference bit for bit
        assert np.array_equal(u_r, u_o, equal_nan=True) and np.array_equal(psf_r, psf_o, equal_nan=True), c["name"]
        assert logs_agree(log_r, tr.log.getvalue()), (c["name"], log_r, tr.log.getvalue())
        _, u_d, psf_d, tr_d = mg.run_orc(case, c, ITERS, conv="direct")      # float64 sums, 0/0 -> 1
        assert not np.isnan(u_d).any() and not np.isnan(psf_d).any(), c["name"]
        nan_frac = float(np.isnan(u_r).mean())
        m = dict(c, ref_nan=nan_frac, log=log_r, black_pixels=int(case["black"].sum()),
                 input_sums=[float(case["image"].astype(np.float64).sum()), float(case["u0"].astype(np.float64).sum())])
        if c["kind"] in FINITE and nan_frac == 0.0:
            eu, ep = mg.rel(u_d, u_r), mg.rel(psf_d, psf_r)
            assert eu < 5e-5 and ep < 1e-6, (c["name"], eu, ep)   # u: 2e-7 typical; 1e-5 at the corners of the saturated patch (FFT noise x lambd)
            m["direct_vs_ref"] = [eu, ep]
            out[c["name"] + "/u"] = u_r
            out[c["name"] + "/psf"] = psf_r
            out[c["name"] + "/M_r"] = np.array(tr.M_r, np.float32)
            out[c["name"] + "/dof"] = np.array([tr.dof_min, tr.dof_max], np.float32)
        metas[c["name"]] = m
        print("%-22s ref NaN %.3f  direct+rule vs ref: %s" % (c["name"], nan_frac, m.get("direct_vs_ref")), flush=True)
    meta = dict(cases=metas, iters=ITERS, finite=list(FINITE), chance=list(CHANCE),
                versions=dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                              reference="aurelienpierre/Image-Cases-Studies lib/deconvolution.pyx (cython language_level=2, -O3 -fopenmp)"))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(mg.OUT, "rl_black.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
