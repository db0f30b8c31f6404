#!/usr/bin/env python3
"""Pin the numpy oracle against the compiled reference (build container only).

TEST INFRASTRUCTURE ONLY.  Runs `/root/reference/lib/deconvolution.pyx` (cythonized into a scratch
dir outside the repo by oracle/build_reference.py) and `oracle/rl_mm_oracle.py` on the same seeded
inputs and reports max |difference|.  With conv="scipy" the expectation is bit-exact equality.
"""
import contextlib
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_reference  # noqa: E402
import rl_mm_oracle as orc  # noqa: E402


def run_ref(ref, case, MK, iters, step, lambd, blind, correlation, window, tau):
    image, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    M, N = image.shape[:2]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = ref.richardson_lucy_MM(image, u, psf, *window, tau, M, N, 3, MK, iters, step, lambd,
                                     blind=blind, correlation=correlation)
    return image, u, psf, out, buf.getvalue()


def run_orc(case, MK, iters, step, lambd, blind, correlation, window, tau, conv="scipy"):
    image, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    M, N = image.shape[:2]
    tr = orc.Trace()
    out = orc.richardson_lucy_MM(image, u, psf, *window, tau, M, N, 3, MK, iters, step, lambd,
                                 blind=blind, correlation=correlation, conv=conv, trace=tr, quiet=True)
    return image, u, psf, out, tr


def main():
    ref = build_reference.load()
    worst = 0.0
    for (M, N, MK, blind, corr, iters, step) in [
        (33, 37, 3, False, False, 4, 1e-3), (65, 65, 7, False, False, 4, 1e-3),
        (65, 49, 9, True, False, 4, 1e-3), (65, 65, 7, True, True, 3, 1e-3),
        (129, 129, 15, True, False, 3, 1e-3), (129, 129, 15, False, False, 10, 1e-4),
    ]:
        case = orc.synth_case(M, N, MK, seed=M + MK, blind=blind)
        window = orc.default_window(M, N, MK)
        tau = 1e9
        r = run_ref(ref, case, MK, iters, step, 10000.0, blind, corr, window, tau)
        o = run_orc(case, MK, iters, step, 10000.0, blind, corr, window, tau)
        du = float(np.max(np.abs(r[1] - o[1])))
        dp = float(np.max(np.abs(r[2] - o[2])))
        di = float(np.max(np.abs(r[0] - o[0])))
        same_log = r[4] == o[4].log.getvalue()
        d = run_orc(case, MK, iters, step, 10000.0, blind, corr, window, tau, conv="direct")
        nf = float(np.max(np.abs(r[1] - d[1])) / np.max(np.abs(r[1])))
        print("M=%d N=%d MK=%d blind=%d corr=%d it=%d: |du|=%g |dpsf|=%g |dimage|=%g log_equal=%s  noise-floor(direct f64 conv) rel=%.2e"
              % (M, N, MK, blind, corr, iters, du, dp, di, same_log, nf))
        if not same_log:
            print("--- reference log ---\n" + r[4] + "--- oracle log ---\n" + o[4].log.getvalue())
        worst = max(worst, du, dp, di)
    print("WORST", worst)
    return 0 if worst == 0.0 else 1


if __name__ == "__main__":
    sys.exit(main())
