#!/usr/bin/env python3
"""Generate tests/golden/rl_*.npz from the COMPILED REFERENCE (build container only).

TEST INFRASTRUCTURE ONLY.  Each fixture holds inputs and the reference's outputs (data only --
no reference source).  The reference is `/root/reference/lib/deconvolution.pyx`, cythonized into a
scratch dir outside the repository by oracle/build_reference.py (language_level=2, -O3 -fopenmp).
Versions that pin the third-party convolution (scipy.signal.convolve) are recorded in each file.

Per case we store snapshots of the full padded `u` and the caller's `psf` after n outer
iterations (separate reference runs with iterations=n: an outer iteration is a pure function of
(image, u, psf), so snapshot[n] -> snapshot[n+1] is also a teacher-forced single step), the
reference's stdout, and the per-outer scalars (M_r, Hu, varu, DoF min/max) taken from the numpy
oracle *after* asserting that the oracle's arrays and log equal the reference's bit for bit.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_reference  # noqa: E402
import rl_mm_oracle as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = [
    # name, M, N, MK, blind, correlation, snapshots(outer iters), step, lambd, tau, extra
    dict(name="nb_33x37_k3", M=33, N=37, MK=3, blind=0, corr=0, snaps=[1, 2, 5, 10], step=1e-3),
    dict(name="nb_65x65_k7", M=65, N=65, MK=7, blind=0, corr=0, snaps=[1, 2, 5, 10], step=1e-3),
    dict(name="nb_65x81_k9_pcpsf", M=65, N=81, MK=9, blind=0, corr=0, snaps=[1, 2, 5], step=1e-3, per_channel_psf=True),
    dict(name="nb_129x129_k15", M=129, N=129, MK=15, blind=0, corr=0, snaps=[1, 2, 10, 20], step=1e-3),
    dict(name="nb_129x129_k15_s1e-4", M=129, N=129, MK=15, blind=0, corr=0, snaps=[50], step=1e-4),
    dict(name="nb_97x97_k5_tau", M=97, N=97, MK=5, blind=0, corr=0, snaps=[40], step=5e-3, tau=0.0),
    dict(name="bl_65x49_k9", M=65, N=49, MK=9, blind=1, corr=0, snaps=[1, 2, 5, 10], step=1e-3),
    dict(name="bl_129x129_k15", M=129, N=129, MK=15, blind=1, corr=0, snaps=[1, 2, 5], step=1e-3),
    dict(name="bl_65x65_k7_corr", M=65, N=65, MK=7, blind=1, corr=1, snaps=[1, 3], step=1e-3),
    dict(name="bl_101x101_k11_s1e-4", M=101, N=101, MK=11, blind=1, corr=0, snaps=[20], step=1e-4),
    # chain of calls with iterations=2: `it > 1` (pyx:643) never holds, so no stop decision is involved
    # and the chain is a 20-outer-iteration (100 inner) blind trajectory free of the fragile M_r test
    dict(name="bl_101x101_k11_chain", M=101, N=101, MK=11, blind=1, corr=0, snaps=[2], chain=10, step=1e-3),
]


def run_ref(ref, case, c, iters, strided=False, u_start=None, psf_start=None):
    image, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    if u_start is not None:
        u, psf = u_start.copy(), psf_start.copy()
    if strided:  # non-contiguous views, as deconvolve.py:278-279 passes them
        big_i = np.zeros((image.shape[0] + 4, image.shape[1] + 6, 3), np.float32); big_i[2:-2, 3:-3] = image; image = big_i[2:-2, 3:-3]
        big_u = np.zeros((u.shape[0] + 2, u.shape[1] + 10, 3), np.float32); big_u[1:-1, 5:-5] = u; u = big_u[1:-1, 5:-5]
    M, N = image.shape[:2]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = ref.richardson_lucy_MM(image, u, psf, *c["window"], c["tau"], M, N, 3, c["MK"], iters,
                                     c["step"], c["lambd"], blind=c["blind"], correlation=c["corr"])
    assert np.shares_memory(out, u)
    return np.ascontiguousarray(image), np.ascontiguousarray(u), psf, buf.getvalue()


def run_orc(case, c, iters, conv="scipy", u_start=None, psf_start=None):
    image, u, psf = case["image"].copy(), case["u0"].copy(), case["psf0"].copy()
    if u_start is not None:
        u, psf = u_start.copy(), psf_start.copy()
    M, N = image.shape[:2]
    tr = orc.Trace()
    orc.richardson_lucy_MM(image, u, psf, *c["window"], c["tau"], M, N, 3, c["MK"], iters, c["step"], c["lambd"],
                           blind=c["blind"], correlation=c["corr"], trace=tr, quiet=True, conv=conv)
    return image, u, psf, tr


def rel(a, b):
    den = float(np.max(np.abs(b.astype(np.float64))))
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / (den if den > 0 else 1.0))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = build_reference.load()
    versions = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                    reference="aurelienpierre/Image-Cases-Studies lib/deconvolution.pyx (cython language_level=2, -O3 -fopenmp)")
    for c in CASES:
        c.setdefault("tau", 1e9)
        c.setdefault("lambd", 10000.0)
        c.setdefault("seed", c["M"] * 7 + c["MK"])
        case = orc.synth_case(c["M"], c["N"], c["MK"], seed=c["seed"], blind=bool(c["blind"]),
                              per_channel_psf=c.get("per_channel_psf", False))
        c["window"] = orc.default_window(c["M"], c["N"], c["MK"])
        payload = dict(image=case["image"], u0=case["u0"], psf0=case["psf0"])
        meta = dict(c)
        meta["versions"] = versions
        logs = {}
        noise = {}
        for n in c["snaps"]:
            img_r, u_r, psf_r, log_r = run_ref(ref, case, c, n)
            img_o, u_o, psf_o, tr = run_orc(case, c, n)
            assert np.array_equal(u_r, u_o, equal_nan=True) and np.array_equal(psf_r, psf_o, equal_nan=True), c["name"]
            assert log_r == tr.log.getvalue(), c["name"]
            assert np.array_equal(img_r, case["image"])  # pyx:549 subtracts exactly 0
            payload["u_%d" % n] = u_r
            payload["psf_%d" % n] = psf_r
            payload["psf_local_%d" % n] = tr.psf_final
            logs[str(n)] = log_r
            last = tr
            # noise floor: the same loop with float64 direct sums instead of scipy's complex64 FFT
            _, u_d, psf_d, tr_d = run_orc(case, c, n, conv="direct")
            noise[str(n)] = [rel(u_d, u_r), rel(psf_d, psf_r), tr_d.iterations]
        if c.get("chain"):
            u_c, psf_c = payload["u_2"], payload["psf_2"]
            u_o, psf_o = u_c, psf_c
            for k in range(2, c["chain"] + 1):
                _, u_c, psf_c, _ = run_ref(ref, case, c, 2, u_start=u_c, psf_start=psf_c)
                _, u_o, psf_o, _ = run_orc(case, c, 2, u_start=u_o, psf_start=psf_o)
                assert np.array_equal(u_c, u_o) and np.array_equal(psf_c, psf_o)
                if k in (c["chain"] // 2, c["chain"]):
                    payload["u_chain_%d" % k] = u_c
                    payload["psf_chain_%d" % k] = psf_c
        # strided-view run must equal the contiguous one
        _, u_s, psf_s, _ = run_ref(ref, case, c, c["snaps"][0], strided=True)
        assert np.array_equal(u_s, payload["u_%d" % c["snaps"][0]], equal_nan=True)
        payload["M_r"] = np.array(last.M_r, np.float32)
        payload["Hu"] = np.array(last.Hu, np.float32)
        payload["varu"] = np.array(last.varu, np.float32)
        payload["dof_min"] = np.array(last.dof_min, np.float32)
        payload["dof_max"] = np.array(last.dof_max, np.float32)
        payload["dt"] = np.array(last.dt, np.float32)
        payload["dtpsf"] = np.array(last.dtpsf, np.float32)
        meta["iterations_done"] = last.iterations
        meta["stopped"] = last.stopped
        meta["logs"] = logs
        meta["noise_floor"] = noise  # per snapshot: [rel err u, rel err psf, iterations] of the f64-direct oracle vs reference
        payload["meta"] = np.array(json.dumps(meta))
        path = os.path.join(OUT, "rl_%s.npz" % c["name"])
        np.savez_compressed(path, **payload)
        print("%-28s snaps=%s done=%d stopped=%s  %6.1f KB" % (c["name"], c["snaps"], last.iterations, last.stopped,
                                                              os.path.getsize(path) / 1024))

    # normalize_kernel (pyx:73-75)
    rng = np.random.default_rng(1234)
    nk = {}
    for MK in (3, 7, 15, 31):
        k = (rng.standard_normal((MK, MK, 3)) * 0.3 + 0.2).astype(np.float32)
        nk["in_%d" % MK] = k.copy()
        ref.normalize_kernel(k, MK)
        nk["out_%d" % MK] = k
        k2 = nk["in_%d" % MK].copy()
        orc.normalize_kernel(k2, MK)
        assert np.array_equal(k, k2)
    nk["meta"] = np.array(json.dumps(dict(versions=versions)))
    np.savez_compressed(os.path.join(OUT, "normalize_kernel.npz"), **nk)

    # BASELINE config 1: non-blind, 512x512x3, 9x9 Gaussian PSF, 20 outer iterations (inputs from seed;
    # output stored as a centre crop + float64 moments to keep the fixture small)
    c = dict(M=512, N=512, MK=9, blind=0, corr=0, step=1e-3, lambd=10000.0, tau=1e9, seed=0)
    case = orc.synth_case(512, 512, 9, seed=0)
    c["window"] = orc.default_window(512, 512, 9)
    _, u_r, psf_r, log_r = run_ref(ref, case, c, 20)
    uf = u_r.astype(np.float64)
    meta = dict(c, versions=versions, log=log_r, crop=[208, 304, 208, 304])
    np.savez_compressed(os.path.join(OUT, "rl_config1_512_k9_20.npz"),
                        u_crop=u_r[208:304, 208:304].copy(), u_rows=u_r[::64, :, :].copy(),
                        moments=np.array([uf.sum(), (uf ** 2).sum(), uf.min(), uf.max()]),
                        chan_sums=uf.sum(axis=(0, 1)), meta=np.array(json.dumps(meta)))
    print("config1 done:", log_r.strip().splitlines()[-1])


if __name__ == "__main__":
    main()
