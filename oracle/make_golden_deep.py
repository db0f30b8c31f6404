#!/usr/bin/env python3
"""Reference trajectories at north_star's DEPTH (50 inner iterations and more) at BASELINE.json's frame sizes (build container only).

TEST INFRASTRUCTURE ONLY (data, no reference source).  The COMPILED REFERENCE (oracle/build_reference.py:
/root/reference/lib/deconvolution.pyx:341-675 itself) runs

  * bl_4096_k15_deep   configs[2] blind 4096 x 4096 x 3, 15 x 15: a CHAIN of five calls with iterations=2 -- 10 outer = 50 inner
                       iterations.  `it > 1` (pyx:643) never holds inside a 2-outer call, so the chain is the 50-iteration
                       trajectory free of the rounding-fragile M_r stop decision (an outer iteration is a pure function of
                       (image, u, psf): pyx:460-462 re-derive ut, pyx:441 psf_rotated, everything else is scratch);
                       snapshots after every call (10, 20, 30, 40, 50 inner iterations);
  * bl_4096_k15_stop   the same problem as ONE call with iterations=10: where the reference's own stop test ends it (pyx:643-654);
  * nb_2048_k15_deep   configs[1] non-blind 2048 x 2048, 15 x 15, step 1e-3, ONE call, 10 outer (50 inner), tau = 1e9;
  * nb_2048_k15_s1e-4  the same at step 1e-4, 50 outer (250 inner) -- SURVEY.md section 8c's long-run regime;
  * bl_6144_k31_deep   configs[3] blind 6144 x 6144, 31 x 31, ONE call with iterations=2 (10 inner);

on orc.synth_case_large(seed) inputs.  A fixture keeps what make_golden_baseline.py keeps (crops at the centre / a corner shared by
four 64 x 64 tiles / frame corner / frame origin, every n-th row and column, float64 moments and quadrant sums of the whole frame, the
PSF, the reference's stdout) plus the per-outer M_r / Hu / varu parsed from the reference's log lines, and, where affordable
(2048^2 only), the noise floor: the numpy oracle with float64 direct sums instead of scipy's complex64 FFT against the reference.
The numpy oracle is asserted bit-equal to the reference at 2048^2; ICS_GOLDEN_SKIP_ORACLE=1 skips that above 2048^2.

Usage: python oracle/make_golden_deep.py [name ...]"""
import json
import os
import re
import sys
import time

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_reference  # noqa: E402
import make_golden as mg  # noqa: E402
import make_golden_baseline as mb  # noqa: E402
import rl_mm_oracle as orc  # noqa: E402

CASES = [
    dict(name="nb_2048_k15_deep", M=2048, N=2048, MK=15, blind=0, iters=10, chain=1, step=1e-3, seed=2048, noise_floor=True),
    dict(name="nb_2048_k15_s1e-4", M=2048, N=2048, MK=15, blind=0, iters=50, chain=1, step=1e-4, seed=2048, noise_floor=False),
    dict(name="bl_4096_k15_deep", M=4096, N=4096, MK=15, blind=1, iters=2, chain=10, step=1e-3, seed=4096),
    dict(name="bl_4096_k15_stop", M=4096, N=4096, MK=15, blind=1, iters=10, chain=1, step=1e-3, seed=4096),
    dict(name="bl_6144_k31_deep", M=6144, N=6144, MK=31, blind=1, iters=2, chain=5, step=1e-3, seed=6144),
]

_num = r"([-+0-9.eE]+|nan|inf)"


def parse_log(log):
    """what the reference prints (pyx:603-621, 656-672): one `DoF : min | max` line per outer iteration, then the closing
    `Stats : autocovariance = M_r | ... | residual = Hu | variance/noise = varu/...` line -- six decimals each"""
    out = dict(dof_min=[], dof_max=[], M_r=[], Hu=[], varu=[])
    for line in log.splitlines():
        m = re.match(r"DoF : min = " + _num + r" \| max = " + _num, line)
        if m:
            out["dof_min"].append(float(m.group(1))); out["dof_max"].append(float(m.group(2)))
        m = re.match(r"Stats : autocovariance = " + _num + r" \| lamdba = " + _num + r" \| residual = " + _num + r" \| variance/noise = " + _num, line)
        if m:
            out["M_r"].append(float(m.group(1))); out["Hu"].append(float(m.group(3))); out["varu"].append(float(m.group(4)))
    return out


def keep(out, tag, u, psf, c):
    s, where = mb.samples(u, c)
    for k, v in s.items():
        out["u_%s_%s" % (k, tag)] = np.ascontiguousarray(v)
    uf = u.astype(np.float64)
    out["moments_%s" % tag] = np.array([uf.sum(), (uf ** 2).sum(), uf.min(), uf.max()])
    h2, w2 = uf.shape[0] // 2, uf.shape[1] // 2
    out["quadrants_%s" % tag] = np.array([[uf[a:a + h2, b:b + w2, ch].sum() for ch in range(3)] for a in (0, h2) for b in (0, w2)])
    out["psf_%s" % tag] = psf.copy()
    return where


def main():
    ref = build_reference.load()
    want = sys.argv[1:]
    for c in CASES:
        if want and c["name"] not in want:
            continue
        c = dict(c, corr=0, lambd=10000.0, tau=1e9)
        M, N, MK = c["M"], c["N"], c["MK"]
        c["row_step"] = 3 * (M // 16) // 2 + 1
        c["window"] = orc.default_window(M, N, MK)
        t0 = time.time()
        case = orc.synth_case_large(M, N, MK, seed=c["seed"], blind=bool(c["blind"]))
        print(c["name"], "inputs %.1f s" % (time.time() - t0), flush=True)
        skip_oracle = M > 2048 and os.environ.get("ICS_GOLDEN_SKIP_ORACLE") == "1"
        out, logs, where = {}, {}, None
        u_r = psf_r = u_o = psf_o = None
        for k in range(1, c["chain"] + 1):
            t0 = time.time()
            img_r, u_r, psf_r, log_r = mg.run_ref(ref, case, c, c["iters"], u_start=u_r, psf_start=psf_r)
            print("  reference call %d/%d (%d outer): %.1f s" % (k, c["chain"], c["iters"], time.time() - t0), flush=True)
            assert np.array_equal(img_r, case["image"])
            if not skip_oracle:
                t0 = time.time()
                _, u_o, psf_o, tr = mg.run_orc(case, c, c["iters"], u_start=u_o, psf_start=psf_o)
                print("  oracle: %.1f s" % (time.time() - t0), flush=True)
                assert np.array_equal(u_r, u_o) and np.array_equal(psf_r, psf_o) and log_r == tr.log.getvalue()
            tag = str(k * c["iters"])
            where = keep(out, tag, u_r, psf_r, c)
            logs[tag] = log_r
            sc = parse_log(log_r)
            for key, v in sc.items():
                out["%s_%s" % (key, tag)] = np.array(v, np.float64)
            print("    log scalars:", {key: len(v) for key, v in sc.items()}, "lines", len(log_r.splitlines()), flush=True)
        done = len(parse_log(logs[tag])["dof_min"])
        noise = None
        if c.get("noise_floor"):
            t0 = time.time()
            _, u_d, psf_d, tr_d = mg.run_orc(case, c, c["iters"], conv="direct")
            noise = [mg.rel(u_d, u_r), mg.rel(psf_d, psf_r), tr_d.iterations]
            print("  float64-direct oracle: %.1f s, noise floor u %.2e psf %.2e" % (time.time() - t0, noise[0], noise[1]), flush=True)
        meta = dict(c, logs=logs, where=where, generator="synth_case_large", tags=[str(k * c["iters"]) for k in range(1, c["chain"] + 1)],
                    outer_done_last_call=done, noise_floor=noise,
                    versions=dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                                  reference="aurelienpierre/Image-Cases-Studies lib/deconvolution.pyx (cython language_level=2, -O3 -fopenmp)"))
        out["meta"] = np.array(json.dumps(meta))
        path = os.path.join(mg.OUT, "rl_%s.npz" % c["name"])
        np.savez_compressed(path, **out)
        print(path, "%.1f KB, outer iterations in the last call: %d" % (os.path.getsize(path) / 1024, done), flush=True)


if __name__ == "__main__":
    main()
