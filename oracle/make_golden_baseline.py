#!/usr/bin/env python3
"""Reference trajectories at the frame sizes BASELINE.json's metric is quoted on (build container only).

TEST INFRASTRUCTURE ONLY (data, no reference source).  The COMPILED REFERENCE (oracle/build_reference.py) runs
  * configs[1]  non-blind 2048 x 2048 x 3, 15 x 15 PSF, 2 outer iterations (10 inner),
  * the blind loop at 2048 x 2048, 2 outer iterations,
  * configs[2]  blind 4096 x 4096 x 3, 15 x 15, 1 outer iteration (5 inner) -- the headline workload,
  * configs[3]  blind 6144 x 6144 x 3, 31 x 31, 1 outer iteration,
on orc.synth_case_large(seed) inputs, and the fixture keeps what fits a small file: centre / tile-seam / corner crops of u,
every n-th row and every n-th column of u, float64 moments of the whole frame, the PSF, the reference's stdout and the
per-outer scalars (from the numpy oracle after asserting it equals the reference bit for bit on these sizes too;
ICS_GOLDEN_SKIP_ORACLE=1 skips that assertion for the 4096^2 case and stores the scalars parsed from the reference's log only).
Usage: python oracle/make_golden_baseline.py [name ...]"""
import json
import os
import sys
import time

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_reference  # noqa: E402
import make_golden as mg  # noqa: E402
import rl_mm_oracle as orc  # noqa: E402

CASES = [
    dict(name="nb_2048_k15", M=2048, N=2048, MK=15, blind=0, iters=[2], step=1e-3, seed=2048),
    dict(name="bl_2048_k15", M=2048, N=2048, MK=15, blind=1, iters=[2], step=1e-3, seed=2049),
    dict(name="bl_4096_k15", M=4096, N=4096, MK=15, blind=1, iters=[1], step=1e-3, seed=4096),
    # configs[3]: blind 6144 x 6144 x 3, 31 x 31 PSF, one outer iteration (the reference needs ~6 minutes and ~12 GB for it here)
    dict(name="bl_6144_k31", M=6144, N=6144, MK=31, blind=1, iters=[1], step=1e-3, seed=6144),
]


def samples(u, c):
    """the parts of the (M + 2 pad)^2 frame a fixture keeps; `c` records where they were taken"""
    H, W = u.shape[:2]
    cy, cx = H // 2, W // 2
    sy, sx = (H // 2 // 64) * 64 + c["MK"] // 2, (W // 3 // 64) * 64 + c["MK"] // 2   # a corner shared by four 64 x 64 tiles of the interior grid
    return dict(centre=u[cy - 48:cy + 48, cx - 48:cx + 48], seam=u[sy - 32:sy + 32, sx - 32:sx + 32], corner=u[-48:, -48:], origin=u[:48, :48],
                rows=u[::c["row_step"]], cols=u[:, ::c["row_step"]]), dict(centre=[cy - 48, cy + 48, cx - 48, cx + 48], seam=[sy - 32, sy + 32, sx - 32, sx + 32], corner=48, origin=48)


def main():
    ref = build_reference.load()
    want = sys.argv[1:]
    for c in CASES:
        if want and c["name"] not in want:
            continue
        c = dict(c, corr=0, lambd=10000.0, tau=1e9)
        M, N, MK = c["M"], c["N"], c["MK"]
        c["row_step"] = 3 * (M // 16) // 2 + 1           # ~11 rows and ~11 columns, not aligned with the tiles
        c["window"] = orc.default_window(M, N, MK)
        t0 = time.time()
        case = orc.synth_case_large(M, N, MK, seed=c["seed"], blind=bool(c["blind"]))
        print(c["name"], "inputs %.1f s" % (time.time() - t0), flush=True)
        out, logs, where = {}, {}, None
        skip_oracle = os.environ.get("ICS_GOLDEN_SKIP_ORACLE") == "1" and M >= 4096
        tr = None
        for n in c["iters"]:
            t0 = time.time()
            img_r, u_r, psf_r, log_r = mg.run_ref(ref, case, c, n)
            print("  reference, %d outer: %.1f s" % (n, time.time() - t0), flush=True)
            assert np.array_equal(img_r, case["image"])
            if not skip_oracle:
                t0 = time.time()
                _, u_o, psf_o, tr = mg.run_orc(case, c, n)
                print("  oracle: %.1f s" % (time.time() - t0), flush=True)
                assert np.array_equal(u_r, u_o) and np.array_equal(psf_r, psf_o) and log_r == tr.log.getvalue()
                del u_o
            s, where = samples(u_r, c)
            for k, v in s.items():
                out["u_%s_%d" % (k, n)] = np.ascontiguousarray(v)
            uf = u_r.astype(np.float64)
            out["moments_%d" % n] = np.array([uf.sum(), (uf ** 2).sum(), uf.min(), uf.max()])
            # per-channel sums over the four quadrants: a misplaced tile moves one of them
            h2, w2 = uf.shape[0] // 2, uf.shape[1] // 2
            out["quadrants_%d" % n] = np.array([[uf[a:a + h2, b:b + w2, ch].sum() for ch in range(3)] for a in (0, h2) for b in (0, w2)])
            del uf
            out["psf_%d" % n] = psf_r
            logs[str(n)] = log_r
        if tr is not None:
            out["M_r"] = np.array(tr.M_r, np.float32); out["Hu"] = np.array(tr.Hu, np.float32); out["varu"] = np.array(tr.varu, np.float32)
        meta = dict(c, logs=logs, where=where, generator="synth_case_large",
                    versions=dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                                  reference="aurelienpierre/Image-Cases-Studies lib/deconvolution.pyx (cython language_level=2, -O3 -fopenmp)"))
        out["meta"] = np.array(json.dumps(meta))
        path = os.path.join(mg.OUT, "rl_%s.npz" % c["name"])
        np.savez_compressed(path, **out)
        print(path, "%.1f KB" % (os.path.getsize(path) / 1024), flush=True)


if __name__ == "__main__":
    main()
