"""The cooperative small-frame iteration (ics_small.hip) against the multi-launch path on the same inputs, and its time:
    python scripts/dbg/small_check.py [quick]
Every case: outer iterations x 5 inner, u / psf / trace differences (relative to the largest value), then ms per inner iteration of both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import numpy as np
import bench
from lib import _native

ctx = _native.Context.get(0)
CASES = [(255, 255, 15, True, 0), (255, 255, 15, True, 1), (255, 255, 15, False, 0), (512, 512, 9, False, 0), (512, 512, 9, True, 0), (255, 255, 7, True, 0), (200, 131, 5, True, 0),
         (255, 255, 31, True, 0), (255, 255, 23, False, 0), (97, 64, 3, True, 0), (300, 500, 11, True, 0), (512, 512, 15, True, 0), (33, 40, 9, True, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    CASES = CASES[:4]
if len(sys.argv) > 1 and sys.argv[1] == "nonblind":   # where does the non-blind loop gain?
    CASES = [(M, M, K, False, 0) for M in (128, 255) for K in (3, 7, 11, 15, 19, 23, 31)]
worst = 0.0
for (M, N, MK, blind, corr) in CASES:
    image, u0, psf_true, psf_uniform = bench.synth_frame(M, N, MK, seed=3)
    pad = MK // 2
    win = (pad + 1, min(M, 255) - pad - 1, pad + 1, min(N, 255) - pad - 1)
    res = {}
    for v in (1, 0):
        _native.debug_set("small_iter", v)
        j = _native.RLJob(M, N, MK, ctx)
        j.upload(image, u0, psf_uniform if blind else psf_true)
        p = j.params(*win, 1e9, 4, 1e-3, 10000.0 if blind else 1000.0, blind, corr, 3, stop_test=2)
        r = j.describe(p)
        st = j.run(p)
        u, psf, psfc = j.download()
        sc = j.scalars()
        res[v] = (u, psf, psfc, np.array([st.M_r, st.Hu, st.varu, st.dof_min, st.dof_max]), r.conv_family, sc)
        ctx.synchronize()
        t0 = time.perf_counter(); j.run(j.params(*win, 1e9, 40, 1e-3, 10000.0 if blind else 1000.0, blind, corr, 3, stop_test=2)); ctx.synchronize()
        res[v] += ((time.perf_counter() - t0) * 1e3 / 200,)
        j.close()
    a, b = res[1], res[0]
    du = float(np.max(np.abs(a[0] - b[0])) / np.max(np.abs(b[0])))
    dp = float(np.max(np.abs(a[1] - b[1])) / np.max(np.abs(b[1])))
    dc = float(np.max(np.abs(a[2] - b[2])) / np.max(np.abs(b[2])))
    ds = float(np.max(np.abs(a[3] - b[3]) / (np.abs(b[3]) + 1e-30)))
    worst = max(worst, du, dp, dc)
    print("%4dx%-4d K %2d %s corr %d: families %d / %d   u %.2e  psf %.2e  caller's psf %.2e  stats %.2e   %.4f vs %.4f ms per inner iteration%s"
          % (M, N, MK, "blind   " if blind else "nonblind", corr, a[4], b[4], du, dp, dc, ds, a[6], b[6], "" if np.isfinite(a[0]).all() else "  NOT FINITE"), flush=True)
print("worst difference %.3e" % worst)
