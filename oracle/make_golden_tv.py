#!/usr/bin/env python3
"""tests/golden/tv.npz: the reference's `TV` (lib/deconvolution.pyx:137-239, `cdef` = invisible from Python) evaluated by the
COMPILED REFERENCE through a scratch-dir wrapper module (build container only).

TEST INFRASTRUCTURE ONLY.  The wrapper lives in the scratch directory of oracle/build_reference.py (outside the repository):
a three-line .pyx that Cython-`include`s the scratch copy of the reference file and adds `def tv_py(u, epsilon, order, norm)`,
compiled with the same flags as the reference (language_level=2, -O3 -fopenmp, no -ffast-math).  Only inputs and outputs are
stored here."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_reference  # noqa: E402
import rl_mm_oracle as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "tv.npz")


def build_wrapper():
    root = build_reference.build()
    libdir = os.path.join(root, "lib")
    with open(os.path.join(libdir, "deconvolution_tv.pyx"), "w") as f:
        f.write(textwrap.dedent('''
            include "deconvolution.pyx"

            def tv_py(np.ndarray[DTYPE_t, ndim=3] u, float epsilon, int order, int norm):
                cdef int M = u.shape[0]
                cdef int N = u.shape[1]
                out = np.zeros_like(u)
                div = np.zeros_like(u)
                cdef float[:, :, :] uv = u
                cdef float[:, :, :] ov = out
                cdef float[:, :, :] dv = div
                with nogil:
                    TV(uv, ov, M, N, epsilon, order, norm, dv)
                return out, div
        '''))
    with open(os.path.join(root, "setup_tv.py"), "w") as f:
        f.write(textwrap.dedent("""
            import numpy
            from setuptools import setup, Extension
            from Cython.Build import cythonize
            ext = Extension("lib.deconvolution_tv", ["lib/deconvolution_tv.pyx"], include_dirs=[numpy.get_include()],
                            extra_compile_args=["-O3", "-fopenmp", "-w"], extra_link_args=["-fopenmp"])
            setup(name="ics_reference_tv", ext_modules=cythonize([ext], language_level=2, quiet=True), script_args=["build_ext", "--inplace"])
        """))
    subprocess.check_call([sys.executable, "setup_tv.py"], cwd=root)
    return root


def main():
    import importlib
    import matplotlib
    matplotlib.use("Agg")
    root = build_wrapper()
    sys.path.insert(0, root)
    for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
        del sys.modules[k]
    tvmod = importlib.import_module("lib.deconvolution_tv")
    rng = np.random.default_rng(77)
    payload = {}
    cases = []
    inputs = {"rand_23x19": rng.random((23, 19, 3), dtype=np.float32),
              "smooth_40x33": orc.synth_case(36, 29, 5, seed=3)["u0"],
              "flat_9x11": np.full((9, 11, 3), 0.25, np.float32),
              "steps_16x16": (np.indices((16, 16)).sum(0) // 4 % 2)[..., None].repeat(3, 2).astype(np.float32) * np.float32(0.7) + np.float32(0.1)}
    for name, u in inputs.items():
        payload["u_" + name] = u
        for eps in (1e-2, 1e-6):
            for order in (1, 2):
                for norm in (1, 2):
                    out, div = tvmod.tv_py(np.ascontiguousarray(u), eps, order, norm)
                    key = "%s_e%g_o%d_n%d" % (name, eps, order, norm)
                    payload["out_" + key] = out
                    payload["div_" + key] = div
                    cases.append([name, eps, order, norm])
                    # the restatement, for the record (tests assert it)
                    o2, d2 = orc.TV(u, u.shape[0], u.shape[1], eps, order, norm)
                    print("%-34s oracle vs reference: out %s (max |d| %.2e), div %s (max |d| %.2e)" % (
                        key, np.array_equal(out, o2), float(np.max(np.abs(out - o2))), np.array_equal(div, d2), float(np.max(np.abs(div - d2)))))
    payload["meta"] = np.array(json.dumps(dict(cases=cases, numpy=np.__version__,
                                               reference="lib/deconvolution.pyx TV (:137-239) via a scratch-dir include wrapper, cython language_level=2, -O3 -fopenmp")))
    np.savez_compressed(OUT, **payload)
    print(OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024))


if __name__ == "__main__":
    main()
