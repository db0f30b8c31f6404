#!/usr/bin/env python3
"""Generate tests/golden/utils.npz by importing the reference's lib/utils.py (build container only).

TEST INFRASTRUCTURE ONLY.  `numba` and `pyfftw` are not installed here; the reference imports them at
module level (lib/utils.py:14,17,24) but the functions exercised below never call into them, so the
harness registers inert module objects under those two names for the duration of the import, and
aliases scipy.signal.gaussian / exponential (removed from modern SciPy) to scipy.signal.windows.*.
The fixture stores inputs and the reference's outputs only.
"""
import json
import os
import sys
import types

import numpy as np
import scipy
import scipy.signal
import scipy.signal.windows

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "utils.npz")
REFERENCE = os.environ.get("ICS_REFERENCE", "/root/reference")


def import_reference_utils():
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f))
    class _T:
        def __getitem__(self, item): return self
        def __call__(self, *a, **k): return self
    numba.float32 = _T()
    pyfftw = types.ModuleType("pyfftw")
    sys.modules["numba"], sys.modules["pyfftw"] = numba, pyfftw
    if not hasattr(scipy.signal, "gaussian"):
        scipy.signal.gaussian = scipy.signal.windows.gaussian
        scipy.signal.exponential = scipy.signal.windows.exponential
    for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE)
    try:
        import importlib
        mod = importlib.import_module("lib.utils")
    finally:
        sys.path.remove(REFERENCE)
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.") or k in ("numba", "pyfftw")]:
            del sys.modules[k]
    return mod


def main():
    ru = import_reference_utils()
    sys.path.insert(0, HERE)
    import utils_oracle as uo
    rng = np.random.default_rng(42)
    out = {}
    for size in (3, 7, 15):
        out["uniform_%d" % size] = ru.uniform_kernel(size)
        out["gaussian_%d" % size] = ru.gaussian_kernel(size, size / 6.0)
        out["kaiser_%d" % size] = ru.kaiser_kernel(size, 3.5)
        out["poisson_%d" % size] = ru.poisson_kernel(size, 2.0)
        assert np.array_equal(out["gaussian_%d" % size], uo.gaussian_kernel(size, size / 6.0))
        assert np.array_equal(out["poisson_%d" % size], uo.poisson_kernel(size, 2.0))
    out["lens_9"] = ru.lens_blur(9)
    src = rng.random((61, 47))
    out["src"] = src
    out["gaussian_blur_7_1.5"] = ru.gaussian_blur(src, 7, 1.5)
    out["bessel_blur_9_4"] = ru.bessel_blur(src, 9, 4.0)
    out["gaussian_blur_4_1"] = ru.gaussian_blur(src, 4, 1.0)   # even-sized kernel: off-centre 'same'
    out["usm_bessel_5_3_0.7"] = ru.USM(src, 5, 3.0, 0.7)
    out["usm_gauss_5_1.2_1.5"] = ru.USM(src, 5, 1.2, 1.5, method="gauss")
    for k in ("gaussian_blur_7_1.5", "bessel_blur_9_4", "gaussian_blur_4_1"):
        pass
    assert np.array_equal(out["gaussian_blur_7_1.5"], uo.gaussian_blur(src, 7, 1.5))
    assert np.array_equal(out["usm_bessel_5_3_0.7"], uo.USM(src, 5, 3.0, 0.7))
    try:
        ru.bilateral_filter(src, 2, 0.1, 1.0, parallel=0)
        bilateral = "runs"
    except NameError as e:
        bilateral = "NameError: %s" % e
    out["meta"] = np.array(json.dumps(dict(numpy=np.__version__, scipy=scipy.__version__, bilateral_in_reference=bilateral)))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; reference bilateral_filter:", bilateral)


if __name__ == "__main__":
    main()
