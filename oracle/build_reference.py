#!/usr/bin/env python3
"""Build the *compiled reference* (Cython) in a scratch directory OUTSIDE this repository.

TEST INFRASTRUCTURE ONLY.  The reference (`/root/reference/lib/deconvolution.pyx`) is a
Cython module; it cannot travel to the GPU box in any form, so it is only ever built and
imported in the build container, and only by `oracle/make_golden.py` (which writes the
small input/output fixtures under `tests/golden/`) and `oracle/check_oracle_vs_reference.py`.

Recipe:
  * language_level=2  -- mandatory: the shipped `lib/deconvolution.c` was generated with
    CYTHON_FUTURE_DIVISION 0, so `1/(u_M*u_N)` (pyx:524,548,574) is the integer 0.
  * -O3 -fopenmp, *without* the reference's -ffast-math/-march=native (setup.py:27) so the
    oracle is reproducible.
  * matplotlib must use the Agg backend before import (pyx:11 imports pyplot).

Nothing is written under /root/repo: the scratch dir defaults to /tmp/ics_reference_build.
"""
import os
import shutil
import subprocess
import sys
import textwrap

REFERENCE = os.environ.get("ICS_REFERENCE", "/root/reference")
SCRATCH = os.environ.get("ICS_REFERENCE_BUILD", "/tmp/ics_reference_build")


def build(force=False):
    src = os.path.join(REFERENCE, "lib", "deconvolution.pyx")
    if not os.path.isfile(src):
        raise FileNotFoundError("reference not present at %s (it only exists in the build container)" % src)
    libdir = os.path.join(SCRATCH, "lib")
    os.makedirs(libdir, exist_ok=True)
    have = [f for f in os.listdir(libdir) if f.startswith("deconvolution") and f.endswith(".so")]
    if have and not force:
        return SCRATCH
    shutil.copyfile(src, os.path.join(libdir, "deconvolution.pyx"))
    open(os.path.join(libdir, "__init__.py"), "w").close()
    with open(os.path.join(SCRATCH, "setup_ref.py"), "w") as f:
        f.write(textwrap.dedent("""
            import numpy
            from setuptools import setup, Extension
            from Cython.Build import cythonize
            ext = Extension("lib.deconvolution", ["lib/deconvolution.pyx"],
                            include_dirs=[numpy.get_include()],
                            extra_compile_args=["-O3", "-fopenmp", "-w"],
                            extra_link_args=["-fopenmp"])
            setup(name="ics_reference",
                  ext_modules=cythonize([ext], language_level=2, quiet=True),
                  script_args=["build_ext", "--inplace"])
        """))
    subprocess.check_call([sys.executable, "setup_ref.py"], cwd=SCRATCH)
    return SCRATCH


def load():
    """Import and return the compiled reference module `lib.deconvolution` (scratch copy)."""
    import importlib
    import matplotlib
    matplotlib.use("Agg")
    root = build()
    sys.path.insert(0, root)
    try:
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
            del sys.modules[k]
        mod = importlib.import_module("lib.deconvolution")
    finally:
        sys.path.remove(root)
        # do not leave the reference's `lib` package name bound: the product package is also `lib`
        for k in [k for k in sys.modules if k == "lib"]:
            del sys.modules[k]
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
