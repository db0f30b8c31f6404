"""Test configuration.

`-m "not gpu"` (run in the GPU-less build container): oracle vs golden vectors, host logic, and
that libics_hip.so loads and exports every symbol declared in include/ics_hip.h.
`-m gpu` (run on a real MI355X): parity of the HIP path against the oracle and the goldens, always
through the C ABI.  Only tests may import `oracle/`; the product package never does.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-cases-studies_amd")
for p in (os.path.join(ROOT, "oracle"), PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ctx():
    """A libics_hip context on the default device; fails loudly (no CPU fallback) if there is none."""
    from lib import _native
    return _native.Context.get()
