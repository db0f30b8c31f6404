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


@pytest.fixture
def debug_switch():
    """Sets test / measurement switches of libics_hip.so (csrc/ics_common.h IcsDebug: "max_wgs", "dynamic_tiles", "conv_rs", ...)
    for the duration of a test: `debug_switch("max_wgs", 8)`.  The library reads its ICS_* environment variables once, at
    first use; tests change the switches through the (non-public) ics_debug_set entry instead."""
    from lib import _native
    undo = []

    def set_(name, value):
        undo.append((name, _native.debug_set(name, value)))

    yield set_
    for name, old in reversed(undo):
        _native.debug_set(name, old)
