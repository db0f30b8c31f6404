"""CPU: the C-ABI shared library loads and exports every symbol include/ics_hip.h declares, the
ctypes structs match the C layout, and -- on a machine without a GPU -- the product path fails
loudly instead of falling back to anything."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ics_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ics_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from lib import _native
    lib = _native.load()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libics_hip.so does not export %s" % n
    assert lib.ics_abi_version() == 4
    assert lib.ics_rl_params_size() == ctypes.sizeof(_native.RLParams) and lib.ics_rl_stats_size() == ctypes.sizeof(_native.RLStats)


def test_driver_entry_checks_the_current_abi_version():
    """__graft_entry__.build() asserts the ABI version of the library it has just built: it must name the binding's constant, not a
    literal that a version bump forgets (round 3 bumped 2 -> 3 and build() still compared with 2)."""
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "_native.ICS_ABI_VERSION" in src
    from lib import _native
    hdr = open(HEADER).read()
    assert re.search(r"#define ICS_ABI_VERSION %d\b" % _native.ICS_ABI_VERSION, hdr)


def test_struct_layout_matches_the_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    from lib import _native
    c = tmp_path / "layout.c"
    c.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ics_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                 'sizeof(ics_rl_params), sizeof(ics_rl_stats), offsetof(ics_rl_params, stop_test),'
                 'offsetof(ics_rl_stats, ms_total), offsetof(ics_rl_stats, launches), offsetof(ics_rl_params, progress),'
                 'offsetof(ics_rl_stats, trace_M_r), offsetof(ics_rl_params, top));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals == [ctypes.sizeof(_native.RLParams), ctypes.sizeof(_native.RLStats), _native.RLParams.stop_test.offset,
                    _native.RLStats.ms_total.offset, _native.RLStats.launches.offset, _native.RLParams.progress.offset,
                    _native.RLStats._p_M_r.offset, _native.RLParams.top.offset]
    assert _native.RLParams.struct_size.offset == 0 and _native.RLStats.struct_size.offset == 0
    c.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ics_hip.h"\nint main(void){printf("%zu %zu %zu\\n",'
                 'sizeof(ics_rl_route), offsetof(ics_rl_route, gradk_family), offsetof(ics_rl_route, graph));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals == [ctypes.sizeof(_native.RLRoute), _native.RLRoute.gradk_family.offset, _native.RLRoute.graph.offset]
    # ABI 4: the progress callback returns int (non-zero = stop after this outer iteration)
    assert _native.PROGRESS_FN._restype_ is ctypes.c_int
    assert re.search(r"typedef int \(\*ics_rl_progress_fn\)", open(HEADER).read())


def test_python_constants_equal_the_header():
    """lib/_native.py restates the header's #defines by hand: every one it restates must carry the header's value."""
    import re
    from lib import _native as nv
    text = open(HEADER).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(ICS_[A-Z0-9_]+)\s+\(?(-?(?:0x[0-9a-fA-F]+|\d+))(?:ull)?\)?", text)}
    pairs = {"ICS_ABI_VERSION": nv.ICS_ABI_VERSION, "ICS_ENOMEM": nv.ICS_ENOMEM, "ICS_ENOSUP": nv.ICS_ENOSUP, "ICS_ENODEV": nv.ICS_ENODEV,
             "ICS_CONV_AUTO": nv.CONV_AUTO, "ICS_CONV_VECTOR": nv.CONV_VECTOR, "ICS_CONV_MATRIX": nv.CONV_MATRIX, "ICS_CONV_FFT": nv.CONV_FFT,
             "ICS_FLAG_NO_FUSED_GRADK": nv.FLAG_NO_FUSED_GRADK, "ICS_FLAG_STAGE_ASYNC": nv.FLAG_STAGE_ASYNC, "ICS_FRAME_LIMIT_BYTES": nv.FRAME_LIMIT_BYTES,
             "ICS_STAGE_SYNTH_RESIDUAL": nv.STAGE_SYNTH_RESIDUAL, "ICS_STAGE_BAND_MASK_E": nv.STAGE_BAND_MASK_E, "ICS_STAGE_SYNTH_GRADK": nv.STAGE_SYNTH_GRADK, "ICS_STAGE_SYNTH_BACKPROJECT": nv.STAGE_SYNTH_BACKPROJECT}
    for name, value in pairs.items():
        assert name in defs, name
        assert defs[name] == value, (name, defs[name], value)


def test_debug_switches_are_not_in_the_public_header_and_round_trip():
    """ICS_TEST_* style hooks live behind ics_debug_set (csrc/ics_common.h), outside include/ics_hip.h, and no launch path
    calls getenv (round-2 verdict: test hooks in the production launch path)."""
    from lib import _native
    assert "ics_debug" not in open(HEADER).read()
    old = _native.debug_set("max_wgs", 5)
    assert _native.debug_set("max_wgs", old) == 5
    with pytest.raises(KeyError):
        _native.debug_set("no_such_switch", 1)
    csrc = os.path.join(ROOT, "image-cases-studies_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith(".hip"):
            txt = open(os.path.join(csrc, f)).read()
            assert "getenv" not in txt or f == "ics_group.hip", f      # (ICS_RCCL_LIB / ICS_GROUP_FORCE_RCCL: once per group)


def test_wrong_struct_size_is_refused_before_anything_else():
    """A caller built against another header (ics_rl_params grew in ABI 3) must get ICS_EINVAL, not a read past its struct."""
    from lib import _native
    lib = _native.load()
    p = _native.RLParams()                      # struct_size left 0
    st = _native.RLStats.with_traces(1)
    fake_job = ctypes.c_void_p(0)
    assert lib.ics_rl_run(fake_job, ctypes.byref(p), ctypes.byref(st)) == _native.ICS_EINVAL
    assert b"struct_size" in lib.ics_last_error()
    p.struct_size = ctypes.sizeof(_native.RLParams)
    st.struct_size = 12
    assert lib.ics_rl_run(fake_job, ctypes.byref(p), ctypes.byref(st)) == _native.ICS_EINVAL
    assert b"ics_rl_stats.struct_size" in lib.ics_last_error()


def test_no_silent_cpu_fallback_without_gpu():
    from lib import _native
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.NativeError) as ei:
        _native.Context(0)
    assert ei.value.code == _native.ICS_ENODEV
    from lib import deconvolution as dc
    img = np.zeros((9, 9, 3), np.float32)
    u = np.zeros((11, 11, 3), np.float32)
    psf = np.full((3, 3, 3), 1 / 9, np.float32)
    with pytest.raises(_native.NativeError):
        dc.richardson_lucy_MM(img, u, psf, 1, 8, 1, 8, 0.0, 9, 9, 3, 3, 1, 1e-3, 1.0, blind=False)
    with pytest.raises(_native.NativeError):
        dc.normalize_kernel(psf, 3)


def test_argument_validation_mirrors_cython_buffer_errors():
    from lib import deconvolution as dc
    a = np.zeros((9, 9, 3), np.float64)
    with pytest.raises(ValueError, match="Buffer dtype mismatch, expected 'DTYPE_t' but got 'double'"):
        dc.richardson_lucy_MM(a, a, a, 0, 1, 0, 1, 0, 9, 9, 3, 3, 1, 1e-3, 1.0)
    b = np.zeros((9, 9), np.float32)
    with pytest.raises(ValueError, match=r"Buffer has wrong number of dimensions \(expected 3, got 2\)"):
        dc.normalize_kernel(b, 3)
    assert dc.DTYPE is np.float32


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image-cases-studies_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                assert "rl_mm_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, os.path.join(d, f)
