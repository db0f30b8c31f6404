"""The reference-side binding printed in INTEGRATION.md ("Option B") is extracted and run VERBATIM: on CPU it must parse, load the
library and agree with the header's struct sizes (the block itself asserts that at import); on the GPU it replaces
lib/deconvolution.pyx for a golden from the compiled reference.  Round-2 verdict: the stub had drifted from the header (two
fields short), so `ics_rl_run` read past the caller's struct -- this test is what keeps the documentation executable."""
import contextlib
import io
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-cases-studies_amd")


def stub_source():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = md.index("```python\n# lib/deconvolution.py in the reference tree") + len("```python\n")
    return md[a:md.index("```", a)]


def run_with_stub(tmp_path, body):
    """a child process whose `lib.deconvolution` is the extracted block (libics_hip.so found through LD_LIBRARY_PATH, as the
    stub's bare CDLL("libics_hip.so") needs)"""
    (tmp_path / "lib").mkdir(exist_ok=True)
    (tmp_path / "lib" / "__init__.py").write_text("")
    (tmp_path / "lib" / "deconvolution.py").write_text(stub_source())
    script = tmp_path / "drive.py"
    script.write_text("import sys\nsys.path.insert(0, %r)\n" % str(tmp_path) + textwrap.dedent(body))
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    return subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)


def test_stub_imports_and_matches_the_header(tmp_path):
    out = run_with_stub(tmp_path, """
        from lib import deconvolution as dc
        import ctypes as C
        print("SIZES", C.sizeof(dc._Params), C.sizeof(dc._Stats))
    """)
    assert out.returncode == 0, out.stderr[-2000:]
    sys.path.insert(0, PKG)
    from lib import _native
    import ctypes as C
    assert out.stdout.split()[-2:] == [str(C.sizeof(_native.RLParams)), str(C.sizeof(_native.RLStats))]


@pytest.mark.gpu
def test_stub_reproduces_a_reference_golden(tmp_path, golden_dir):
    out = run_with_stub(tmp_path, """
        import json, numpy as np
        from lib import deconvolution as dc
        z = np.load(%r)
        meta = json.loads(str(z["meta"]))
        M, N, MK = meta["M"], meta["N"], meta["MK"]
        res = {}
        for n in (1, 2):
            u, psf = z["u0"].copy(), z["psf0"].copy()
            big = np.zeros((u.shape[0] + 2, u.shape[1] + 6, 3), np.float32); big[1:-1, 3:-3] = u; uv = big[1:-1, 3:-3]    # a strided view, as deconvolve.py:278 passes
            out = dc.richardson_lucy_MM(z["image"].copy(), uv, psf, *meta["window"], meta["tau"], M, N, 3, MK, n, meta["step"], meta["lambd"], blind=True)
            assert np.shares_memory(out, big) and out.shape == (M, N, 3)
            den_u, den_p = np.abs(z["u_%%d" %% n]).max(), np.abs(z["psf_%%d" %% n]).max()
            res[n] = [float(np.abs(uv - z["u_%%d" %% n]).max() / den_u), float(np.abs(psf - z["psf_%%d" %% n]).max() / den_p)]
        print("RESULT " + json.dumps(res))
    """ % os.path.join(golden_dir, "rl_bl_65x49_k9.npz"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    res = json.loads([l for l in lines if l.startswith("RESULT ")][-1][7:])
    for n, (eu, ep) in res.items():
        assert eu < 1e-4 and ep < 1e-4, (n, eu, ep)          # the north-star bar; measured ~2e-7
    # the reference's progress lines came out of the callback
    meta = json.loads(str(np.load(os.path.join(golden_dir, "rl_bl_65x49_k9.npz"))["meta"]))
    ref_lines = meta["logs"]["2"].splitlines()
    got = [l for l in lines if not l.startswith("RESULT ")]
    assert sum(l.startswith("DoF : min") for l in got) == 3           # one per outer iteration of the two calls, from the callback
    assert ref_lines[-2] in got                                       # "Did not converge after 2 iterations. ..." as the reference prints it
