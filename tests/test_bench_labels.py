"""bench.py's precision / kernel-family labels come from the library's routing (ics_describe, include/ics_hip.h), not from a rule of
bench.py's own: round 3 shipped `matrix = conv in (0, 2) and MK <= 37` while ICS_CONV_AUTO had been matrix-core at every size for a
day, so `--psf 39 ... 127` printed "f32" / "vector" for runs that used the fp16-split kernels.  CPU only: ics_describe needs no device."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _route(M, MK, blind, conv=0, tv_mode=0, fuse=0, flags=0):
    from lib import _native as nv
    return nv.describe(M, M, MK, nv.RLJob.params(1, 200, 1, 200, 1e9, 1, 1e-3, 1e4, blind, conv=conv, tv_mode=tv_mode, fuse=fuse, flags=flags))


@pytest.mark.parametrize("MK", [15, 31, 45, 65])
@pytest.mark.parametrize("blind", [False, True])
def test_labels_follow_the_library_routing(MK, blind):
    import bench
    from lib import _native as nv
    for conv in (nv.CONV_AUTO, nv.CONV_VECTOR):
        r = _route(2048, MK, blind, conv)
        lab = bench.labels(r)
        # the precision label is a statement about the products of the convolutions that ran
        assert ("fp16x2-split MFMA convolutions" in lab["dtype"]) == bool(r.conv_fp16_split)
        assert lab["matrix"] == bool(r.conv_fp16_split)
        assert lab["traffic_key"] == ("kernels_fft" if r.conv_family == 5 else ("kernels_matrix" if r.conv_fp16_split else "kernels_vector"))
        assert lab["conv"] == nv.RLRoute.CONV_FAMILIES[r.conv_family] and lab["gradk"] == nv.RLRoute.GRADK_FAMILIES[r.gradk_family]
        if conv == nv.CONV_AUTO and 19 <= MK <= 97:      # round 5: fp32 transform tiles for wide PSFs on frames >= 1.5 Mpx, the PSF gradient included
            assert r.conv_fp16_split == 0 and r.conv_family == 5 and "FFT tiles" in lab["dtype_note"] and lab["dtype"] == "f32"
            assert r.gradk_family == (7 if blind else 0) and r.gradk_fp16_split == 0      # round 6: A11 + A13 fused on the tiles
            small = _route(512, MK, blind, conv)         # ... and the matrix cores below that
            assert small.conv_fp16_split == 1 and small.conv_family == (1 if MK <= 49 else 2)
        elif conv == nv.CONV_AUTO and MK == 15 and not blind:      # round 6: non-blind 15 x 15 from 4 Mpx on the tiles (A1 + A3 as one unit)
            assert r.conv_fp16_split == 0 and r.conv_family == 5 and r.gradk_family == 0 and lab["dtype"] == "f32"
        elif conv == nv.CONV_AUTO:    # every PSF size has a matrix-core path since round 3
            assert r.conv_fp16_split == 1 and r.conv_family == (1 if MK <= 49 else 2)
            assert r.gradk_family == (0 if not blind else (1 if MK <= 15 else (2 if MK <= 31 else 3)))
        else:                         # fp32 products everywhere
            assert r.conv_fp16_split == 0 and r.gradk_fp16_split == 0 and lab["dtype"] == "f32"
            assert r.conv_family == (3 if MK <= 63 else 4)
    # the opt-in fused update + convolution kernel is a packed-fp32 kernel whatever AUTO resolves to
    # (compared where AUTO still resolves to the fp16-split matrix cores: 1024^2)
    assert bench.labels(_route(1024, 15, blind, 0, fuse=1), fuse=True)["dtype"] != bench.labels(_route(1024, 15, blind, 0))["dtype"]


def test_route_switches():
    from lib import _native as nv
    # transform tiles under AUTO (csrc/ics_api.hip fft_preferred, measured with scripts/ab_fft.py; round 6, with A11 + A13 and A1 + A3 fused on the tiles):
    # 19 x 19 ... 65 x 65 from 1.5 Mpx (blind: 1 Mpx), 17 x 17 from 4 Mpx (blind: 2 Mpx), 15 x 15 from 4 Mpx (blind: 6 Mpx), 9 x 9 ... 13 x 13 from 6 Mpx, smaller from 12 Mpx
    assert _route(4096, 17, True).conv_family == 5 and _route(4096, 17, False).conv_family == 5
    assert _route(2048, 17, True).conv_family == 5 and _route(2048, 17, False).conv_family == 5 and _route(1024, 17, True).conv_family == 1
    assert _route(1024, 31, True).conv_family == 5 and _route(1024, 31, False).conv_family == 1 and _route(1448, 21, False).conv_family == 5
    assert _route(4096, 15, True).conv_family == 5 and _route(4096, 15, False).conv_family == 5 and _route(2900, 15, True).conv_family == 5
    assert _route(2048, 15, True).conv_family == 1 and _route(2048, 15, False).conv_family == 5 and _route(1448, 15, False).conv_family == 1
    assert _route(6144, 13, True).conv_family == 5 and _route(2900, 9, True).conv_family == 5 and _route(2048, 9, False).conv_family == 1 and _route(2048, 13, True).conv_family == 1
    assert _route(4096, 5, True).conv_family == 5 and _route(2900, 5, True).conv_family == 1
    # 67 ... 97 (late round 6): the tiles from 0.5 Mpx (4-7 times the matrix cores' tap blocks); 99 ... 255: tap blocks on the tiles, same threshold
    assert _route(4096, 67, True).conv_family == 5 and _route(1024, 97, False).conv_family == 5 and _route(512, 67, True).conv_family == 2
    assert _route(4096, 99, True).conv_family == 5 and _route(4096, 127, False).conv_family == 5 and _route(2048, 255, True).gradk_family == 6
    assert _route(300, 129, True).conv_family == 2
    # the PAM kinds follow with their convolutions and PSF gradient (the TV term and the update stay on the HWC frames); active MM-TV does not
    assert _route(4096, 31, True, tv_mode=3).conv_family == 5 and _route(4096, 31, True, tv_mode=3).gradk_family == 7
    assert _route(4096, 31, True, tv_mode=1).conv_family == 1 and _route(2048, 15, False, tv_mode=2).conv_family == 1
    assert _route(2048, 15, True).gradk_family == 1 and _route(2048, 15, True, flags=nv.FLAG_NO_FUSED_GRADK).gradk_family == 2
    assert _route(4096, 15, True).gradk_family == 7 and _route(4096, 15, True, flags=nv.FLAG_NO_FUSED_GRADK).gradk_family == 6      # the tiles' own fused unit / two kernels
    assert _route(2048, 15, True).image_in_accumulator_order == 1 and _route(2048, 15, True, tv_mode=1).image_in_accumulator_order == 0
    assert _route(4096, 15, True).image_in_accumulator_order == 0
    assert _route(4096, 15, True).graph == 0 and _route(512, 9, False).graph == 0      # one hipGraph per outer iteration: opt-in (ICS_GRAPH=1; measured without gain)
    with pytest.raises(nv.NativeError) as ei:      # an explicit MATRIX request is never served by fp32 kernels
        _route(512, 55, True, conv=nv.CONV_MATRIX, tv_mode=2)
    assert ei.value.code == nv.ICS_ENOSUP
    # a planar mirror pads every plane row to 64 floats and can be LARGER than the HWC frame it mirrors -- 18784 x 9256 with a
    # 33 x 33 PSF passes the 2 GiB frame limit and its mirror does not fit the tile kernels' 32-bit offsets: AUTO keeps the matrix cores, an
    # explicit request for the tiles is refused
    pr = nv.RLJob.params(1, 200, 1, 200, 1e9, 1, 1e-3, 1e4, True)
    assert 0 < nv.frame_bytes(18784, 9256, 33) < nv.FRAME_LIMIT_BYTES
    assert nv.describe(18784, 9256, 33, pr).conv_family == 1 and nv.describe(9000, 9256, 33, pr).conv_family == 5
    with pytest.raises(nv.NativeError) as ei:
        nv.describe(18784, 9256, 33, nv.RLJob.params(1, 200, 1, 200, 1e9, 1, 1e-3, 1e4, True, conv=nv.CONV_FFT))
    assert ei.value.code == nv.ICS_ENOSUP and "mirror" in str(ei.value)
    big = _route(512, 129, True)                   # 129 ... 255 on frames below 0.5 Mpx: tap blocks on the matrix cores and nothing else
    assert (big.conv_family, big.conv_fp16_split, big.gradk_family) == (2, 1, 3)
    with pytest.raises(nv.NativeError) as ei:
        _route(512, 129, True, conv=nv.CONV_VECTOR)
    assert ei.value.code == nv.ICS_ENOSUP
    with pytest.raises(nv.NativeError):
        _route(512, 257, True)


def test_bench_help_and_docs_do_not_restate_a_size_rule():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "MK <= 37" not in src and "23..37" not in src
    dc = open(os.path.join(ROOT, "image-cases-studies_amd", "lib", "deconvolution.py")).read()
    assert "23..37" not in dc


def test_small_frames_route_to_the_cooperative_kernel_and_not_under_a_profiler():
    """Frames up to ~290 px a side, PSF <= 31, shipped loop: one cooperative launch per outer iteration (csrc/ics_small.hip; conv family 6, gradient
    family 8, fp32).  With a rocprofiler tool library preloaded the route is off by default -- a process that made a cooperative launch under
    rocprofv3 (ROCm 7.2) crashes in its exit handlers -- and ICS_SMALL_ITER=1 turns it back on.  (The switches are read once per process: subprocesses.)"""
    import json
    import subprocess
    import bench
    prog = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from lib import _native as nv\n"
            "P = nv.RLJob.params\n"
            "f = lambda M, K, blind: nv.describe(M, M, K, P(1, 100, 1, 100, 1e9, 1, 1e-3, 1e4, blind)).conv_family\n"
            "print(json.dumps([f(255, 15, True), f(255, 31, True), f(255, 15, False), f(255, 23, False), f(128, 7, False), f(300, 15, True), f(255, 33, True)]))\n"
            % (ROOT, os.path.join(ROOT, "image-cases-studies_amd")))
    def run(extra):
        env = {k: v for k, v in os.environ.items() if k not in ("ROCP_TOOL_LIBRARIES", "ICS_SMALL_ITER", "ICS_CONV_PATH")}
        env.update(extra)
        if env.pop("PROFILER", None):      # (a path that does not exist: the loader warns and goes on; ROCP_TOOL_LIBRARIES itself would be dlopened by the HIP runtime)
            env["LD_PRELOAD"] = ":".join(x for x in (os.environ.get("LD_PRELOAD", ""), "/nonexistent/librocprofiler-sdk-tool.so") if x)
        return json.loads(subprocess.run([sys.executable, "-c", prog], env=env, check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    assert run({}) == [6, 6, 1, 6, 6, 1, 1]
    assert 6 not in run({"PROFILER": "1"})
    assert run({"PROFILER": "1", "ICS_SMALL_ITER": "1"}) == [6, 6, 1, 6, 6, 1, 1]
    assert 6 not in run({"ICS_SMALL_ITER": "0"}) and 6 not in run({"ICS_CONV_PATH": "matrix"})
    from lib import _native as nv
    lab = bench.labels(nv.describe(255, 255, 15, nv.RLJob.params(1, 100, 1, 100, 1e9, 1, 1e-3, 1e4, True)))
    if lab["conv"] == "lds-resident":
        assert lab["dtype"] == "f32" and lab["gradk"] == "lds-resident" and not lab["matrix"]
