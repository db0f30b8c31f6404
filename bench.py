#!/usr/bin/env python3
"""Headline benchmark: MPixels/s per inner iteration of Richardson-Lucy (MM) deconvolution on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode blind|nonblind] [--size 4096] [--psf 15]

A *step* is one inner iteration of `richardson_lucy_MM` (lib/deconvolution.pyx:473-591) over one
H x W x 3 fp32 frame; 5 steps make one outer iteration, which also pays for the `ut = u` copy and
the residual-whiteness stop test (evaluated on device every outer iteration, `stop_test=2`: computed
but never allowed to end the run, so that exactly K steps are timed).  Default workload =
BASELINE.json configs[2]: blind, 4096 x 4096 x 3, 15 x 15 initial PSF (the configuration the metric
is quoted on); `--mode nonblind` gives the non-blind inner iteration on the same frame.  Inputs are
synthetic and already resident in HBM when the timed region starts.

For N > 1 the driver launches one rank per GPU (torch.distributed.run); every rank deconvolves its
own frame (seed = rank), nothing is exchanged during the iterations, and the only collectives are
the barrier / max-over-ranks of the contract below (RCCL).  value = N * pixels * K / max-rank time.

One JSON line on rank 0, with `roofline` (dominant kernel: algorithmic bytes per launch over its
HIP-event duration, against 8 TB/s) and `cpu_baseline` (the numpy/scipy port of the reference loop in
oracle/, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# algorithmic bytes per pixel and launch (one fp32 x 3 frame transit T = 12 B/px)
BYTES_PER_PX = {"synth_residual": 36.0, "backproject": 48.0, "update": 60.0, "psf_gradient": 24.0, "synth_gradk": 60.0,
                "update_synth": 72.0,
                "synth_backproject": 84.0}  # update_synth: fused update + convolution, 4 reads + 2 writes; synth_backproject: A1 + A3 in one unit = S1 + S2 of SURVEY.md 8d (the residual is never materialised)
ITER_BYTES_PER_PX = {"nonblind": 144.0, "blind": 204.0}


def labels(route, fuse=False):
    """Precision / kernel-family labels of a run from the LIBRARY's routing (lib._native.describe / RLJob.describe -> ics_rl_route):
    what the JSON says about arithmetic must be what was launched, at every --psf (round-3 verdict: a K <= 37 test of bench.py's
    own had drifted from the routing).  Returns dict(matrix, dtype, dtype_note, conv, gradk, traffic_key)."""
    from lib import _native
    conv_f = _native.RLRoute.CONV_FAMILIES[route.conv_family]
    gradk_f = _native.RLRoute.GRADK_FAMILIES[route.gradk_family]
    split_conv = bool(route.conv_fp16_split) and not fuse
    split_gradk = bool(route.gradk_fp16_split)
    if route.conv_family == 5 and not fuse:
        dtype = "f32" if not split_gradk else "f32 (fp32 transform-tile convolutions; fp16x2-split MFMA PSF gradient, fp32 accumulate)"
        note = ("fp32 throughout in the two PSF convolutions: 128 x 128 overlap-save FFT tiles held in LDS, on channel-planar mirrors of the frames "
                "(ics_conv_fft.hip)%s" % ("; the PSF gradient on the matrix cores with fp16x2-split operands" if split_gradk else ""))
    elif split_conv:
        dtype = "f32 (fp16x2-split MFMA convolutions, fp32 accumulate)"
        note = ("frames, sums and every elementwise step in fp32; the two PSF convolutions%s run on the matrix cores%s with each fp32 operand split into "
                "two fp16 terms (22 significand bits), three fp16 MFMAs per product, fp32 accumulation; `--conv vector` runs fp32 products throughout"
                % (" and the PSF gradient" if split_gradk else "", " as tap blocks of <= 33 x 33" if route.conv_family == 2 else ""))
    elif split_gradk:
        dtype = "f32 (fp32-product convolutions; fp16x2-split MFMA PSF gradient, fp32 accumulate)"
        note = "fp32 products in the two PSF convolutions (packed-fp32 vector kernels); the PSF gradient on the matrix cores with fp16x2-split operands"
    elif route.conv_family == 6 and not fuse:
        dtype = "f32"
        note = ("fp32 throughout: one cooperative launch per outer iteration, a (tile, channel) of the frame per compute unit with its operands resident "
                "in LDS, fp32 FMA convolutions in a fixed order (ics_small.hip)")
    else:
        dtype = "f32"
        note = "fp32 throughout (packed-fp32 vector convolutions%s)" % (", fp32-MFMA PSF gradient" if route.gradk_family else "")
    return {"matrix": split_conv, "dtype": dtype, "dtype_note": note, "conv": conv_f if not fuse else "vector (fused update + convolution)", "gradk": gradk_f,
            "traffic_key": "kernels_fft" if (route.conv_family == 5 and not fuse) else ("kernels_matrix" if split_conv else "kernels_vector")}


def gaussian_1d(MK):
    n = np.arange(MK) - (MK - 1) / 2.0
    w = np.exp(-0.5 * (n / (MK / 6.0)) ** 2)
    return (w / w.sum()).astype(np.float32)


_FRAME_CACHE = {}


def synth_frame(M, N, MK, seed):
    """the last frame built is kept: the secondary configurations reuse one size several times, and host-side frame synthesis was most of
    the wall time of a default run (the GPU idled through it)"""
    key = (M, N, MK, seed)
    if key not in _FRAME_CACHE:
        _FRAME_CACHE.clear()
        _FRAME_CACHE[key] = _synth_frame(M, N, MK, seed)
    return _FRAME_CACHE[key]


def _synth_frame(M, N, MK, seed):
    """Synthetic problem of SURVEY.md 8d at full size, cheap enough for a benchmark prologue: smooth
    random scene on the padded frame, blurred by the separable Gaussian PSF (sigma = MK/6), + noise."""
    rng = np.random.default_rng(seed)
    pad = MK // 2
    uM, uN = M + 2 * pad, N + 2 * pad
    coarse = rng.random(((uM + 7) // 8 + 2, (uN + 7) // 8 + 2, 3), dtype=np.float32)
    sharp = np.repeat(np.repeat(coarse, 8, axis=0), 8, axis=1)[3:3 + uM, 5:5 + uN]
    for axis in (0, 1):  # 3 box passes ~ Gaussian smoothing of the 8x8 blocks
        for _ in range(2):
            sharp = (sharp + np.roll(sharp, 2, axis=axis) + np.roll(sharp, -2, axis=axis) + np.roll(sharp, 4, axis=axis)) * np.float32(0.25)
    sharp = sharp * np.float32(0.8) + np.float32(0.1)
    w = gaussian_1d(MK)
    tmp = np.zeros((M, uN, 3), np.float32)
    for p in range(MK):
        tmp += w[p] * sharp[p:p + M]
    image = np.zeros((M, N, 3), np.float32)
    for q in range(MK):
        image += w[q] * tmp[:, q:q + N]
    image += np.float32(1e-3) * rng.standard_normal(image.shape, dtype=np.float32)
    u0 = np.ascontiguousarray(np.pad(image, ((pad, pad), (pad, pad), (0, 0)), mode="edge"))
    k2 = np.outer(w, w).astype(np.float32)
    psf_true = np.ascontiguousarray(np.dstack((k2, k2, k2)))
    psf_uniform = np.full((MK, MK, 3), 1.0 / (MK * MK), np.float32)
    return image, u0, psf_true, psf_uniform


def _port_outer(orc, mode, MK, S, max_outer, budget_s):
    """whole outer iterations of the numpy / scipy port on an S x S frame: (MPix/s/iter, outer iterations, seconds)"""
    image, u0, psf_true, psf_uniform = synth_frame(S, S, MK, seed=0)
    psf = (psf_uniform if mode == "blind" else psf_true).copy()
    win = (MK // 2 + 1, 255 - MK // 2 - 1, MK // 2 + 1, 255 - MK // 2 - 1)
    outer = 0
    t0 = time.perf_counter()
    u = u0.copy()
    while True:  # chain of single outer iterations until the budget is used (at least one)
        orc.richardson_lucy_MM(image, u, psf, *win, 1e9, S, S, 3, MK, 1, 1e-3, 10000.0, blind=(mode == "blind"), quiet=True)
        outer += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or outer >= max_outer:
            break
    return S * S * 5 * outer / dt / 1e6, outer, dt, (image, u0, psf_true, psf_uniform, win)


def cpu_baseline(mode, MK, M_full, budget_s=10.0):
    """The oracle (numpy/scipy port of lib/deconvolution.pyx, same FFT call pattern as the reference: scipy.signal.convolve ->
    pocketfft, which runs on ONE thread whatever the host has) on the host cores, on a bounded sample of the METRIC's workload:
    one outer iteration (5 inner) at the frame size of `value` (~25 s blind at 4096^2) -> `value`; beside it the 2048^2 sample of
    earlier rounds, the same port with scipy's FFT on 8 workers, and -- static, labelled as such -- the compiled reference's own
    figures from the survey container (it cannot travel to the GPU box)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rl_mm_oracle as orc  # cpu_baseline leg only
    S = M_full if M_full <= 4096 else 4096
    v_full, outer, dt, pack = _port_outer(orc, mode, MK, S, 1, budget_s)
    v_2048, outer2, dt2, pack2 = (v_full, outer, dt, pack) if S == 2048 else _port_outer(orc, mode, MK, 2048, 4, budget_s * 0.5)
    image, u0, psf_true, psf_uniform, win = pack2
    # beside it (never `value`): the same port with scipy's FFT allowed up to 8 host cores (scipy.fft.set_workers) -- the reference's own
    # FFT calls are single-threaded, its elementwise loops OpenMP over all cores (lib/deconvolution.pyx:16,484); this bounds what more
    # cores could buy the FFT-dominated loop
    mt = None
    try:
        import scipy.fft
        ncpu = min(8, os.cpu_count() or 1)   # (256 workers on a 256-core host ran 6x SLOWER than one: 0.61 vs 3.5 MPix/s/iter; 8 = the survey container's count)
        u2, psf2 = u0.copy(), (psf_uniform if mode == "blind" else psf_true).copy()
        t1 = time.perf_counter()
        with scipy.fft.set_workers(ncpu):
            orc.richardson_lucy_MM(image, u2, psf2, *win, 1e9, 2048, 2048, 3, MK, 1, 1e-3, 10000.0, blind=(mode == "blind"), quiet=True)
        d2 = time.perf_counter() - t1
        mt = {"value": round(2048 * 2048 * 5 / d2 / 1e6, 4), "unit": "MPixels/s/iter", "fft_workers": ncpu, "sample": "2048^2, 1 outer (= 5 inner) iterations, %.1f s" % d2}
    except Exception as exc:   # (a baseline detail must not fail the bench line)
        mt = {"error": str(exc)[:200]}
    return {"value": round(v_full, 4), "unit": "MPixels/s/iter", "cores": 1, "kind": "port", "fft_8_workers": mt,
            "host_cores_available": os.cpu_count(),
            "threads": "1 (scipy.signal.convolve -> scipy.fft pocketfft with workers=None = single thread, numpy elementwise single thread; "
                       "OMP_NUM_THREADS=%s)" % os.environ.get("OMP_NUM_THREADS", "unset"),
            "sample": "%s, %dx%dx3, %dx%d PSF, %d outer (=%d inner) iterations of oracle/rl_mm_oracle.py "
                      "(numpy + scipy.signal.convolve FFT, single thread), %.1f s" % (mode, S, S, MK, MK, outer, 5 * outer, dt),
            "at_2048": {"value": round(v_2048, 4), "sample": "%d outer iterations at 2048^2, %.1f s (the sample of rounds 1-4)" % (outer2, dt2)},
            "reference_compiled_static": {"note": "NOT measured in this run: the compiled reference (lib/deconvolution.pyx, Cython -O3 -fopenmp, 8 OpenMP threads, scipy FFT single-threaded) "
                                                  "in the survey container, BASELINE.md section 2; it cannot travel to the GPU box",
                                          "MPixels_per_s_per_iter": {"nonblind_2048_k15": 1.80, "blind_2048_k15": 0.68, "blind_1024_k15": 0.86, "nonblind_512_k9": 2.98}}}


def mfma_counters(kernel, M, MK):
    """Matrix-pipe counters of `kernel` from the committed SQ pass of this command (profiles/rNN_mfma_counters.json, newest round first, written by
    scripts/make_mfma_json.py from rocprofv3 --pmc runs): static, NOT measured in this run -- like `traffic`."""
    try:
        for name in ("r06_mfma_counters.json", "r06_6144_31_mfma_counters.json", "r05_mfma_counters.json", "r05_6144_31_mfma_counters.json", "r04_mfma_counters.json", "r04_6144_31_mfma_counters.json"):
            fn = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(fn):
                continue
            mj = json.load(open(fn))
            if mj["workload"] == {"size": M, "psf": MK} and kernel in mj["kernels"]:
                return dict(mj["kernels"][kernel], source="profiles/%s (static: rocprofv3 SQ passes of a run of this command at commit %s)" % (name, mj.get("commit", "unrecorded")))
    except (OSError, ValueError, KeyError):
        pass
    return None


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU, RCCL rendezvous through
    a file) BEFORE this process touches HIP, relay rank 0's JSON line, fail loudly when the box has fewer than N GPUs."""
    import subprocess
    pkg = os.path.join(ROOT, "image-cases-studies_amd")
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from lib import _native; print(_native.device_count())" % pkg],
                           capture_output=True, text=True)
    try:
        ndev = int(probe.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        ndev = 0
    if ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but %d gfx950 device(s) are visible on this box; one GPU per rank is required "
                         "(no fallback to fewer GPUs)\n" % (args.gpus, ndev))
        return 2
    import multi_gpu   # (ctypes / os only: no HIP call happens in this process)
    rc, out0 = multi_gpu.launch_ranks([sys.executable, os.path.abspath(__file__)] + argv, args.gpus,
                                      timeout_s=float(os.environ.get("ICS_BENCH_TIMEOUT_S", "3600")), logdir=os.environ.get("ICS_BENCH_LOGDIR"))
    sys.stdout.write(out0)
    return rc


def conv_rel_err(ctx, MK):
    """max |residual - float64| / max |synth| of one A1 + A2 pass at 1024^2 for both convolution paths (float64 FFT convolution on
    the host): the number the fp16-split matrix-core path is judged by, measured in this run."""
    from scipy.signal import fftconvolve
    from lib import _native
    S = 1024
    image, u0, psf_true, _ = synth_frame(S, S, MK, seed=11)
    rng = np.random.default_rng(12)
    u = (u0 + np.float32(0.02) * rng.standard_normal(u0.shape, dtype=np.float32)).astype(np.float32)
    job = _native.RLJob(S, S, MK, ctx)
    job.upload(image, u, psf_true)
    synth = np.stack([fftconvolve(u[..., c].astype(np.float64), psf_true[..., c].astype(np.float64), mode="valid") for c in range(3)], -1)
    out = {}
    for name, conv in (("matrix", 2), ("vector", 1)):
        try:
            job.stage(_native.STAGE_SYNTH_RESIDUAL, job.params(1, 200, 1, 200, 1e9, 1, 1e-3, 10000.0, False, conv=conv))
            out[name] = float(np.max(np.abs(job.read(_native.BUF_ERROR) - (synth - image))) / np.max(np.abs(synth)))
        except _native.NativeError:
            out[name] = None
    job.close()
    return out


def timed_run(ctx, M, MK, blind, tv_mode, conv, steps, warm, seed=0):
    """ms per inner iteration of one configuration on a fresh job (secondary lines of the JSON)."""
    from lib import _native
    image, u0, psf_true, psf_uniform = synth_frame(M, M, MK, seed=seed)
    job = _native.RLJob(M, M, MK, ctx)
    job.upload(image, u0, psf_uniform if blind else psf_true)
    pad = MK // 2
    win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)
    route = job.describe(job.params(*win, 1e9, 1, 1e-3, 10000.0, blind, 0, 3, stop_test=2, tv_mode=tv_mode, conv=conv))
    job.run(job.params(*win, 1e9, max(1, warm // 5), 1e-3, 10000.0, blind, 0, 3, stop_test=2, tv_mode=tv_mode, conv=conv))
    ctx.synchronize()
    t0 = time.perf_counter()
    job.run(job.params(*win, 1e9, steps // 5, 1e-3, 10000.0, blind, 0, 3, stop_test=2, profile=0, tv_mode=tv_mode, conv=conv))
    ctx.synchronize()
    el = time.perf_counter() - t0
    # (kernel table from a second, event-bracketed run: the brackets cost 1-2 % of a step)
    st = job.run(job.params(*win, 1e9, max(1, steps // 10), 1e-3, 10000.0, blind, 0, 3, stop_test=2, profile=4, tv_mode=tv_mode, conv=conv))
    names = _native.KERNEL_NAMES
    kern = {names[k]: round(st.ms_kernel[k], 5) for k in range(len(names)) if st.launches[k]}
    job.close()
    mode = "blind" if blind else "nonblind"
    gb = ITER_BYTES_PER_PX[mode] * M * M / (el / steps) / 1e9
    return {"ms_per_step": round(el * 1e3 / steps, 4), "conv": _native.RLRoute.CONV_FAMILIES[route.conv_family], "MPixels_per_s_per_iter": round(M * M * steps / el / 1e6, 1),
            "algorithmic_bytes_per_px": ITER_BYTES_PER_PX[mode], "frac_of_8TBps": round(gb / HBM_PEAK_GBPS, 4), "kernels_ms": kern, "steps": steps}


def deblur_end_to_end(size=4096, blur=15, iterations=20):
    """`deblur_module` (the reference's driver, deconvolve.py:65-368: blind pass on the 255-px mask window and non-blind pass on the whole frame,
    at every pyramid level) end to end on a synthetic 8-bit picture, reference defaults otherwise: seconds per call with the frames
    resident in HBM (default) and with the frames on the host between the solver calls; the resident call's blind / non-blind split.
    The first resident call builds the jobs (allocation); the figure is the second."""
    import contextlib
    import io
    import deconvolve as dv
    rng = np.random.default_rng(0)
    coarse = rng.random((size // 8 + 2, size // 8 + 2, 3))
    pic = (np.repeat(np.repeat(coarse, 8, 0), 8, 1)[:size, :size] * 200 + 20).astype(np.uint8)
    kw = dict(mask=[size // 2, size // 2], mask_size=255, display=False, iterations=iterations, save=False)
    out = {"picture": "%dx%dx3 uint8, blur width %d, iterations=%d (outer, per solver call), mask 255 px, pyramid %s" % (size, size, blur, iterations, dv.build_pyramid(blur, 10)[1])}
    for name, dev, reps in (("frames_resident_in_hbm", True, 2), ("frames_on_host_between_solver_calls", False, 1)):
        dt = None
        for _ in range(reps):
            with contextlib.redirect_stdout(io.StringIO()):
                t = time.perf_counter()
                dv.deblur_module(pic, "t", ".", blur, device_resident=dev, **kw)
                dt = time.perf_counter() - t
        out[name] = {"seconds": round(dt, 4)}
        if dev:
            out[name]["phase_seconds"] = {k: round(v, 4) for k, v in dv.deblur_module.last_phase_seconds.items()}
    return out


def bench_bands(args, grp):
    """`--bands N`: one frame over N ranks (row bands, lib.banded.BandRank), strong scaling.  Secondary workload: the headline stays
    one frame per GPU.  Same step definition (one inner iteration), same barrier / max-over-ranks timing."""
    from lib import _native, banded
    M = N = args.size if args.size != 4096 else 12288
    MK = args.psf
    blind = args.mode == "blind"
    steps, warm = ((args.steps + 4) // 5) * 5, ((args.warmup + 4) // 5) * 5
    conv = {"auto": 0, "vector": 1, "matrix": 2, "fft": 3}[args.conv]
    pad = MK // 2
    win = (M // 2 - 127, M // 2 + 128, N // 2 - 127, N // 2 + 128)      # 255-px window at the centre: straddles the bands for even N
    image, u0, psf_true, psf_uniform = synth_frame(M, N, MK, seed=0)   # (every rank builds the same frame and takes its rows)
    br = banded.BandRank(grp, M, N, MK, *win, 1e9, 1e-3, 10000.0, blind=blind, conv=conv, device=int(os.environ.get("ICS_DEVICE", grp.local_rank)),
                         stop_test=False)
    br.upload(image, u0, psf_uniform if blind else psf_true)
    del image, u0
    if warm:
        br.run(warm // 5)
    grp.barrier()
    t0 = time.perf_counter()
    st = br.run(steps // 5)
    grp.barrier()
    elapsed = grp.max(time.perf_counter() - t0)
    rec = grp.gather([float(st.iterations_done), float(st.M_r), float(br.bd.y1 - br.bd.y0)])
    desc = grp.describe()
    br.close()
    if grp.rank == 0:
        ms = elapsed * 1e3 / steps
        gb = ITER_BYTES_PER_PX[args.mode] * M * N / (ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "MPixels/sec/iter RL-TV deconv @%d^2x3 fp32, %dx%d PSF, ONE frame over %d row bands" % (M, MK, MK, grp.size),
            "value": round(M * N * steps / elapsed / 1e6, 1), "unit": "MPixels/s/iter", "n_gpus": grp.size, "steps": steps, "warmup": warm,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 (fp16x2-split MFMA convolutions, fp32 accumulate)" if conv != 1 else "f32",
            "data": "synthetic",
            "config": {"workload": "%s Richardson-Lucy MM, ONE %dx%dx3 frame split into %d row bands (one rank per GPU), %dx%d PSF; per inner iteration: "
                                   "all-reduce(max) of 6 keys, two point-to-point halo exchanges of %d rows, %s; host-synchronous stage calls"
                                   % (args.mode, M, N, grp.size, MK, MK, 2 * pad, "all-reduce(sum) of 3 K^2 doubles" if blind else "no further exchange"),
                       "mode": args.mode, "parallelism": "row-bands x%d" % grp.size, "band_rows": [int(r[2]) for r in rec]},
            "hbm_roofline_iteration": {"algorithmic_bytes_per_px": ITER_BYTES_PER_PX[args.mode], "achieved_GBps": round(gb, 1),
                                       "frac_of_aggregate_8TBps_x_N": round(gb / (HBM_PEAK_GBPS * grp.size), 4)},
            "rccl": dict(desc, ranks_gathered=len(rec))}))
    grp.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--conv", choices=["auto", "vector", "matrix", "fft"], default="auto",
                    help="convolution kernels: auto = matrix cores (fp16-split MFMA) at every PSF size (whole PSF to 49 x 49, tap blocks above); "
                         "vector = fp32 products everywhere; the JSON's dtype / config.conv come from the library's own routing (ics_rl_describe)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--mode", choices=["blind", "nonblind"], default="blind")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--psf", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the secondary (untimed for `value`) run of the other mode")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the secondary lines for BASELINE.json configs[1], configs[3] and the TV variants")
    ap.add_argument("--tv-mode", type=int, default=0, help="0 = shipped loop (TV term dead, the parity-pinned path); 1 = build-defined active MM-TV")
    ap.add_argument("--fuse", action="store_true", help="fused update+convolution kernel (opt-in; measured slower)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--no-sustained", action="store_true", help="skip the `sustained` block (25 + 200 extra steps after the timed region)")
    ap.add_argument("--no-preheat", action="store_true", help="skip the cold-start measurement and the 150-step clock pre-heat in front of the timed region")
    ap.add_argument("--bands", type=int, default=0, help="N > 0: ONE frame (--size, default 12288 here) split into N row bands, one rank per GPU "
                    "(lib.banded.BandRank: RCCL max / sum all-reduce + point-to-point halos); strong scaling, a secondary workload (SURVEY.md 8f N4)")
    args = ap.parse_args()

    if args.bands:
        args.gpus = args.bands
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))   # (no HIP / torch call has happened in this process)

    import multi_gpu
    from lib import _native
    grp = multi_gpu.Group()
    if grp.size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, grp.size))
    if args.bands:
        raise SystemExit(bench_bands(args, grp))
    M = N = args.size
    MK = args.psf
    steps = ((args.steps + 4) // 5) * 5
    warm = ((args.warmup + 4) // 5) * 5
    blind = args.mode == "blind"
    conv = {"auto": 0, "vector": 1, "matrix": 2, "fft": 3}[args.conv]
    if conv == 0 and os.environ.get("ICS_CONV_PATH", "")[:1] == "v":
        conv = 1

    ndev = _native.device_count()
    dev = int(os.environ.get("ICS_DEVICE", grp.local_rank))
    if dev >= ndev:
        raise SystemExit("bench.py: rank %d wants device %d but only %d device(s) are visible" % (grp.rank, dev, ndev))
    ctx = _native.Context.get(dev)
    image, u0, psf_true, psf_uniform = synth_frame(M, N, MK, seed=grp.rank)
    job = _native.RLJob(M, N, MK, ctx)
    job.upload(image, u0, psf_uniform if blind else psf_true)
    pad = MK // 2
    win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)  # 255-px stats window as deconvolve.py:281 passes it

    def run(n_inner, profile):
        p = job.params(*win, 1e9, n_inner // 5, 1e-3, 10000.0, blind, 0, 3, stop_test=2, profile=profile, fuse=int(args.fuse), tv_mode=args.tv_mode, conv=conv)
        return job.run(p)

    # which kernels the run resolves to: asked of the library (ics_rl_describe), not re-derived here
    lab = labels(job.describe(job.params(*win, 1e9, 1, 1e-3, 10000.0, blind, 0, 3, stop_test=2, fuse=int(args.fuse), tv_mode=args.tv_mode, conv=conv)), args.fuse)
    matrix = lab["matrix"]

    # Order of the measurements.  (1) `cold_start`: W warm-up + K steps exactly as they come after the host-side frame synthesis (the GPU
    # has idled for seconds: the first ~50 steps run while the clocks ramp, NOTES_r03.md 4c) -- reported, not `value`.  (2) a clock pre-heat
    # of PREHEAT untimed steps on EVERY rank at EVERY N (so that N = 1 and N = 8 are measured in the same state).  (3) the contract's
    # measurement: W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides -> `value`.
    PREHEAT = 0 if args.no_preheat else 150
    cold = None
    if PREHEAT:
        if warm:
            run(warm, 0)
        ctx.synchronize()
        tc = time.perf_counter()
        run(steps, 0)
        ctx.synchronize()
        ec = time.perf_counter() - tc
        cold = {"ms_per_step": round(ec * 1e3 / steps, 4), "MPixels_per_s_per_iter": round(M * N * steps / ec / 1e6, 1), "steps": steps, "warmup": warm,
                "note": "the same W + K steps run FIRST, straight after the host-side frame synthesis (GPU idle for seconds, clocks ramping); rank-local, not `value`"}
        run(PREHEAT, 0)
    if warm:
        run(warm, 0)
    ctx.synchronize()
    grp.barrier()
    t0 = time.perf_counter()
    # HIP events around the kernels of every 4th inner iteration of the timed region: bracketing every launch
    # costs ~4 % of the step time, a sample does not; 4 is coprime with the 5 inner iterations per outer one, so
    # every position of the inner loop is sampled (the first update of an outer iteration reads u == ut and is cheaper)
    st = run(steps, 0 if args.no_profile else 4)
    ctx.synchronize()
    grp.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = grp.max(elapsed)
    assert st.inner_iterations == steps, (st.inner_iterations, steps)

    # secondary measurement (not `value`): the other mode on the same resident frame, same schedule
    other = None
    if grp.size == 1 and not args.no_other_mode:
        omode = "nonblind" if blind else "blind"
        job.upload(image, u0, psf_true if blind else psf_uniform)
        po = job.params(*win, 1e9, steps // 5, 1e-3, 10000.0, not blind, 0, 3, stop_test=2, profile=0, fuse=int(args.fuse), tv_mode=args.tv_mode, conv=conv)
        job.run(job.params(*win, 1e9, max(1, warm // 5), 1e-3, 10000.0, not blind, 0, 3, stop_test=2, conv=conv))
        ctx.synchronize()
        t1 = time.perf_counter()
        job.run(po)
        ctx.synchronize()
        e2 = time.perf_counter() - t1
        ogb = ITER_BYTES_PER_PX[omode] * M * N / (e2 / steps) / 1e9
        other = {"mode": omode, "ms_per_step": round(e2 * 1e3 / steps, 4), "MPixels_per_s_per_iter": round(M * N * steps / e2 / 1e6, 1),
                 "algorithmic_bytes_per_px": ITER_BYTES_PER_PX[omode], "frac_of_8TBps": round(ogb / HBM_PEAK_GBPS, 4)}

    # beside `value` (never instead of it): a SUSTAINED measurement -- 25 warm-up + 200 timed steps whatever --steps / --warmup were --
    # with its own kernel table.  A 20-step run after 5 warm-up steps is timed while the clocks still ramp (NOTES_r02.md section 4a);
    # this block is what the per-kernel fractions in DESIGN.md are quoted from.
    sustained = None
    if grp.size == 1 and not args.no_sustained:
        run(25, 0)
        ctx.synchronize()
        ts = time.perf_counter()
        run(200, 0)
        ctx.synchronize()
        es = time.perf_counter() - ts
        sk = run(100, 4)                      # kernel table from an event-bracketed run of its own (the brackets cost 1-2 %)
        ctx.synchronize()
        names_ = _native.KERNEL_NAMES
        sgb = ITER_BYTES_PER_PX[args.mode] * M * N / (es / 200) / 1e9
        sustained = {"steps": 200, "warmup": 25, "ms_per_step": round(es * 1e3 / 200, 4), "MPixels_per_s_per_iter": round(M * N * 200 / es / 1e6, 1),
                     "frac_of_8TBps": round(sgb / HBM_PEAK_GBPS, 4),
                     "kernels_ms": {names_[k]: round(sk.ms_kernel[k], 5) for k in range(len(names_)) if sk.launches[k]}}

    per_rank = grp.gather([st.ms_total, float(st.iterations_done), float(st.M_r), float(st.has_nan)])
    rccl = grp.describe()
    rccl["ranks_gathered"] = len(per_rank)
    rccl["hardware_note"] = ("an RCCL communicator with more than one rank has never run in this build's history: every box it was given had one GPU "
                             "(the N > 1 path is covered by gloo world-size-2 tests on CPU, tests/test_multi_gpu.py)") if grp.size == 1 else "this run is the N > 1 evidence"
    job.close()
    if grp.rank == 0:
        ms_per_step = elapsed * 1e3 / steps
        value = grp.size * M * N * steps / elapsed / 1e6
        names = _native.KERNEL_NAMES
        kern = {names[k]: {"ms": round(st.ms_kernel[k], 5), "launches": st.launches[k]} for k in range(len(names)) if st.launches[k]}
        roof = None
        traffic, traffic_file, traffic_commit = None, "profiles/r06_hbm_traffic.json", "unrecorded"
        for tf in ("profiles/r06_hbm_traffic.json", "profiles/r06_6144_31_hbm_traffic.json", "profiles/r05_hbm_traffic.json", "profiles/r05_6144_31_hbm_traffic.json", "profiles/r04_hbm_traffic.json", "profiles/r04_6144_31_hbm_traffic.json"):
            try:  # measured HBM bytes per launch (rocprofv3 PMC passes, committed under profiles/): static, NOT measured in this run
                tj = json.load(open(os.path.join(ROOT, tf)))
                if tj["workload"] == {"size": M, "psf": MK} and lab["traffic_key"] in tj:
                    traffic, traffic_file, traffic_commit = tj[lab["traffic_key"]], tf, tj.get("commit", "unrecorded")
                    break
            except (OSError, ValueError, KeyError):
                traffic = None
        if kern:
            dom = max((k for k in kern if k in BYTES_PER_PX), key=lambda k: kern[k]["ms"] * kern[k]["launches"])
            bytes_launch = BYTES_PER_PX[dom] * M * N
            ach = bytes_launch / (kern[dom]["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "limiter": ("issue / lds / matrix pipe (hbm is the yardstick)" if matrix else ("lds transfers + the CU's vector-memory path, which do not overlap (hbm is the yardstick; NOTES_r05.md)" if lab["traffic_key"] == "kernels_fft" else "fp32 valu (hbm is the yardstick)")) if dom != "update" else "hbm",
                    "bound_note": "HBM (8 TB/s) is the yardstick north_star sets and what `achieved` / `peak` / `frac` are quoted against; what LIMITS the matrix-core kernels is issue / LDS / "
                    "matrix-pipe time, not bytes (traffic < algorithmic bytes for the fused kernel; SQ and FETCH / WRITE passes under profiles/) -- see `mfma`", "kernel": dom,
                    "mfma": mfma_counters(dom, M, MK), "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBPS, 4),
                    "traffic": (traffic[dom]["hbm_bytes"] if traffic and dom in traffic else None),
                    "traffic_source": "%s (static: rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE passes of a run of this command at commit %s, not measured live)" % (traffic_file, traffic_commit),
                    "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": kern[dom]["ms"],
                    "per_kernel": {k: {"algorithmic_GBps": round(BYTES_PER_PX[k] * M * N / (kern[k]["ms"] * 1e-3) / 1e9, 1),
                                       "frac": round(BYTES_PER_PX[k] * M * N / (kern[k]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                       "traffic": (traffic[k]["hbm_bytes"] if traffic and k in traffic else None)} for k in kern if k in BYTES_PER_PX}}
        it_gbps = ITER_BYTES_PER_PX[args.mode] * M * N / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "MPixels/sec/iter RL-TV deconv @%d^2x3 fp32, %dx%d PSF" % (M, MK, MK),
            "value": round(value, 1), "unit": "MPixels/s/iter", "n_gpus": grp.size, "steps": steps, "warmup": warm,
            "ms_per_step": round(ms_per_step, 4), "cold_start_ms_per_step": cold["ms_per_step"] if cold else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": lab["dtype"], "data": "synthetic",
            "cold_start": cold,
            "order": ("cold_start (W + K steps after idle, reported beside), %d untimed pre-heat steps, then the contract's W warm-up + K timed steps = `value`" % PREHEAT) if PREHEAT else "W warm-up + K timed steps (no pre-heat)",
            "dtype_note": lab["dtype_note"],
            "config": {"workload": "%s Richardson-Lucy MM (lib/deconvolution.pyx loop as shipped: the TV term is arithmetically dead in the reference, tv_mode %d), %dx%dx3 fp32, %dx%d PSF, one frame per GPU, "
                                   "stop test evaluated every outer iteration" % ("blind" if blind else "non-blind", args.tv_mode, M, N, MK, MK),
                       "mode": args.mode, "tv_mode": args.tv_mode, "conv": lab["conv"], "psf_gradient": lab["gradk"], "step_is": "one inner iteration (5 per outer iteration)", "parallelism": "image-per-gpu x%d" % grp.size,
                       "collective": "none in the iterations; %s barrier / max / all-gather of a 4-double record per rank" % (grp.backend if grp.size > 1 else "no")},
            "hbm_roofline_iteration": {"algorithmic_bytes_per_px": ITER_BYTES_PER_PX[args.mode], "achieved_GBps": round(it_gbps, 1),
                                       "frac_of_8TBps": round(it_gbps / HBM_PEAK_GBPS, 4)},
            "kernels_ms": kern, "device_ms_total_rank0": round(st.ms_total, 3),
            "per_rank": [{"device_ms": round(r[0], 3), "outer_done": int(r[1])} for r in per_rank],
            "roofline": roof,
            "sustained": sustained,
            "rccl": rccl,
            "other_mode_same_frame": other,
        }
        if grp.size == 1 and not args.no_other_configs and args.size == 4096 and args.psf == 15 and args.tv_mode == 0 and not args.fuse:
            # secondary lines (never `value`): the other BASELINE.json configurations and the build-defined TV variants they name
            out["conv_rel_err_vs_f64"] = conv_rel_err(ctx, MK)
            # since round 6 the headline's default path is fp32 throughout (transform tiles); the same workload on the other two kernel families,
            # every round, beside it: packed-fp32 vector convolutions + fp32-MFMA gradient, and the matrix cores with fp16x2-split operands
            out["other_product_paths"] = {"fp32 products, vector (ics_conv.hip + fp32-MFMA PSF gradient)": timed_run(ctx, M, MK, blind, 0, 1, 50, 10),
                                          "fp16x2-split products on the matrix cores (ics_conv_mfma.hip, fused A11 + A13 kernel; the headline path of rounds 1-5)": timed_run(ctx, M, MK, blind, 0, 2, 50, 10),
                                          "fp32 transform tiles (ics_conv_fft.hip), forced with --conv fft": timed_run(ctx, M, MK, blind, 0, 3, 50, 10)}
            oc = {}
            oc["configs[0] non-blind 512^2 9x9 (the reference's CPU plumbing case; launch-bound on a GPU)"] = timed_run(ctx, 512, 9, False, 0, conv, 400, 50)   # (a 34-us step: 100 steps were 3 ms, a third of the call's fixed cost in the figure)
            # the reference's blind workload (deconvolve.py:277-286: a 255 x 255 window at every pyramid level): one cooperative launch per outer iteration
            # since round 6 (ics_small.hip), and the multi-launch path it replaced (matrix cores, forced with conv = 2)
            oc["blind 255^2 15x15 (the blind window of deblur_module; ICS_CONV_AUTO: the cooperative small-frame kernel, `conv`: lds-resident -- not under rocprofv3, see README)"] = timed_run(ctx, 255, 15, True, 0, conv, 400, 50)
            oc["blind 255^2 15x15 on the multi-launch path (conv = matrix)"] = timed_run(ctx, 255, 15, True, 0, 2, 400, 50)
            oc["configs[1] non-blind 2048^2 15x15 (shipped loop)"] = timed_run(ctx, 2048, 15, False, 0, conv, 100, 10)
            oc["configs[1] non-blind 2048^2 15x15 + active MM-TV (tv_mode 1, build-defined)"] = timed_run(ctx, 2048, 15, False, 1, conv, 50, 5)
            oc["configs[1] non-blind 2048^2 15x15 + PAM isotropic TV (tv_mode 2, build-defined)"] = timed_run(ctx, 2048, 15, False, 2, conv, 50, 5)
            oc["configs[2] blind 4096^2 15x15 PAM isotropic TV (tv_mode 2, build-defined)"] = timed_run(ctx, 4096, 15, True, 2, conv, 50, 5)
            oc["configs[2] blind 4096^2 15x15 PAM collaborative TV (tv_mode 3, build-defined)"] = timed_run(ctx, 4096, 15, True, 3, conv, 50, 5)
            oc["configs[3] blind 6144^2 31x31 (shipped loop)"] = timed_run(ctx, 6144, 31, True, 0, conv, 25, 5)
            oc["configs[3] blind 6144^2 31x31 PAM collaborative TV (tv_mode 3, build-defined)"] = timed_run(ctx, 6144, 31, True, 3, conv, 25, 5)
            # beyond BASELINE.json: the largest PSF of the reference's own examples (deconvolve.py:409, blur width 45)
            oc["example blind 4096^2 45x45 (shipped loop; reference deconvolve.py:409)"] = timed_run(ctx, 4096, 45, True, 0, conv, 10, 5)
            # ... and a PSF beyond 65 x 65: the tiles serve sizes to 97 since round 6 (the matrix cores' tap blocks above)
            oc["blind 4096^2 95x95 (shipped loop; the reference has no PSF size limit, pyx:378-390)"] = timed_run(ctx, 4096, 95, True, 0, conv, 10, 5)
            out["other_configs"] = oc
            out["deblur_module_end_to_end"] = {"4096^2 blur 15, reference defaults (iterations=20)": deblur_end_to_end(4096, 15, 20),
                                               "2048^2 blur 15, reference defaults (iterations=20)": deblur_end_to_end(2048, 15, 20)}
        if not args.no_cpu_baseline and grp.size == 1:   # (rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(args.mode, MK, M)
        print(json.dumps(out))
    grp.close()


if __name__ == "__main__":
    main()
