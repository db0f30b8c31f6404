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
                "update_synth": 72.0}  # fused update + convolution: 4 reads + 2 writes
ITER_BYTES_PER_PX = {"nonblind": 144.0, "blind": 204.0}


def gaussian_1d(MK):
    n = np.arange(MK) - (MK - 1) / 2.0
    w = np.exp(-0.5 * (n / (MK / 6.0)) ** 2)
    return (w / w.sum()).astype(np.float32)


def synth_frame(M, N, MK, seed):
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


def cpu_baseline(mode, MK, budget_s=20.0):
    """The oracle (numpy/scipy port of lib/deconvolution.pyx, same FFT call pattern) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rl_mm_oracle as orc  # cpu_baseline leg only
    S = 2048
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
        if dt > budget_s * 0.5 or outer >= 8:
            break
    inner = 5 * outer
    return {"value": round(S * S * inner / dt / 1e6, 4), "unit": "MPixels/s/iter", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": "%s, %dx%dx3, %dx%d PSF, %d outer (=%d inner) iterations of oracle/rl_mm_oracle.py "
                      "(numpy + scipy.signal.convolve FFT, single thread), %.1f s" % (mode, S, S, MK, MK, outer, inner, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--conv", choices=["auto", "vector", "matrix"], default="auto",
                    help="convolution kernels: auto = matrix-core (fp16-split MFMA) where built and faster (PSF <= 17, 23..37), else packed-fp32 vector")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--mode", choices=["blind", "nonblind"], default="blind")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--psf", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the secondary (untimed for `value`) run of the other mode")
    ap.add_argument("--tv-mode", type=int, default=0, help="0 = shipped loop (TV term dead, the parity-pinned path); 1 = build-defined active MM-TV")
    ap.add_argument("--fuse", action="store_true", help="fused update+convolution kernel (opt-in; measured slower)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    args = ap.parse_args()

    import multi_gpu
    from lib import _native
    grp = multi_gpu.Group()
    if grp.size != args.gpus and grp.size > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, grp.size))
    M = N = args.size
    MK = args.psf
    steps = ((args.steps + 4) // 5) * 5
    warm = ((args.warmup + 4) // 5) * 5
    blind = args.mode == "blind"
    conv = {"auto": 0, "vector": 1, "matrix": 2}[args.conv]
    if conv == 0 and os.environ.get("ICS_CONV_PATH", "")[:1] == "v":
        conv = 1
    # which convolution kernels the run resolves to (include/ics_hip.h ICS_CONV_*, csrc ics_conv_mfma_preferred)
    matrix = (conv == 2 and MK <= 37) or (conv == 0 and (MK <= 17 or 23 <= MK <= 37))

    ndev = max(1, _native.device_count())
    ctx = _native.Context.get(grp.local_rank % ndev)  # (% ndev only matters when ranks share a GPU in tests)
    image, u0, psf_true, psf_uniform = synth_frame(M, N, MK, seed=grp.rank)
    job = _native.RLJob(M, N, MK, ctx)
    job.upload(image, u0, psf_uniform if blind else psf_true)
    pad = MK // 2
    win = (pad + 1, 255 - pad - 1, pad + 1, 255 - pad - 1)  # 255-px stats window as deconvolve.py:281 passes it

    def run(n_inner, profile):
        p = job.params(*win, 1e9, n_inner // 5, 1e-3, 10000.0, blind, 0, 3, stop_test=2, profile=profile, fuse=int(args.fuse), tv_mode=args.tv_mode, conv=conv)
        return job.run(p)

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

    per_rank = grp.gather([st.ms_total, float(st.iterations_done), float(st.M_r), float(st.has_nan)])
    if grp.rank == 0:
        ms_per_step = elapsed * 1e3 / steps
        value = grp.size * M * N * steps / elapsed / 1e6
        names = _native.KERNEL_NAMES
        kern = {names[k]: {"ms": round(st.ms_kernel[k], 5), "launches": st.launches[k]} for k in range(len(names)) if st.launches[k]}
        roof = None
        traffic = None
        try:  # measured HBM bytes per launch (rocprofv3 PMC passes, committed under profiles/)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            if tj["workload"] == {"size": M, "psf": MK}:
                traffic = tj["kernels_matrix" if matrix and not args.fuse else "kernels_vector"]
        except (OSError, ValueError, KeyError):
            traffic = None
        if kern:
            dom = max((k for k in kern if k in BYTES_PER_PX), key=lambda k: kern[k]["ms"] * kern[k]["launches"])
            bytes_launch = BYTES_PER_PX[dom] * M * N
            ach = bytes_launch / (kern[dom]["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBPS, 4),
                    "traffic": (traffic[dom]["hbm_bytes"] if traffic and dom in traffic else None),
                    "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": kern[dom]["ms"]}
        it_gbps = ITER_BYTES_PER_PX[args.mode] * M * N / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "MPixels/sec/iter RL-TV deconv @%d^2x3 fp32, %dx%d PSF" % (M, MK, MK),
            "value": round(value, 1), "unit": "MPixels/s/iter", "n_gpus": grp.size, "steps": steps, "warmup": warm,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": ("frames, sums and every elementwise step in fp32; the two PSF convolutions run on the matrix cores with each "
                           "fp32 operand split into two fp16 terms (22 significand bits), three fp16 MFMAs per product, fp32 accumulation"
                           if matrix and not args.fuse else "fp32 throughout (packed-fp32 vector convolutions)"),
            "config": {"workload": "%s Richardson-Lucy MM (lib/deconvolution.pyx loop), %dx%dx3 fp32, %dx%d PSF, one frame per GPU, "
                                   "stop test evaluated every outer iteration" % ("blind" if blind else "non-blind", M, N, MK, MK),
                       "mode": args.mode, "tv_mode": args.tv_mode, "conv": "matrix" if matrix and not args.fuse else "vector", "step_is": "one inner iteration (5 per outer iteration)", "parallelism": "image-per-gpu x%d" % grp.size},
            "hbm_roofline_iteration": {"algorithmic_bytes_per_px": ITER_BYTES_PER_PX[args.mode], "achieved_GBps": round(it_gbps, 1),
                                       "frac_of_8TBps": round(it_gbps / HBM_PEAK_GBPS, 4)},
            "kernels_ms": kern, "device_ms_total_rank0": round(st.ms_total, 3),
            "per_rank": [{"device_ms": round(r[0], 3), "outer_done": int(r[1])} for r in per_rank],
            "roofline": roof,
            "other_mode_same_frame": other,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.mode, MK)
        print(json.dumps(out))
    job.close()
    grp.close()


if __name__ == "__main__":
    main()
