"""One image over several GPUs: row-band sharding of `richardson_lucy_MM`.

Not in the reference (a single-process Cython function, lib/deconvolution.pyx:341-675); this is the split SURVEY.md 8e
describes for frames beyond one GPU's comfort.  The image rows [0, M) are cut into contiguous bands; band b lives in its
own device job (its own GPU, or several bands on one GPU for testing) that holds its rows PLUS a halo of pad = MK // 2 image
rows on either side, i.e. u rows [a, b + 2 pad) for extended image rows [a, b).  Every kernel of an inner iteration runs on
the whole band job; only the rows the band OWNS come out exact, and three things cross bands per inner iteration:

  1. step size (pyx:523-524): max |g_k| and max u_k over the owned rows only (ICS_STAGE_BAND_REDUCE), combined over the
     bands as a max of 6 order-preserving keys and written back to every band before the update;
  2. halo exchange: after the update (pyx:527-552) every band sends its first / last 2 pad owned u rows to its neighbours
     (the back-projection of the next iteration reaches 2 pad rows beyond the owned ones);
  3. blind: the PSF gradient (pyx:567-571) is summed over the owned rows only (ICS_STAGE_BAND_MASK_E zeroes the rest of the
     residual first), the 3 MK^2 partial sums are added over the bands in float64, fixed order, and written back; the PSF
     step (pyx:574-589) then runs identically on every band.

The stop-test statistics (pyx:593-654) run on a small job of their own that receives the window's rows of the residual and of
u from the bands that own them, once per outer iteration (the window may straddle bands).  Transfers stay on the
devices: rows are copied device to device (`ics_rl_copy_rows`: peer access over xGMI between GPUs, a plain device copy when
bands share a GPU) and fall back to the host where two devices cannot reach each other; the 6 reduction keys and the 3 MK^2
gradient sums (a few KB) go through the host.  Band jobs run concurrently, one host thread per band (ctypes releases the GIL
inside the C calls).  With the fp32 convolution kernels (`conv=1`) a non-blind
banded run is bit-identical to the single-job run; the matrix-core kernels scale per tile, so results agree to ~1e-7.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native
from .deconvolution import _check_buffer, _progress, _report

INNER = 5   # pyx:375


class _Band:
    def __init__(self, index, y0, y1, M, pad, device):
        self.index, self.y0, self.y1 = index, y0, y1                  # owned image rows [y0, y1)
        self.a, self.b = max(0, y0 - pad), min(M, y1 + pad)           # extended image rows
        self.first, self.last = y0 == 0, y1 == M
        # owned u rows, global and job-local: image row i <-> u row i + pad; the first / last band also own the border ring
        self.u0 = 0 if self.first else y0 + pad
        self.u1 = M + 2 * pad if self.last else y1 + pad
        self.lu0, self.lu1 = self.u0 - self.a, self.u1 - self.a
        self.device = device
        self.job = None


def split_rows(M, bands, pad):
    """[y0, y1) per band: contiguous, balanced; every band must be at least 2 pad rows high (halo exchange)."""
    edges = [M * k // bands for k in range(bands + 1)]
    out = [(edges[k], edges[k + 1]) for k in range(bands)]
    if any(y1 - y0 < max(2 * pad, 1) for y0, y1 in out):
        raise ValueError("%d rows over %d bands: a band must hold at least 2 * (MK // 2) = %d rows" % (M, bands, 2 * pad))
    return out


def richardson_lucy_MM_banded(image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations, step_factor, lambd,
                              blind=True, correlation=False, p=1., norm=1, order=2, priority=0, refocus=0, *, bands=2, devices=None, conv=0):
    """`richardson_lucy_MM` (same arguments, same in-place semantics, same printed lines) with the image split into `bands`
    row bands; `devices` = one device index per band (default: band k on device k mod device_count)."""
    _check_buffer("image", image); _check_buffer("u", u); _check_buffer("psf", psf)
    M, N, MK = int(M), int(N), int(MK)
    pad = MK // 2
    if u.shape != (M + 2 * pad, N + 2 * pad, 3) or image.shape != (M, N, 3) or psf.shape != (MK, MK, 3):
        raise ValueError("expected image (%d,%d,3), u (%d,%d,3), psf (%d,%d,3)" % (M, N, M + 2 * pad, N + 2 * pad, MK, MK))
    ndev = max(1, _native.device_count())
    devices = list(devices) if devices is not None else [k % ndev for k in range(bands)]
    B = [_Band(k, y0, y1, M, pad, devices[k]) for k, (y0, y1) in enumerate(split_rows(M, bands, pad))]
    if not (0 <= top < bottom <= M):
        raise ValueError("stats window rows [%d, %d) outside the %d image rows" % (top, bottom, M))
    image_c = np.ascontiguousarray(image, dtype=np.float32)
    u_c = np.ascontiguousarray(u, dtype=np.float32)
    psf_c = np.ascontiguousarray(psf, dtype=np.float32)
    nv = _native
    # the statistics job holds image rows [top, bottom) at the frame's full width: it is a job like any other and has the same 2 GiB frame
    # limit.  Checked before anything is allocated (the library would refuse it with ICS_ENOSUP after the band jobs were built).
    if not (0 < nv.frame_bytes(bottom - top, N, MK) < nv.FRAME_LIMIT_BYTES):
        raise ValueError("stop-test window rows [%d, %d) of a %d-px-wide frame: its statistics job would hold %d bytes per frame buffer, the limit is %d "
                         "(choose a window of fewer rows; the reference's drivers pass the 255-px mask window, deconvolve.py:277-313)"
                         % (top, bottom, N, nv.frame_bytes(bottom - top, N, MK), nv.FRAME_LIMIT_BYTES))
    pool = ThreadPoolExecutor(max_workers=len(B))
    sj = None

    def par(fn):
        return list(pool.map(fn, B))

    def setup(bd):
        bd.job = nv.RLJob(bd.b - bd.a, N, MK, nv.Context.get(bd.device))
        bd.job.upload(image_c[bd.a:bd.b], u_c[bd.a:bd.b + 2 * pad], psf_c)
        bd.P = lambda **kw: bd.job.params(top - bd.a, bottom - bd.a, left, right, tau, 1, step_factor, lambd, blind, correlation, channels=C,
                                          conv=conv, flags=nv.FLAG_NO_FUSED_GRADK, **kw)
    sP = None

    # Rows move between jobs device to device (`ics_rl_copy_rows`: same GPU, or peer access over xGMI); if two devices cannot
    # reach each other the library says ICS_ENOSUP once and every transfer goes through the host from then on.
    d2d = [True]

    def move(dst, which, drow, src, srow, n):
        if d2d[0]:
            try:
                dst.copy_rows_from(which, drow, src, which, srow, n)
                return
            except nv.NativeError as e:
                if e.code != nv.ICS_ENOSUP:
                    raise
                d2d[0] = False
        dst.write_rows(which, drow, src.read_rows(which, srow, n))

    def gather(which, g0, g1):
        """global rows [g0, g1) of a frame buffer, from the bands that own them, into the statistics job (u-frame rows for BUF_U,
        image rows for BUF_ERROR)"""
        for bd in B:
            o0, o1 = (bd.u0, bd.u1) if which == nv.BUF_U else (bd.y0, bd.y1)
            lo, hi = max(g0, o0), min(g1, o1)
            if lo < hi:
                move(sj, which, lo - g0, bd.job, lo - bd.a, hi - lo)

    def key2f(k):
        k = np.uint32(k)
        return (np.uint32(k & np.uint32(0x7FFFFFFF)) if k & np.uint32(0x80000000) else np.uint32(~k)).view(np.float32)

    def statistics():
        gather(nv.BUF_ERROR, top, bottom)
        gather(nv.BUF_U, top, bottom + 2 * pad)
        sj.stage(nv.STAGE_STATS, sP)
        out = sj.scalars()
        # DoF extrema of the last update (printed diagnostics, pyx:593): over the rows of every band job, halos included
        keys = np.stack(par(lambda bd: bd.job.red_keys()))
        nan = bool(keys[:, 14].any())
        out["dof_min"] = float("nan") if nan else float(key2f(keys[:, 12].min()))
        out["dof_max"] = float("nan") if nan else float(key2f(keys[:, 13].max()))
        return out

    st = nv.RLStats.with_traces(iterations)
    sc = {"Hu": float("nan"), "varu": float("nan")}
    it, stop = 0, 0
    M_r = M_r_prev = 0.0
    try:
        par(setup)
        # the statistics job: image rows [top, bottom) of the frame (u rows [top, bottom + 2 pad))
        sj = nv.RLJob(bottom - top, N, MK, nv.Context.get(devices[0]))
        sP = sj.params(0, bottom - top, left, right, tau, 1, step_factor, lambd, blind, correlation, channels=C, conv=conv)
        while it < iterations and not stop:                                               # pyx:460
            par(lambda bd: bd.job.stage(nv.STAGE_MAJORIZE, bd.P()))                       # pyx:462
            for itt in range(INNER):                                                      # pyx:473
                par(lambda bd: bd.job.stage(nv.STAGE_SYNTH_RESIDUAL, bd.P()))             # A1 + A2
                par(lambda bd: bd.job.stage(nv.STAGE_BACKPROJECT, bd.P()))                # A3
                par(lambda bd: bd.job.stage(nv.STAGE_BAND_REDUCE, bd.P(band_rows=(bd.lu0, bd.lu1))))
                keys = np.max(np.stack(par(lambda bd: bd.job.red_keys())), axis=0)        # (1) max over the bands
                par(lambda bd: bd.job.set_red_keys(keys))
                par(lambda bd: bd.job.stage(nv.STAGE_UPDATE, bd.P()))                     # A5 - A10
                # (2) halo exchange of the updated u: 2 pad owned rows each way
                if len(B) > 1:
                    def halo(bd):   # only halo rows are written, only owned rows are read: the copies of all bands commute
                        if not bd.first:
                            prev = B[bd.index - 1]
                            move(bd.job, nv.BUF_U, bd.lu0 - 2 * pad, prev.job, prev.lu1 - 2 * pad, 2 * pad)   # rows above my owned rows
                        if not bd.last:
                            nxt = B[bd.index + 1]
                            move(bd.job, nv.BUF_U, bd.lu1, nxt.job, nxt.lu0, 2 * pad)                      # rows below them
                    par(halo)
                if blind:                                                                 # pyx:555
                    par(lambda bd: bd.job.stage(nv.STAGE_SYNTH_RESIDUAL, bd.P()))         # A11
                    if itt == INNER - 1:
                        sc = statistics()                                                 # A18 + A19 need the unmasked residual
                    par(lambda bd: bd.job.stage(nv.STAGE_BAND_MASK_E, bd.P(band_rows=(bd.y0 - bd.a, bd.y1 - bd.a))))
                    par(lambda bd: bd.job.stage(nv.STAGE_PSF_GRADIENT, bd.P()))           # A12 + A13 over the owned rows
                    gk = np.zeros((MK, MK, 3), np.float64)
                    for g in par(lambda bd: bd.job.read(nv.BUF_GRADK)):                   # (3) fixed order
                        gk += g
                    gk32 = gk.astype(np.float32)
                    par(lambda bd: bd.job.write(nv.BUF_GRADK, gk32))
                    par(lambda bd: bd.job.stage(nv.STAGE_PSF_UPDATE, bd.P()))             # A14 - A17
                elif itt == INNER - 1:
                    sc = statistics()
            if it > 0:
                M_r_prev = M_r
            M_r = sc["M_r"]
            st.trace_M_r[it], st.trace_Hu[it], st.trace_varu[it] = sc["M_r"], sc["Hu"], sc["varu"]
            st.trace_dof_min[it], st.trace_dof_max[it] = sc["dof_min"], sc["dof_max"]
            st.trace_len = it + 1
            if it > 1:                                                                    # pyx:643-654
                if blind:
                    stop = int(M_r > M_r_prev)
                else:
                    stop = int((M_r - M_r_prev) / (M_r + M_r_prev) > tau)
            it += 1
            _progress(it, stop, sc["dof_min"], sc["dof_max"], sc["M_r"], sc["Hu"], sc["varu"])  # pyx:593,648,658-659
        # gather: every band's owned rows -> the caller's u (in place, pyx:675), PSF of band 0
        rows = par(lambda bd: bd.job.read_rows(nv.BUF_U, bd.lu0, bd.lu1 - bd.lu0))
        for bd, r in zip(B, rows):
            u[bd.u0:bd.u1] = r
        if blind:
            psf[...] = B[0].job.download_psf_caller()
        st.iterations_done, st.stopped = it, stop
        st.M_r, st.Hu, st.varu = M_r, sc["Hu"], sc["varu"]
        st.has_nan = int(np.isnan(u).any())
        st.inner_iterations = INNER * it
    finally:
        for bd in B:
            if bd.job is not None:
                bd.job.close()
        if sj is not None:
            sj.close()
        pool.shutdown()
    _report(st, top, bottom, left, right, lambd)
    richardson_lucy_MM_banded.last = st
    return u[pad:pad + M, pad:pad + N, ...]


richardson_lucy_MM_banded.last = None


class BandRank:
    """One rank's share of a row-band split with ONE PROCESS PER BAND (one rank per GPU; `group` = multi_gpu.Group): rank r owns
    band r of `group.size` bands.  What crosses bands goes through the group -- RCCL over xGMI on a multi-GPU node, the CPU
    stand-in where ranks share a GPU:
      1. the six step-size keys: `group.reduce_band_keys` (ics_rl_allreduce_keys: ncclMax on the uint32 keys, in place on the device);
      2. the halo rows: `group.exchange_rows` (ncclSend / ncclRecv between the band jobs' device frames), both directions;
      3. blind: the 3 MK^2 PSF-gradient partial sums: `group.reduce_band_gradk` (ics_rl_allreduce_gradk: one float64 ncclSum on the device);
      4. the stop-test window: its rows travel to rank 0, which holds the statistics job; the scalars come back with `gather`.
    upload() once, run() as often as wanted on the resident frames (bench.py --bands times run() alone), download() the owned rows."""

    def __init__(self, group, M, N, MK, top, bottom, left, right, tau, step_factor, lambd, blind=True, correlation=False, C=3, conv=0,
                 device=None, stop_test=True):
        nv = _native
        self.group, self.M, self.N, self.MK = group, int(M), int(N), int(MK)
        self.pad = pad = self.MK // 2
        self.R, self.W = group.rank, group.size
        rows = split_rows(self.M, self.W, pad)
        self.bd = bd = _Band(self.R, rows[self.R][0], rows[self.R][1], self.M, pad, device if device is not None else nv.default_device())
        self.bands = [_Band(k, rows[k][0], rows[k][1], self.M, pad, 0) for k in range(self.W)]
        if not (0 <= top < bottom <= self.M):
            raise ValueError("stats window rows [%d, %d) outside the %d image rows" % (top, bottom, self.M))
        self.win, self.tau, self.blind, self.stop_test = (top, bottom, left, right), tau, bool(blind), stop_test
        if not (0 < nv.frame_bytes(bottom - top, self.N, self.MK) < nv.FRAME_LIMIT_BYTES):   # (every rank: the same answer everywhere, before anything is built)
            raise ValueError("stop-test window rows [%d, %d) of a %d-px-wide frame: its statistics job would exceed the %d-byte frame limit" % (top, bottom, self.N, nv.FRAME_LIMIT_BYTES))
        ctx = nv.Context.get(bd.device)
        self.job = nv.RLJob(bd.b - bd.a, self.N, self.MK, ctx)
        self.sj = nv.RLJob(bottom - top, self.N, self.MK, ctx) if self.R == 0 else None
        # stages are queued, not waited for (ICS_FLAG_STAGE_ASYNC): the collectives and the row exchanges run on the job's stream, the host
        # waits where it reads -- the statistics' scalars, once per outer iteration (ICS_BAND_ASYNC=0: a synchronisation per stage, as before round 5)
        import os
        flags = nv.FLAG_NO_FUSED_GRADK | (nv.FLAG_STAGE_ASYNC if os.environ.get("ICS_BAND_ASYNC", "1") != "0" else 0)
        self.fused_ok = MK <= 15 and conv in (nv.CONV_AUTO, nv.CONV_MATRIX)               # ICS_STAGE_SYNTH_GRADK exists (matrix-core path)
        self._P = lambda **kw: self.job.params(top - bd.a, bottom - bd.a, left, right, tau, 1, step_factor, lambd, blind, correlation, channels=C,
                                               conv=conv, flags=flags, **kw)
        self._sP = self.sj.params(0, bottom - top, left, right, tau, 1, step_factor, lambd, blind, correlation, channels=C, conv=conv) if self.sj else None

    def upload(self, image, u, psf):
        """every rank passes the full arrays and takes its own rows of them"""
        bd, pad = self.bd, self.pad
        self.job.upload(np.ascontiguousarray(image[bd.a:bd.b], np.float32), np.ascontiguousarray(u[bd.a:bd.b + 2 * pad], np.float32),
                        np.ascontiguousarray(psf, np.float32))

    def download(self, u, psf=None):
        """the rows this rank OWNS into u[u0:u1] (u-frame rows); returns (u0, u1)"""
        bd = self.bd
        u[bd.u0:bd.u1] = self.job.read_rows(_native.BUF_U, bd.lu0, bd.lu1 - bd.lu0)
        if psf is not None and self.blind:
            psf[...] = self.job.download_psf_caller()
        return bd.u0, bd.u1

    def close(self):
        self.job.close()
        if self.sj is not None:
            self.sj.close()

    def _gather_window(self, which, g0, g1):
        """global rows [g0, g1) of a frame buffer -> the statistics job on rank 0, from the ranks that own them, in rank order"""
        nv, R, bd, sj, job = _native, self.R, self.bd, self.sj, self.job
        for k, nb in enumerate(self.bands):
            o0, o1 = (nb.u0, nb.u1) if which == nv.BUF_U else (nb.y0, nb.y1)
            lo, hi = max(g0, o0), min(g1, o1)
            if lo >= hi:
                continue
            if k == 0 and R == 0:
                sj.copy_rows_from(which, lo - g0, job, which, lo - bd.a, hi - lo)
            elif R == 0:
                self.group.exchange_rows(sj, which, recv=(lo - g0, hi - lo, k))
            elif R == k:
                self.group.exchange_rows(job, which, send=(lo - bd.a, hi - lo, 0))

    def _statistics(self):
        nv = _native
        top, bottom = self.win[0], self.win[1]
        self._gather_window(nv.BUF_ERROR, top, bottom)
        self._gather_window(nv.BUF_U, top, bottom + 2 * self.pad)
        sc = [0.0, 0.0, 0.0]
        if self.R == 0:
            self.sj.stage(nv.STAGE_STATS, self._sP)
            d = self.sj.scalars()
            sc = [d["M_r"], d["Hu"], d["varu"]]
        keys = self.job.red_keys()
        rec = self.group.gather(sc + [float(keys[12]), float(keys[13]), float(keys[14])])
        kmin, kmax = min(int(r[3]) for r in rec), max(int(r[4]) for r in rec)
        nan = any(r[5] for r in rec)

        def key2f(k):
            k = np.uint32(k)
            return float((np.uint32(k & np.uint32(0x7FFFFFFF)) if k & np.uint32(0x80000000) else np.uint32(~k)).view(np.float32))
        return dict(M_r=np.float32(rec[0][0]), Hu=np.float32(rec[0][1]), varu=np.float32(rec[0][2]),
                    dof_min=float("nan") if nan else key2f(kmin), dof_max=float("nan") if nan else key2f(kmax))

    def run(self, iterations, quiet=True):
        nv, job, bd, group, R, pad, blind, P = _native, self.job, self.bd, self.group, self.R, self.pad, self.blind, self._P
        st = nv.RLStats.with_traces(iterations)
        sc = {"Hu": float("nan"), "varu": float("nan"), "M_r": 0.0, "dof_min": 0.0, "dof_max": 0.0}
        it, stop = 0, 0
        M_r = M_r_prev = np.float32(0.0)
        one = self.W == 1                                                                 # a single band: no halos, nothing to combine
        fused = one and blind and self.fused_ok
        while it < iterations and not stop:                                               # pyx:460
            job.stage(nv.STAGE_MAJORIZE, P())                                             # pyx:462
            for itt in range(INNER):                                                      # pyx:473
                job.stage(nv.STAGE_SYNTH_RESIDUAL, P())                                   # A1 + A2
                job.stage(nv.STAGE_BACKPROJECT, P())                                      # A3
                if not one:                                                               # (a single band owns every row: the back-projection's own maxima)
                    job.stage(nv.STAGE_BAND_REDUCE, P(band_rows=(bd.lu0, bd.lu1)))
                    group.reduce_band_keys(job)                                           # (1) in place on the device (RCCL), no host hop
                job.stage(nv.STAGE_UPDATE, P())                                           # A5 - A10
                # (2) my first 2 pad owned rows go up, the rows below my owned ones arrive from below; then the other way round
                up = None if bd.first else (bd.lu0, 2 * pad, R - 1)
                dn = None if bd.last else (bd.lu1 - 2 * pad, 2 * pad, R + 1)
                if self.W > 1:
                    group.exchange_rows(job, nv.BUF_U, send=up, recv=None if bd.last else (bd.lu1, 2 * pad, R + 1))
                    group.exchange_rows(job, nv.BUF_U, send=dn, recv=None if bd.first else (bd.lu0 - 2 * pad, 2 * pad, R - 1))
                if blind and fused:                                                       # one band, PSF <= 15: A11 + A12 + A13 in the fused kernel
                    job.stage(nv.STAGE_SYNTH_GRADK, P())
                    if itt == INNER - 1:
                        sc = self._statistics()
                    job.stage(nv.STAGE_PSF_UPDATE, P())                                   # A14 - A17
                elif blind:                                                               # pyx:555
                    job.stage(nv.STAGE_SYNTH_RESIDUAL, P())                               # A11
                    if itt == INNER - 1:
                        sc = self._statistics()                                           # A18 + A19 need the unmasked residual
                    if not one:
                        job.stage(nv.STAGE_BAND_MASK_E, P(band_rows=(bd.y0 - bd.a, bd.y1 - bd.a)))
                    job.stage(nv.STAGE_PSF_GRADIENT, P())                                 # A12 + A13 over the owned rows
                    if not one:
                        group.reduce_band_gradk(job)                                      # (3) one float64 all-reduce on the device
                    job.stage(nv.STAGE_PSF_UPDATE, P())                                   # A14 - A17
                elif itt == INNER - 1:
                    sc = self._statistics()
            if it > 0:
                M_r_prev = M_r
            M_r = np.float32(sc["M_r"])
            st.trace_M_r[it], st.trace_Hu[it], st.trace_varu[it] = sc["M_r"], sc["Hu"], sc["varu"]
            st.trace_dof_min[it], st.trace_dof_max[it] = sc["dof_min"], sc["dof_max"]
            st.trace_len = it + 1
            if it > 1 and self.stop_test:                                                 # pyx:643-654 (the same float32 scalars on every rank)
                if blind:
                    stop = int(M_r > M_r_prev)
                else:
                    with np.errstate(divide="ignore", invalid="ignore"):
                        stop = int(np.float32(M_r - M_r_prev) / np.float32(M_r + M_r_prev) > np.float32(self.tau))
            it += 1
            if R == 0 and not quiet:
                _progress(it, stop, sc["dof_min"], sc["dof_max"], sc["M_r"], sc["Hu"], sc["varu"])
        st.iterations_done, st.stopped = it, stop
        st.M_r, st.Hu, st.varu = float(M_r), float(sc["Hu"]), float(sc["varu"])
        st.inner_iterations = INNER * it
        return st


def richardson_lucy_MM_band_rank(group, image, u, psf, top, bottom, left, right, tau, M, N, C, MK, iterations, step_factor, lambd,
                                 blind=True, correlation=False, *, conv=0, device=None, quiet=False):
    """`richardson_lucy_MM` for ONE RANK of a row-band split (see BandRank): every rank passes the full arrays; `u` receives the rows
    this rank OWNS (u-frame rows [u0, u1)) and `psf` the PSF -- assembling a full frame is the caller's business (bench.py --bands
    does not need it, tests/test_banded.py gathers through files).  Returns (u0, u1, stats)."""
    _check_buffer("image", image); _check_buffer("u", u); _check_buffer("psf", psf)
    br = BandRank(group, M, N, MK, top, bottom, left, right, tau, step_factor, lambd, blind=blind, correlation=correlation, C=C, conv=conv, device=device)
    try:
        br.upload(image, u, psf)
        st = br.run(iterations, quiet=quiet)
        u0, u1 = br.download(u, psf)
        st.has_nan = int(np.isnan(u[u0:u1]).any())
    finally:
        br.close()
    if group.rank == 0 and not quiet:
        _report(st, top, bottom, left, right, lambd)
    return u0, u1, st
