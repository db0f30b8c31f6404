"""ctypes binding of libics_hip.so (C ABI declared in include/ics_hip.h).

This is the only place the product touches native code.  There is deliberately no CPU fallback:
if the shared library is missing, or no gfx950 device is usable, importing succeeds (so that the
symbol-level tests can run on a GPU-less machine) but every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICS_HIP_LIB", os.path.join(os.path.dirname(_HERE), "libics_hip.so"))

ICS_ABI_VERSION = 4
ICS_KERNEL_COUNT = 12
KERNEL_NAMES = ("synth_residual", "backproject", "update", "psf_gradient", "psf_update", "majorize", "stats", "update_synth",
                "synth_gradk", "synth_backproject", "small_iteration", "_11")

# error codes (include/ics_hip.h)
ICS_OK, ICS_EINVAL, ICS_ENODEV, ICS_EHIP, ICS_ENOMEM, ICS_ESTATE, ICS_ENOSUP = 0, -1, -2, -3, -4, -5, -6

# stages / buffers
STAGE_SYNTH_RESIDUAL, STAGE_BACKPROJECT, STAGE_UPDATE, STAGE_PSF_GRADIENT, STAGE_PSF_UPDATE, STAGE_MAJORIZE, STAGE_STATS, STAGE_UPDATE_SYNTH, STAGE_TVTERM, STAGE_SYNTH_GRADK, STAGE_BAND_REDUCE, STAGE_BAND_MASK_E, STAGE_SYNTH_BACKPROJECT = range(1, 14)
FLAG_NO_FUSED_GRADK = 1   # ics_rl_params.flags (include/ics_hip.h ICS_FLAG_*)
FLAG_STAGE_ASYNC = 2      # ics_rl_stage returns once the stage is queued
BUF_U, BUF_UT, BUF_GRADU, BUF_IMAGE, BUF_ERROR, BUF_PSF, BUF_GRADK, BUF_SCALARS, BUF_TV, BUF_RED = range(10)
CONV_AUTO, CONV_VECTOR, CONV_MATRIX, CONV_FFT = range(4)   # ics_rl_params.conv (include/ics_hip.h ICS_CONV_*)
SCALAR_NAMES = ("dt0", "dt1", "dt2", "maxu0", "maxu1", "maxu2", "maxg0", "maxg1", "maxg2", "dtpsf", "M_r", "Hu", "varu",
                "dof_min", "dof_max", "_")


# void (*ics_rl_progress_fn)(void *user, int it, int stopped, float dof_min, float dof_max, float M_r, float Hu, float varu)
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float)


class RLParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("top", C.c_int), ("bottom", C.c_int), ("left", C.c_int), ("right", C.c_int),
                ("tau", C.c_float), ("iterations", C.c_int), ("step_factor", C.c_float), ("lambd", C.c_float),
                ("blind", C.c_int), ("correlation", C.c_int), ("channels", C.c_int), ("tv_mode", C.c_int),
                ("stop_test", C.c_int), ("profile", C.c_int), ("fuse", C.c_int), ("conv", C.c_int), ("flags", C.c_int), ("band_row0", C.c_int), ("band_row1", C.c_int),
                ("progress", PROGRESS_FN), ("progress_user", C.c_void_p)]


class RLStats(C.Structure):
    """struct ics_rl_stats.  The per-outer-iteration traces live in caller-owned arrays (`with_traces(n)` allocates them as
    numpy arrays and keeps them alive on the object: st.trace_M_r[:st.trace_len] etc. then read as before)."""
    _fields_ = [("struct_size", C.c_uint32), ("iterations_done", C.c_int), ("stopped", C.c_int), ("has_nan", C.c_int),
                ("M_r", C.c_float), ("Hu", C.c_float), ("varu", C.c_float),
                ("dof_min", C.c_float), ("dof_max", C.c_float), ("trace_len", C.c_int), ("trace_cap", C.c_int),
                ("_p_M_r", C.POINTER(C.c_float)), ("_p_Hu", C.POINTER(C.c_float)), ("_p_varu", C.POINTER(C.c_float)),
                ("_p_dof_min", C.POINTER(C.c_float)), ("_p_dof_max", C.POINTER(C.c_float)),
                ("ms_total", C.c_float), ("inner_iterations", C.c_int),
                ("ms_kernel", C.c_float * ICS_KERNEL_COUNT), ("launches", C.c_int * ICS_KERNEL_COUNT)]
    TRACES = ("M_r", "Hu", "varu", "dof_min", "dof_max")

    @classmethod
    def with_traces(cls, cap):
        st = cls()
        st.struct_size = C.sizeof(cls)
        st.trace_cap = max(0, int(cap))
        for name in cls.TRACES:
            arr = np.zeros(max(1, st.trace_cap), np.float32)
            setattr(st, "trace_" + name, arr)                       # plain attribute: keeps the buffer alive
            setattr(st, "_p_" + name, arr.ctypes.data_as(C.POINTER(C.c_float)))
        return st


class RLRoute(C.Structure):
    """struct ics_rl_route (ics_rl_describe): which kernel families a run with these parameters launches."""
    _fields_ = [("struct_size", C.c_uint32), ("conv_family", C.c_int), ("conv_fp16_split", C.c_int), ("gradk_family", C.c_int),
                ("gradk_fp16_split", C.c_int), ("image_in_accumulator_order", C.c_int), ("graph", C.c_int)]
    CONV_FAMILIES = {1: "matrix", 2: "matrix-blocks", 3: "vector", 4: "vector-big", 5: "fft-tiles", 6: "lds-resident"}
    GRADK_FAMILIES = {0: "none", 1: "fused-matrix", 2: "matrix", 3: "matrix-blocks", 4: "fp32-mfma", 5: "fp32-big", 6: "fft-tiles", 7: "fused-fft-tiles", 8: "lds-resident"}


def describe(M, N, MK, params):
    """ics_describe: the kernel families a run of an M x N frame with an MK x MK PSF and these parameters launches (no device needed)."""
    r = RLRoute()
    r.struct_size = C.sizeof(RLRoute)
    _check(load().ics_describe(int(M), int(N), int(MK), C.byref(params), C.byref(r)))
    return r


FRAME_LIMIT_BYTES = 1 << 31      # include/ics_hip.h ICS_FRAME_LIMIT_BYTES


def frame_bytes(M, N, MK):
    """ics_rl_frame_bytes: device bytes of one frame buffer of such a job, 0 for an invalid shape (no device needed)."""
    return int(load().ics_rl_frame_bytes(int(M), int(N), int(MK)))


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libics_hip error %d: %s" % (code, message))
        self.code = code


_lib = None


def _fp(dtype):
    return np.ctypeslib.ndpointer(dtype=dtype, flags="C_CONTIGUOUS")


def load():
    """dlopen libics_hip.so and declare the prototypes.  Raises ImportError loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError("libics_hip.so not found at %s -- build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                          "(or `make -C image-cases-studies_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ci, cf, cd = C.c_void_p, C.c_int, C.c_float, C.c_double
    lib.ics_abi_version.restype = ci
    lib.ics_last_error.restype = C.c_char_p
    lib.ics_device_count.argtypes = [C.POINTER(ci)]
    lib.ics_ctx_create.argtypes = [ci, C.POINTER(vp)]
    lib.ics_ctx_destroy.argtypes = [vp]; lib.ics_ctx_destroy.restype = None
    lib.ics_ctx_synchronize.argtypes = [vp]
    lib.ics_ctx_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(ci), C.POINTER(C.c_uint64)]
    lib.ics_ctx_last_kernel_ms.argtypes = [vp, C.POINTER(cf)]; lib.ics_ctx_last_kernel_ms.restype = ci
    lib.ics_rl_create.argtypes = [vp, ci, ci, ci, C.POINTER(vp)]
    lib.ics_rl_destroy.argtypes = [vp]; lib.ics_rl_destroy.restype = None
    lib.ics_rl_upload.argtypes = [vp, vp, vp, vp]
    lib.ics_rl_download.argtypes = [vp, vp, vp, vp]
    lib.ics_rl_run.argtypes = [vp, C.POINTER(RLParams), C.POINTER(RLStats)]
    lib.ics_rl_stage.argtypes = [vp, ci, C.POINTER(RLParams)]
    lib.ics_rl_describe.argtypes = [vp, C.POINTER(RLParams), C.POINTER(RLRoute)]; lib.ics_rl_describe.restype = ci
    lib.ics_describe.argtypes = [ci, ci, ci, C.POINTER(RLParams), C.POINTER(RLRoute)]; lib.ics_describe.restype = ci
    lib.ics_rl_frame_bytes.argtypes = [ci, ci, ci]; lib.ics_rl_frame_bytes.restype = C.c_ulonglong
    lib.ics_rl_read.argtypes = [vp, ci, vp, C.c_size_t]
    lib.ics_rl_write.argtypes = [vp, ci, vp, C.c_size_t]
    lib.ics_rl_read_rows.argtypes = [vp, ci, ci, ci, vp]
    lib.ics_rl_write_rows.argtypes = [vp, ci, ci, ci, vp]
    lib.ics_rl_copy_rows.argtypes = [vp, ci, ci, vp, ci, ci, ci]
    lib.ics_normalize_kernel.argtypes = [vp, vp, ci]
    lib.ics_tv.argtypes = [vp, vp, ci, ci, cf, ci, ci, vp, vp]
    lib.ics_conv2d_symm.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp]
    lib.ics_usm.argtypes = [vp, vp, ci, ci, vp, ci, ci, cd, vp]
    lib.ics_bilateral.argtypes = [vp, vp, ci, ci, ci, cd, cd, vp]
    lib.ics_resize_bicubic.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci]
    lib.ics_img_create.argtypes = [vp, ci, ci, C.POINTER(vp)]
    lib.ics_img_destroy.argtypes = [vp]; lib.ics_img_destroy.restype = None
    lib.ics_img_shape.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    lib.ics_img_upload.argtypes = [vp, vp]
    lib.ics_img_upload_int.argtypes = [vp, vp, ci]
    lib.ics_img_download.argtypes = [vp, vp]
    lib.ics_img_pad_edge.argtypes = [vp, ci, ci, ci, ci, C.POINTER(vp)]
    lib.ics_img_crop.argtypes = [vp, ci, ci, ci, ci, C.POINTER(vp)]
    lib.ics_img_paste.argtypes = [vp, ci, ci, vp]
    lib.ics_img_gamma.argtypes = [vp, cf, cf, cf, ci]
    lib.ics_img_resize.argtypes = [vp, ci, ci, C.POINTER(vp)]
    lib.ics_rl_upload_img.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp]
    lib.ics_rl_download_img.argtypes = [vp, vp, ci, ci]
    lib.ics_group_create.argtypes = [ci, ci, ci, C.c_char_p, ci, C.POINTER(vp)]
    lib.ics_group_destroy.argtypes = [vp]; lib.ics_group_destroy.restype = None
    lib.ics_group_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    lib.ics_group_barrier.argtypes = [vp]
    lib.ics_group_allreduce_max.argtypes = [vp, vp, ci]
    lib.ics_group_allreduce_sum.argtypes = [vp, vp, ci]
    lib.ics_group_describe.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.c_char_p, C.c_size_t]
    lib.ics_rl_exchange_rows.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci]; lib.ics_rl_exchange_rows.restype = ci
    lib.ics_rl_allreduce_keys.argtypes = [vp, vp]; lib.ics_rl_allreduce_keys.restype = ci
    lib.ics_rl_allreduce_gradk.argtypes = [vp, vp]; lib.ics_rl_allreduce_gradk.restype = ci
    lib.ics_rl_params_size.restype = C.c_size_t
    lib.ics_rl_stats_size.restype = C.c_size_t
    lib.ics_debug_set.argtypes = [C.c_char_p, ci]; lib.ics_debug_set.restype = ci      # csrc/ics_common.h IcsDebug, not in the public header
    lib.ics_debug_get.argtypes = [C.c_char_p, C.POINTER(ci)]; lib.ics_debug_get.restype = ci
    lib.ics_group_allgather.argtypes = [vp, vp, ci, vp]
    for name in ("ics_device_count", "ics_ctx_create", "ics_ctx_synchronize", "ics_ctx_info", "ics_rl_create", "ics_rl_upload",
                 "ics_rl_download", "ics_rl_run", "ics_rl_stage", "ics_rl_read", "ics_rl_write", "ics_rl_read_rows", "ics_rl_write_rows", "ics_rl_copy_rows", "ics_normalize_kernel",
                 "ics_tv", "ics_conv2d_symm", "ics_usm", "ics_bilateral", "ics_resize_bicubic", "ics_img_create", "ics_img_shape",
                 "ics_img_upload", "ics_img_upload_int", "ics_img_download", "ics_img_pad_edge", "ics_img_crop", "ics_img_paste", "ics_img_gamma", "ics_img_resize",
                 "ics_rl_upload_img", "ics_rl_download_img", "ics_group_create", "ics_group_info", "ics_group_barrier",
                 "ics_group_allreduce_max", "ics_group_allreduce_sum", "ics_group_describe", "ics_group_allgather"):
        getattr(lib, name).restype = ci
    if lib.ics_abi_version() != ICS_ABI_VERSION:
        raise ImportError("libics_hip.so ABI version %d, expected %d" % (lib.ics_abi_version(), ICS_ABI_VERSION))
    if lib.ics_rl_params_size() != C.sizeof(RLParams) or lib.ics_rl_stats_size() != C.sizeof(RLStats):
        raise ImportError("libics_hip.so struct sizes (%d, %d) differ from this binding's (%d, %d)" % (
            lib.ics_rl_params_size(), lib.ics_rl_stats_size(), C.sizeof(RLParams), C.sizeof(RLStats)))
    _lib = lib
    return lib


def _check(rc):
    if rc != ICS_OK:
        raise NativeError(rc, load().ics_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def debug_set(name, value):
    """test / measurement switches of the library (csrc/ics_common.h IcsDebug; e.g. "max_wgs", "dynamic_tiles", "conv_rs"):
    read from the environment once at first use, changed at run time here.  Returns the previous value."""
    lib = load()
    old = C.c_int(0)
    if lib.ics_debug_get(name.encode(), C.byref(old)) != 0 or lib.ics_debug_set(name.encode(), int(value)) != 0:
        raise KeyError(name)
    return old.value


def device_count():
    n = C.c_int(0)
    rc = load().ics_device_count(C.byref(n))
    return n.value if rc == ICS_OK else 0


def default_device():
    """One process per GPU: LOCAL_RANK picks the device (torchrun / bench.py), ICS_DEVICE overrides."""
    return int(os.environ.get("ICS_DEVICE", os.environ.get("LOCAL_RANK", "0")))


class Context:
    """ics_ctx: one HIP stream on one gfx950 device."""
    _cache = {}

    def __init__(self, device=None):
        lib = load()
        self.device = default_device() if device is None else int(device)
        h = C.c_void_p()
        _check(lib.ics_ctx_create(self.device, C.byref(h)))
        self._h = h
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        hbm = C.c_uint64(0)
        _check(lib.ics_ctx_info(h, name, 256, C.byref(cus), C.byref(hbm)))
        self.name, self.compute_units, self.hbm_bytes = name.value.decode(), cus.value, hbm.value

    @classmethod
    def get(cls, device=None):
        device = default_device() if device is None else int(device)
        if device not in cls._cache:
            cls._cache[device] = cls(device)
        return cls._cache[device]

    def synchronize(self):
        _check(load().ics_ctx_synchronize(self._h))

    def last_kernel_ms(self):
        """device time of the kernels of the last blur / USM / bilateral call (transfers excluded)"""
        ms = C.c_float(0)
        _check(load().ics_ctx_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def close(self):
        """Destroys the stream (ics_ctx_destroy).  Jobs and images of this context must be closed first."""
        if getattr(self, "_h", None):
            load().ics_ctx_destroy(self._h)
            self._h = None
            if Context._cache.get(self.device) is self:
                del Context._cache[self.device]

    def __del__(self):
        try:
            if Context._cache.get(self.device) is not self:    # cached contexts live as long as the process
                self.close()
        except Exception:
            pass

    # ---- standalone operators ---------------------------------------------------------------
    def normalize_kernel(self, kern, MK):
        _check(load().ics_normalize_kernel(self._h, _ptr(kern), int(MK)))

    def tv(self, u, epsilon, order, norm):
        u = np.ascontiguousarray(u, dtype=np.float32)
        out, div = np.empty_like(u), np.empty_like(u)
        _check(load().ics_tv(self._h, _ptr(u), u.shape[0], u.shape[1], float(epsilon), int(order), int(norm), _ptr(out), _ptr(div)))
        return out, div

    def conv2d_symm(self, src, kern):
        src = np.ascontiguousarray(src, dtype=np.float64)
        kern = np.ascontiguousarray(kern, dtype=np.float64)
        out = np.empty_like(src)
        _check(load().ics_conv2d_symm(self._h, _ptr(src), src.shape[0], src.shape[1], _ptr(kern), kern.shape[0], kern.shape[1], _ptr(out)))
        return out

    def usm(self, src, kern, amount):
        src = np.ascontiguousarray(src, dtype=np.float64)
        kern = np.ascontiguousarray(kern, dtype=np.float64)
        out = np.empty_like(src)
        _check(load().ics_usm(self._h, _ptr(src), src.shape[0], src.shape[1], _ptr(kern), kern.shape[0], kern.shape[1], float(amount), _ptr(out)))
        return out

    def bilateral(self, src, radius, std_i, std_s):
        src = np.ascontiguousarray(src, dtype=np.float64)
        out = np.empty_like(src)
        _check(load().ics_bilateral(self._h, _ptr(src), src.shape[0], src.shape[1], int(radius), float(std_i), float(std_s), _ptr(out)))
        return out


    def resize_bicubic(self, img, shape):
        """deconvolve.py:245-249 (skimage.transform.resize order=3, mode="edge") on the device; H x W x C float64 in and out."""
        img = np.ascontiguousarray(img, dtype=np.float64)
        if img.ndim == 2:
            return self.resize_bicubic(img[..., None], shape)[..., 0]
        oh, ow = int(shape[0]), int(shape[1])
        out = np.empty((oh, ow, img.shape[2]), np.float64)
        _check(load().ics_resize_bicubic(self._h, _ptr(img), img.shape[0], img.shape[1], img.shape[2], _ptr(out), oh, ow))
        return out


class DeviceImage:
    """ics_img: an H x W x 3 float32 image that lives in HBM (the frames `deconvolve.deblur_module` carries between two
    richardson_lucy_MM calls).  Methods that return an image create a new one; `gamma` and `paste` work in place."""

    def __init__(self, handle, ctx):
        self._h, self.ctx = handle, ctx

    @classmethod
    def from_host(cls, arr, ctx=None):
        ctx = ctx or Context.get()
        arr = np.asarray(arr)
        as_int = arr.dtype in (np.uint8, np.uint16)      # pixels as read from a file: converted on the device (exact), 1 / 2 bytes per value over PCIe
        arr = np.ascontiguousarray(arr) if as_int else np.ascontiguousarray(arr, dtype=np.float32)
        if arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError("DeviceImage needs an H x W x 3 array")
        h = C.c_void_p()
        _check(load().ics_img_create(ctx._h, arr.shape[0], arr.shape[1], C.byref(h)))
        img = cls(h, ctx)
        if as_int:
            _check(load().ics_img_upload_int(img._h, arr.ctypes.data_as(C.c_void_p), arr.dtype.itemsize))
        else:
            _check(load().ics_img_upload(img._h, _ptr(arr)))
        return img

    @property
    def shape(self):
        H, W = C.c_int(), C.c_int()
        _check(load().ics_img_shape(self._h, C.byref(H), C.byref(W)))
        return (H.value, W.value, 3)

    def to_host(self):
        out = np.empty(self.shape, np.float32)
        _check(load().ics_img_download(self._h, _ptr(out)))
        return out

    def _new(self, fn, *args):
        h = C.c_void_p()
        _check(fn(self._h, *args, C.byref(h)))
        return DeviceImage(h, self.ctx)

    def pad_edge(self, top, bottom, left, right):
        return self._new(load().ics_img_pad_edge, int(top), int(bottom), int(left), int(right))

    def crop(self, y0, y1, x0, x1):
        """self[y0:y1, x0:x1]"""
        return self._new(load().ics_img_crop, int(y0), int(x0), int(y1 - y0), int(x1 - x0))

    def copy(self):
        H, W, _ = self.shape
        return self.crop(0, H, 0, W)

    def resize(self, oh, ow):
        return self._new(load().ics_img_resize, int(oh), int(ow))

    def paste(self, y0, x0, src):
        _check(load().ics_img_paste(self._h, int(y0), int(x0), src._h))

    def gamma(self, div, exponent, mul=1.0, clip01=False):
        _check(load().ics_img_gamma(self._h, float(div), float(exponent), float(mul), int(bool(clip01))))

    def close(self):
        if self._h:
            load().ics_img_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RLJob:
    """ics_rl: device-resident frames of one richardson_lucy_MM problem (M x N x 3, MK x MK x 3)."""

    def __init__(self, M, N, MK, ctx=None):
        self.ctx = ctx or Context.get()
        self.M, self.N, self.MK = int(M), int(N), int(MK)
        self.pad = self.MK // 2
        self.uM, self.uN = self.M + 2 * self.pad, self.N + 2 * self.pad
        h = C.c_void_p()
        _check(load().ics_rl_create(self.ctx._h, self.M, self.N, self.MK, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            load().ics_rl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, which):
        if which in (BUF_U, BUF_UT, BUF_GRADU, BUF_TV):
            return (self.uM, self.uN, 3)
        if which in (BUF_IMAGE, BUF_ERROR):
            return (self.M, self.N, 3)
        if which in (BUF_PSF, BUF_GRADK):
            return (self.MK, self.MK, 3)
        return (16,)   # BUF_SCALARS, BUF_RED

    def upload(self, image, u, psf):
        image = np.ascontiguousarray(image, dtype=np.float32)
        u = np.ascontiguousarray(u, dtype=np.float32)
        psf = np.ascontiguousarray(psf, dtype=np.float32)
        assert image.shape == (self.M, self.N, 3) and u.shape == (self.uM, self.uN, 3) and psf.shape == (self.MK, self.MK, 3), \
            (image.shape, u.shape, psf.shape)
        _check(load().ics_rl_upload(self._h, _ptr(image), _ptr(u), _ptr(psf)))

    def upload_img(self, image, image_origin, u, u_origin, psf):
        """frames from windows of device images (ics_rl_upload_img)"""
        psf = np.ascontiguousarray(psf, dtype=np.float32)
        assert psf.shape == (self.MK, self.MK, 3), psf.shape
        _check(load().ics_rl_upload_img(self._h, image._h, int(image_origin[0]), int(image_origin[1]), u._h, int(u_origin[0]), int(u_origin[1]), _ptr(psf)))

    def download_img(self, u, u_origin):
        _check(load().ics_rl_download_img(self._h, u._h, int(u_origin[0]), int(u_origin[1])))

    def download_psf_caller(self):
        psf_caller = np.empty((self.MK, self.MK, 3), np.float32)
        _check(load().ics_rl_download(self._h, None, None, _ptr(psf_caller)))
        return psf_caller

    def download(self, u_out=None):
        """(u, psf_local, psf_caller).  `u_out`: a C-contiguous float32 array of u's shape to download into (the caller's own `u`: saves a
        frame-sized host copy, 10 ms at 4096^2); anything else gets a fresh array."""
        if u_out is not None and isinstance(u_out, np.ndarray) and u_out.dtype == np.float32 and u_out.shape == (self.uM, self.uN, 3) and u_out.flags["C_CONTIGUOUS"] and u_out.flags["WRITEABLE"]:
            u = u_out
        else:
            u = np.empty((self.uM, self.uN, 3), np.float32)
        psf_local = np.empty((self.MK, self.MK, 3), np.float32)
        psf_caller = np.empty((self.MK, self.MK, 3), np.float32)
        _check(load().ics_rl_download(self._h, _ptr(u), _ptr(psf_local), _ptr(psf_caller)))
        return u, psf_local, psf_caller

    def read(self, which):
        out = np.empty(self._shape(which), np.float32)
        _check(load().ics_rl_read(self._h, which, _ptr(out), out.size))
        return out

    def write(self, which, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        assert arr.shape == self._shape(which), (arr.shape, self._shape(which))
        _check(load().ics_rl_write(self._h, which, _ptr(arr), arr.size))

    def read_rows(self, which, row0, nrows):
        cols = self._shape(which)[1]
        out = np.empty((int(nrows), cols, 3), np.float32)
        _check(load().ics_rl_read_rows(self._h, which, int(row0), int(nrows), _ptr(out)))
        return out

    def write_rows(self, which, row0, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        assert arr.ndim == 3 and arr.shape[1:] == self._shape(which)[1:], (arr.shape, self._shape(which))
        _check(load().ics_rl_write_rows(self._h, which, int(row0), arr.shape[0], _ptr(arr)))

    def copy_rows_from(self, which, row0, src, src_which, src_row0, nrows):
        """rows [src_row0, + nrows) of a frame buffer of job `src` -> rows [row0, + nrows) of this job's buffer, device to device"""
        _check(load().ics_rl_copy_rows(self._h, which, int(row0), src._h, src_which, int(src_row0), int(nrows)))

    def red_keys(self):
        """reduction keys of the last stage call: [0..2] max|g_k|, [3..5] max u_k as order-preserving uint32 keys"""
        return self.read(BUF_RED).view(np.uint32)

    def set_red_keys(self, keys):
        k = np.zeros(16, np.uint32); k[:len(keys)] = keys
        _check(load().ics_rl_write(self._h, BUF_RED, _ptr(k.view(np.float32)), 16))

    def scalars(self):
        return dict(zip(SCALAR_NAMES, self.read(BUF_SCALARS).tolist()))

    @staticmethod
    def params(top, bottom, left, right, tau, iterations, step_factor, lambd, blind, correlation=0, channels=3,
               stop_test=1, profile=0, fuse=0, tv_mode=0, conv=0, flags=0, band_rows=(0, 0)):
        p = RLParams()
        p.struct_size = C.sizeof(RLParams)
        p.top, p.bottom, p.left, p.right = int(top), int(bottom), int(left), int(right)
        p.tau, p.iterations, p.step_factor, p.lambd = float(tau), int(iterations), float(step_factor), float(lambd)
        p.blind, p.correlation, p.channels, p.tv_mode = int(bool(blind)), int(bool(correlation)), int(channels), int(tv_mode)
        p.stop_test, p.profile, p.fuse, p.conv, p.flags = int(stop_test), int(profile), int(fuse), int(conv), int(flags)
        p.band_row0, p.band_row1 = int(band_rows[0]), int(band_rows[1])
        return p

    def describe(self, params):
        """ics_rl_describe: the kernel families `run(params)` will launch (RLRoute)."""
        r = RLRoute()
        r.struct_size = C.sizeof(RLRoute)
        _check(load().ics_rl_describe(self._h, C.byref(params), C.byref(r)))
        return r

    def run(self, params, progress=None):
        """ics_rl_run.  `progress(it, stopped, dof_min, dof_max, M_r, Hu, varu)`, if given, is called after every outer iteration;
        a true return value stops the run after that iteration (stats.stopped = 2).  An exception raised inside it -- a
        KeyboardInterrupt while a progress line is printed, for instance -- also stops the run there (ctypes would otherwise
        swallow it and the device loop would run to the end); it is re-raised from here once ics_rl_run has returned, with the
        statistics of the interrupted run attached as `.ics_stats`, so that the caller can still fetch the partial result the
        way deconvolve.py:338-342 keeps it.  The caller's `params` struct is left as it was (its own callback included)."""
        st = RLStats.with_traces(params.iterations)
        if progress is None:
            _check(load().ics_rl_run(self._h, C.byref(params), C.byref(st)))
            return st
        raised = []

        def trampoline(_user, it, stopped, dmin, dmax, mr, hu, varu):
            try:
                return 1 if progress(it, stopped, dmin, dmax, mr, hu, varu) else 0
            except BaseException as e:      # noqa: BLE001 -- KeyboardInterrupt included, on purpose
                raised.append(e)
                return 1
        cb = PROGRESS_FN(trampoline)
        mine = RLParams.from_buffer_copy(params)     # run on a copy: the caller's struct keeps its own progress / progress_user
        mine.progress = cb
        mine.progress_user = None
        _check(load().ics_rl_run(self._h, C.byref(mine), C.byref(st)))
        if raised:
            raised[0].ics_stats = st
            raise raised[0]
        return st

    def stage(self, stage, params):
        _check(load().ics_rl_stage(self._h, int(stage), C.byref(params)))
