"""ctypes binding of include/aptgpu.h with the reference's names (see package docstring)."""
import ctypes as C
import os
import sys
import subprocess
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

FINAL_RATE = 4160    # decode.rs:14
PX_PER_ROW = 2080    # decode.rs:35
CARRIER_FREQ = 2400  # decode.rs:38

MODE_STRICT = 0
MODE_GENERIC = 1
MODE_FP16_TAPS = 2
MODE_FAST = 3

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, in-tree.  No environment variable redirects this module to another shared object; timing tools
# that want the probe build (`make -C noaa_apt_amd/csrc probe-lib`) say so in code, through use_library().
_LIB = os.path.join(_HERE, "libaptgpu.so")
_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_i8p = C.POINTER(C.c_int8)
_u8p = C.POINTER(C.c_uint8)
_ERRCAP = 1024


# ------------------------------------------------------------------ errors (err.rs:9-44)
class AptError(Exception):
    code = -1


class InternalError(AptError):       # err::Error::Internal
    code = 1


class RateOverflowError(AptError):   # err::Error::RateOverflow
    code = 2


class HipError(AptError):            # device/runtime failure
    code = 3


class InvalidError(AptError):        # FFI misuse
    code = 4


class WavOpenError(AptError):        # err::Error::WavOpen
    code = 6


class IoError(AptError):             # err::Error::Io
    code = 7


class UnsupportedError(AptError):
    code = 5


_ERRORS = {c.code: c for c in (InternalError, RateOverflowError, HipError, InvalidError,
                               UnsupportedError, WavOpenError, IoError)}


def _check(rc, err=None):
    if rc != 0:
        msg = err.value.decode("utf-8", "replace") if err is not None else ""
        raise _ERRORS.get(rc, AptError)(msg or f"aptgpu status {rc}")


# ------------------------------------------------------------------ C structs
class _CFilter(C.Structure):
    _fields_ = [("kind", C.c_int32), ("cutout_pi_rad", C.c_float), ("atten", C.c_float),
                ("delta_w_pi_rad", C.c_float)]


class _CSettings(C.Structure):
    _fields_ = [("work_rate", C.c_uint32), ("resample_atten", C.c_float),
                ("resample_delta_freq", C.c_float), ("resample_cutout", C.c_float),
                ("demodulation_atten", C.c_float), ("export_wav", C.c_int32),
                ("export_resample_filtered", C.c_int32)]


_STATUS_FN = C.CFUNCTYPE(None, C.c_float, C.c_char_p, C.c_void_p)
_STEP_FN = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_int, _f32p, C.c_size_t, C.c_uint32, C.c_void_p)


class _CContext(C.Structure):
    _fields_ = [("status", _STATUS_FN), ("step", _STEP_FN), ("user", C.c_void_p),
                ("device", C.c_int32), ("mode", C.c_int32), ("stream", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("work_len", C.c_uint64), ("n_sync", C.c_uint64), ("n_rows", C.c_uint64),
                ("l", C.c_uint32), ("m", C.c_uint32), ("n_resample_taps", C.c_uint32),
                ("n_lowpass_taps", C.c_uint32), ("fused", C.c_int32), ("orbit_path", C.c_int32)]


class PlanInfo(C.Structure):
    _fields_ = [("l", C.c_uint32), ("m", C.c_uint32), ("n_resample_taps", C.c_uint32),
                ("n_lowpass_taps", C.c_uint32), ("n_sync_taps", C.c_uint32),
                ("samples_per_work_row", C.c_uint32), ("min_distance", C.c_uint32),
                ("max_samples", C.c_uint64), ("max_work_len", C.c_uint64),
                ("max_rows", C.c_uint64), ("fused", C.c_int32), ("max_batch", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("reason", C.c_int32), ("n_rows", C.c_uint32),
                ("n_sync", C.c_uint32), ("work_len", C.c_uint64), ("n_out", C.c_uint64)]


class ImageResult(C.Structure):
    """aptgpu_image_result: contrast limits, image height and the telemetry values."""
    _fields_ = [("status", C.c_int32), ("reason", C.c_int32), ("height", C.c_uint32),
                ("telemetry_row", C.c_uint32), ("low", C.c_float), ("high", C.c_float),
                ("telemetry_quality", C.c_float), ("channel_a", C.c_int32), ("channel_b", C.c_int32),
                ("reserved", C.c_uint32), ("n_px", C.c_uint64), ("values_a", C.c_float * 16),
                ("values_b", C.c_float * 16)]


class WavSpec(C.Structure):
    """aptgpu_wav_spec: hound::WavSpec plus where the samples are."""
    _fields_ = [("channels", C.c_uint16), ("bits_per_sample", C.c_uint16),
                ("bytes_per_sample", C.c_uint16), ("sample_format", C.c_uint16),
                ("sample_rate", C.c_uint32), ("codec", C.c_int32), ("data_offset", C.c_uint64),
                ("data_len", C.c_uint64), ("n_samples", C.c_uint64), ("n_frames", C.c_uint64)]


class BatchStats(C.Structure):
    """aptgpu_batch_stats: what a host-fed batch moved and how long it took."""
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32),
                ("seconds", C.c_double), ("samples", C.c_uint64), ("h2d_bytes", C.c_uint64),
                ("d2h_bytes", C.c_uint64), ("h2d_seconds", C.c_double), ("d2h_seconds", C.c_double),
                ("workers", C.c_int32), ("recordings_per_call", C.c_int32),
                ("gate_wait_seconds", C.c_double), ("setup_seconds", C.c_double),
                ("sessions_created", C.c_int32), ("workers_pinned", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("avg_ms", C.c_double), ("launches", C.c_uint64)]


# ------------------------------------------------------------------ library loading
_lib = None


def lib_path():
    return _LIB


def use_library(path):
    """tools/ only: load `path` (the probe build, libaptgpu_probe.so — same ABI, APTGPU_DEBUG_* switches that leave
    kernels out, so its rows can be garbage) instead of the product library.  Must be called before the first lib();
    says so on stderr."""
    global _LIB
    if _lib is not None:
        raise RuntimeError("use_library() after the library was loaded")
    _LIB = os.path.abspath(path)
    sys.stderr.write(f"aptgpu: loading {_LIB} instead of the product library (timing experiments only)\n")


def build():
    """Compile libaptgpu.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j8", "all"])


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so whose SONAME
    (libamdhip64.so.7) equals /opt/rocm's; whichever is loaded first satisfies the other's
    NEEDED entry only in one direction (torch asks for "libamdhip64.so").  Loading torch's
    copy first makes libaptgpu.so and torch share it, so torch tensors' device pointers and
    streams are valid in our launches regardless of import order.  Without torch installed,
    libaptgpu.so's RUNPATH finds /opt/rocm's runtime as usual."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded libaptgpu.so.  Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    if not os.path.exists(_LIB):
        raise ImportError(f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; "
                          f"g.build()'` (or `make -C noaa_apt_amd/csrc`) first")
    L = C.CDLL(_LIB)
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    L.aptgpu_version.restype = C.c_char_p
    L.aptgpu_abi_version.restype = C.c_int
    L.aptgpu_device_count.restype = i32
    L.aptgpu_free.argtypes = [vp]
    L.aptgpu_free.restype = None
    L.aptgpu_cache_clear.argtypes = []
    L.aptgpu_cache_clear.restype = None
    L.aptgpu_cache_info.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    L.aptgpu_cache_info.restype = None
    L.aptgpu_decode.argtypes = [C.POINTER(_CContext), C.POINTER(_CSettings), _f32p, sz, u32, i32,
                                C.POINTER(_f32p), C.POINTER(sz), C.POINTER(Stats), C.c_char_p, sz]
    L.aptgpu_plan_create.argtypes = [C.POINTER(_CContext), C.POINTER(_CSettings), u32, i32, sz,
                                     i32, C.POINTER(vp), C.c_char_p, sz]
    L.aptgpu_plan_destroy.argtypes = [vp]
    L.aptgpu_plan_destroy.restype = None
    L.aptgpu_plan_get_info.argtypes = [vp, C.POINTER(PlanInfo)]
    L.aptgpu_plan_decode_device.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp),
                                            C.POINTER(sz), C.c_char_p, sz]
    L.aptgpu_plan_results.argtypes = [vp, i32, C.POINTER(Result)]
    L.aptgpu_plan_sync_positions.argtypes = [vp, i32, _u64p, sz, C.POINTER(sz)]
    L.aptgpu_plan_synchronize.argtypes = [vp]
    L.aptgpu_plan_join.argtypes = [vp]
    L.aptgpu_plan_read_internal.argtypes = [vp, i32, C.c_char_p, vp, sz, C.POINTER(sz)]
    L.aptgpu_plan_enable_timing.argtypes = [vp, i32]
    L.aptgpu_plan_collect_timing.argtypes = [vp, C.POINTER(KernelTime), sz, C.POINTER(sz)]
    L.aptgpu_filter_design.argtypes = [C.POINTER(_CFilter), C.POINTER(_f32p), C.POINTER(sz)]
    L.aptgpu_filter_resample.argtypes = [C.POINTER(_CFilter), u32, u32]
    L.aptgpu_filter_resample.restype = None
    L.aptgpu_generate_sync_frame.argtypes = [u32, C.POINTER(_i8p), C.POINTER(sz), C.c_char_p, sz]
    L.aptgpu_resample_with_filter.argtypes = [C.POINTER(_CContext), _f32p, sz, u32, u32, _CFilter,
                                              C.POINTER(_f32p), C.POINTER(sz), C.c_char_p, sz]
    L.aptgpu_resample.argtypes = [C.POINTER(_CContext), _f32p, sz, u32, u32, C.c_float, C.c_float,
                                  C.POINTER(_f32p), C.POINTER(sz), C.c_char_p, sz]
    L.aptgpu_demodulate.argtypes = [C.POINTER(_CContext), _f32p, sz, C.c_float, C.POINTER(_f32p),
                                    C.c_char_p, sz]
    L.aptgpu_filter_signal.argtypes = [C.POINTER(_CContext), _f32p, sz, _CFilter, C.POINTER(_f32p),
                                       C.c_char_p, sz]
    L.aptgpu_find_sync.argtypes = [C.POINTER(_CContext), _f32p, sz, u32, C.POINTER(_u64p),
                                   C.POINTER(sz), C.POINTER(_f32p), C.POINTER(sz), C.c_char_p, sz]
    cp, f = C.POINTER(_CContext), C.c_float
    wsp = C.POINTER(WavSpec)
    L.aptgpu_wav_parse.argtypes = [C.c_char_p, sz, wsp, C.c_char_p, sz]
    L.aptgpu_load_wav.argtypes = [cp, C.c_char_p, sz, C.POINTER(_f32p), C.POINTER(sz), C.POINTER(u32),
                                  wsp, C.c_char_p, sz]
    L.aptgpu_load_wav_file.argtypes = [cp, C.c_char_p, C.POINTER(_f32p), C.POINTER(sz), C.POINTER(u32),
                                       wsp, C.c_char_p, sz]
    L.aptgpu_decode_wav.argtypes = [cp, C.POINTER(_CSettings), C.c_char_p, sz, i32, C.POINTER(_f32p),
                                    C.POINTER(sz), C.POINTER(Stats), C.POINTER(u32), C.c_char_p, sz]
    L.aptgpu_write_wav_i16.argtypes = [cp, _f32p, sz, u32, C.POINTER(vp), C.POINTER(sz), C.c_char_p, sz]
    L.aptgpu_resample_wav.argtypes = [cp, C.c_char_p, sz, u32, f, f, C.c_char_p, C.POINTER(vp),
                                      C.POINTER(sz), C.c_char_p, sz]
    L.aptgpu_resample_wav_file.argtypes = [cp, C.c_char_p, C.c_char_p, u32, f, f, C.c_char_p, sz]
    L.aptgpu_resample_wav_ex.argtypes = [cp, C.c_char_p, sz, u32, f, f, i32, C.c_char_p, C.POINTER(vp),
                                         C.POINTER(sz), C.c_char_p, sz]
    L.aptgpu_resample_wav_file_ex.argtypes = [cp, C.c_char_p, C.c_char_p, u32, f, f, i32, C.c_char_p, sz]
    L.aptgpu_plan_decode_device_wav.argtypes = [vp, i32, C.POINTER(vp), wsp, C.POINTER(vp),
                                                C.POINTER(sz), C.c_char_p, sz]
    i32p = C.POINTER(C.c_int32)
    for name in ("aptgpu_decode_batch", "aptgpu_decode_batch_wav"):
        getattr(L, name).argtypes = [cp, C.POINTER(_CSettings), u32, i32, i32, C.POINTER(vp), C.POINTER(sz), i32p, i32,
                                     i32, C.POINTER(_f32p), C.POINTER(sz), i32p, C.POINTER(Result),
                                     C.POINTER(BatchStats), C.c_char_p, sz]
    L.aptgpu_host_alloc.argtypes = [sz]
    L.aptgpu_host_alloc.restype = vp
    L.aptgpu_host_free.argtypes = [vp]
    L.aptgpu_host_free.restype = None
    L.aptgpu_host_affinity.argtypes = [i32, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, sz]
    L.aptgpu_host_affinity_from_sysfs.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, sz]
    L.aptgpu_get_min.argtypes = [cp, _f32p, sz, _f32p, C.c_char_p, sz]
    L.aptgpu_get_max.argtypes = [cp, _f32p, sz, _f32p, C.c_char_p, sz]
    L.aptgpu_percent.argtypes = [cp, _f32p, sz, f, _f32p, _f32p, C.c_char_p, sz]
    L.aptgpu_map_signal_u8.argtypes = [cp, _f32p, sz, f, f, C.POINTER(_u8p), C.c_char_p, sz]
    L.aptgpu_read_telemetry.argtypes = [cp, _f32p, sz, C.POINTER(ImageResult), C.c_char_p, sz]
    L.aptgpu_channel_name.argtypes = [i32]
    L.aptgpu_channel_name.restype = C.c_char_p
    L.aptgpu_process_gray.argtypes = [cp, _f32p, sz, i32, f, i32, C.POINTER(_u8p), C.POINTER(sz),
                                      C.POINTER(ImageResult), C.c_char_p, sz]
    L.aptgpu_plan_process_device.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), i32, f, i32,
                                             C.POINTER(vp), C.c_char_p, sz]
    L.aptgpu_plan_image_results.argtypes = [vp, i32, C.POINTER(ImageResult)]
    _lib = L
    return L


def version():
    return lib().aptgpu_version().decode()


def abi_version():
    return int(lib().aptgpu_abi_version())


def device_count():
    return int(lib().aptgpu_device_count())


def cache_clear():
    """Releases every idle session (plan + device buffers) of the host-array entry points' cache."""
    lib().aptgpu_cache_clear()


def cache_info():
    """(idle sessions, device bytes they hold)."""
    e, b = C.c_int32(0), C.c_uint64(0)
    lib().aptgpu_cache_info(C.byref(e), C.byref(b))
    return int(e.value), int(b.value)


def host_affinity(device=0):
    """(PCI address, NUMA node, CPU list) a batch worker of `device` pins itself to; node -1 / '' when the
    platform does not say."""
    bdf, node, cpus = C.create_string_buffer(32), C.c_int32(-1), C.create_string_buffer(4096)
    lib().aptgpu_host_affinity(int(device), bdf, C.byref(node), cpus, 4096)
    return bdf.value.decode(), int(node.value), cpus.value.decode()


def host_affinity_from_sysfs(sysfs_root, pci_bdf):
    """The same lookup on a PCI address under a given sysfs root (no GPU needed): (NUMA node, CPU list)."""
    node, cpus = C.c_int32(-1), C.create_string_buffer(4096)
    rc = lib().aptgpu_host_affinity_from_sysfs(str(sysfs_root).encode(), str(pci_bdf).encode(), C.byref(node), cpus, 4096)
    if rc != 0:
        raise InvalidError("bad argument to aptgpu_host_affinity_from_sysfs")
    return int(node.value), cpus.value.decode()


class _Owned:
    """A buffer the library malloc'd, exposed to numpy without a copy (ten megabytes of rows per ten-minute recording:
    a copy costs as much as a fifth of the decode); aptgpu_free runs when the last array viewing it is gone."""
    __slots__ = ("addr", "free", "__array_interface__")

    def __init__(self, addr, n, dtype):
        self.addr = addr
        self.free = lib().aptgpu_free  # (bound now: module globals may be gone when the last array dies at exit)
        self.__array_interface__ = {"shape": (n,), "typestr": np.dtype(dtype).str, "data": (addr, False), "version": 3}

    def __del__(self):
        self.free(self.addr)


def _take(ptr, n, dtype=np.float32):
    n = int(n)
    addr = C.cast(ptr, C.c_void_p).value
    if n and addr and C.sizeof(ptr._type_) == np.dtype(dtype).itemsize:
        return np.asarray(_Owned(addr, n, dtype))
    out = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True) if n else np.zeros(0, dtype)
    lib().aptgpu_free(C.cast(ptr, C.c_void_p))
    return out


def _as_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


# ------------------------------------------------------------------ frequency.rs
@dataclass(frozen=True)
class Rate:
    """Sample rate in Hz (frequency.rs:98-117)."""
    hz_: int

    @staticmethod
    def hz(r):
        return Rate(int(r))

    def get_hz(self):
        return self.hz_


@dataclass(frozen=True)
class Freq:
    """Discrete-time frequency as a fraction of pi rad/sample, f32 (frequency.rs:30-88)."""
    pi_rad_: float

    @staticmethod
    def pi_rad(f):
        return Freq(float(np.float32(f)))

    @staticmethod
    def hz(f, rate: Rate):
        return Freq(float(np.float32(2.0) * np.float32(f) / np.float32(rate.get_hz())))

    def get_pi_rad(self):
        return self.pi_rad_

    def __truediv__(self, o):
        return Freq(float(np.float32(self.pi_rad_) / np.float32(o)))


# ------------------------------------------------------------------ config.rs / context.rs
@dataclass
class Settings:
    """The config::Settings fields decode() reads; defaults = the `standard` profile
    (/root/reference/src/default_settings.toml:108-116)."""
    work_rate: int = 12480
    resample_atten: float = 30.0
    resample_delta_freq: float = 1000.0
    resample_cutout: float = 4800.0
    demodulation_atten: float = 25.0
    export_wav: bool = False
    export_resample_filtered: bool = False
    # used only by the WAV->WAV resample tool (config.rs:100-106)
    wav_resample_atten: float = 40.0
    wav_resample_delta_freq: float = 0.1

    @staticmethod
    def profile(name):
        p = {"standard": dict(work_rate=12480, resample_atten=30.0, resample_delta_freq=1000.0,
                              resample_cutout=4800.0, demodulation_atten=25.0,
                              wav_resample_atten=40.0, wav_resample_delta_freq=0.1),
             "fast": dict(work_rate=16640, resample_atten=30.0, resample_delta_freq=3000.0,
                          resample_cutout=4800.0, demodulation_atten=23.0,
                          wav_resample_atten=30.0, wav_resample_delta_freq=0.2),
             "slow": dict(work_rate=20800, resample_atten=40.0, resample_delta_freq=500.0,
                          resample_cutout=4800.0, demodulation_atten=25.0,
                          wav_resample_atten=50.0, wav_resample_delta_freq=0.05)}[name]
        return Settings(**p)

    def _c(self):
        return _CSettings(self.work_rate, self.resample_atten, self.resample_delta_freq,
                          self.resample_cutout, self.demodulation_atten,
                          1 if self.export_wav else 0, 1 if self.export_resample_filtered else 0)


@dataclass
class Context:
    """context::Context: progress callback + step export (context.rs:100-211)."""
    ui_callback: Optional[Callable[[float, str], None]] = None
    step_callback: Optional[Callable[[str, int, np.ndarray, Optional[int]], None]] = None
    device: int = 0
    mode: int = MODE_STRICT
    stream: int = 0
    _keep: list = field(default_factory=list, repr=False)

    @staticmethod
    def decode(ui_callback=None, step_callback=None, device=0, mode=MODE_STRICT):
        return Context(ui_callback, step_callback, device, mode)

    def _c(self):
        def status(progress, text, _user):
            if self.ui_callback:
                self.ui_callback(float(progress), text.decode())

        def step(ident, variant, data, n, rate, _user):
            if self.step_callback:
                arr = (np.ctypeslib.as_array(data, shape=(int(n),)).copy() if n
                       else np.zeros(0, np.float32))
                try:
                    self.step_callback(ident.decode(), int(variant), arr, int(rate) or None)
                except Exception:  # propagate like `?`
                    return 1
            return 0

        s, t = _STATUS_FN(status), _STEP_FN(step)
        self._keep = [s, t]
        return _CContext(s, t, None, self.device, self.mode, self.stream or None)


# ------------------------------------------------------------------ filters.rs
@dataclass
class NoFilter:
    kind = 0

    def _c(self):
        return _CFilter(0, 0.0, 0.0, 0.0)

    def design(self):
        return _design(self._c())

    def resample(self, input_rate: Rate, output_rate: Rate):
        pass


@dataclass
class Lowpass:
    cutout: Freq
    atten: float
    delta_w: Freq
    kind = 1

    def _c(self):
        return _CFilter(self.kind, self.cutout.get_pi_rad(), self.atten, self.delta_w.get_pi_rad())

    def design(self):
        return _design(self._c())

    def resample(self, input_rate: Rate, output_rate: Rate):
        c = self._c()
        lib().aptgpu_filter_resample(C.byref(c), input_rate.get_hz(), output_rate.get_hz())
        self.cutout, self.delta_w = Freq(float(c.cutout_pi_rad)), Freq(float(c.delta_w_pi_rad))


@dataclass
class LowpassDcRemoval(Lowpass):
    kind = 2


def _design(cf):
    out, n = _f32p(), C.c_size_t()
    _check(lib().aptgpu_filter_design(C.byref(cf), C.byref(out), C.byref(n)))
    return _take(out, n.value)


# ------------------------------------------------------------------ decode.rs / dsp.rs
def decode(context: Optional[Context], settings: Settings, signal, input_rate: Rate, sync: bool,
           return_stats=False):
    """noaa_apt::decode — returns the raw image, line by line (flat f32, rows*2080)."""
    ctx = (context or Context())
    cctx, cs = ctx._c(), settings._c()
    x, xp = _as_f32(signal)
    out, n, st = _f32p(), C.c_size_t(), Stats()
    err = C.create_string_buffer(_ERRCAP)
    rc = lib().aptgpu_decode(C.byref(cctx), C.byref(cs), xp, x.size, input_rate.get_hz(),
                             1 if sync else 0, C.byref(out), C.byref(n), C.byref(st), err, _ERRCAP)
    _check(rc, err)
    rows = _take(out, n.value)
    return (rows, st) if return_stats else rows


def resample_with_filter(context, signal, input_rate: Rate, output_rate: Rate, filt):
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    out, n = _f32p(), C.c_size_t()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_resample_with_filter(C.byref(cctx), xp, x.size, input_rate.get_hz(),
                                             output_rate.get_hz(), filt._c(), C.byref(out),
                                             C.byref(n), err, _ERRCAP), err)
    return _take(out, n.value)


def resample(context, signal, input_rate: Rate, output_rate: Rate, atten, delta_w: Freq):
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    out, n = _f32p(), C.c_size_t()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_resample(C.byref(cctx), xp, x.size, input_rate.get_hz(),
                                 output_rate.get_hz(), atten, delta_w.get_pi_rad(), C.byref(out),
                                 C.byref(n), err, _ERRCAP), err)
    return _take(out, n.value)


def demodulate(context, signal, carrier_freq: Freq):
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    out = _f32p()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_demodulate(C.byref(cctx), xp, x.size, carrier_freq.get_pi_rad(),
                                   C.byref(out), err, _ERRCAP), err)
    return _take(out, x.size)


def filter(context, signal, filt):  # noqa: A001 - the reference's name (dsp.rs:386)
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    out = _f32p()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_filter_signal(C.byref(cctx), xp, x.size, filt._c(), C.byref(out), err,
                                      _ERRCAP), err)
    return _take(out, x.size)


def find_sync(context, signal, work_rate: Rate, return_correlation=False):
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    pos, npos, corr, ncorr = _u64p(), C.c_size_t(), _f32p(), C.c_size_t()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_find_sync(C.byref(cctx), xp, x.size, work_rate.get_hz(), C.byref(pos),
                                  C.byref(npos), C.byref(corr) if return_correlation else None,
                                  C.byref(ncorr), err, _ERRCAP), err)
    p = _take(pos, npos.value, np.uint64)
    return (p, _take(corr, ncorr.value)) if return_correlation else p


def generate_sync_frame(work_rate: Rate):
    out, n = _i8p(), C.c_size_t()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_generate_sync_frame(work_rate.get_hz(), C.byref(out), C.byref(n), err,
                                            _ERRCAP), err)
    return _take(out, n.value, np.int8)


# ------------------------------------------------------------------ WAV ingest
def wav_parse(file_bytes: bytes) -> WavSpec:
    """hound::WavReader::new(...).spec() on an in-memory file image (host only)."""
    spec = WavSpec()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_wav_parse(file_bytes, len(file_bytes), C.byref(spec), err, _ERRCAP), err)
    return spec


def load(input_filename, context=None, return_spec=False):
    """noaa_apt::load (noaa_apt.rs:114-130): (Signal, Rate).  Also accepts the file's bytes."""
    cctx = (context or Context())._c()
    out, n, rate, spec = _f32p(), C.c_size_t(), C.c_uint32(), WavSpec()
    err = C.create_string_buffer(_ERRCAP)
    if isinstance(input_filename, (bytes, bytearray, memoryview)):
        data = bytes(input_filename)
        _check(lib().aptgpu_load_wav(C.byref(cctx), data, len(data), C.byref(out), C.byref(n),
                                     C.byref(rate), C.byref(spec), err, _ERRCAP), err)
    else:
        _check(lib().aptgpu_load_wav_file(C.byref(cctx), os.fsencode(input_filename), C.byref(out),
                                          C.byref(n), C.byref(rate), C.byref(spec), err, _ERRCAP), err)
    res = (_take(out, n.value), Rate.hz(rate.value))
    return res + (spec,) if return_spec else res


def decode_wav(context: Optional[Context], settings: Settings, file_bytes: bytes, sync: bool,
               return_stats=False):
    """load() + decode() without the host-side f32 detour: the data chunk goes to the GPU as it
    is and is converted there (inside the fused front end for mono PCM16)."""
    cctx = (context or Context())._c()
    cs = settings._c()
    out, n, st, rate = _f32p(), C.c_size_t(), Stats(), C.c_uint32()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_decode_wav(C.byref(cctx), C.byref(cs), file_bytes, len(file_bytes), int(sync),
                                   C.byref(out), C.byref(n), C.byref(st), C.byref(rate), err, _ERRCAP), err)
    rows = _take(out, n.value)
    return (rows, st) if return_stats else rows


def _take_bytes(ptr, n):
    out = C.string_at(ptr, n) if n else b""
    lib().aptgpu_free(ptr)
    return out


def write_wav(signal, sample_rate: Rate, context=None) -> bytes:
    """wav::write_wav (wav.rs:59-98) with the {1 channel, 16 bit, Int} spec: the file image."""
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    out, n = C.c_void_p(), C.c_size_t()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_write_wav_i16(C.byref(cctx), xp, x.size, sample_rate.get_hz(), C.byref(out),
                                      C.byref(n), err, _ERRCAP), err)
    return _take_bytes(out, n.value)


def resample_wav(context, settings, input_wav, output_filename, output_rate: int):
    """resample::resample (resample.rs:17-71).  `input_wav` / `output_filename` are paths (the
    result is written, modification time copied) — or pass the input file's bytes and get the
    output file's bytes back (output_filename then only labels the status text)."""
    cctx = (context or Context())._c()
    atten, delta = settings.wav_resample_atten, settings.wav_resample_delta_freq
    flag = 1 if settings.export_resample_filtered else 0  # main.rs:125-130 -> Context::resample
    err = C.create_string_buffer(_ERRCAP)
    if isinstance(input_wav, (bytes, bytearray, memoryview)):
        data = bytes(input_wav)
        out, n = C.c_void_p(), C.c_size_t()
        name = os.fsencode(output_filename) if output_filename else None
        _check(lib().aptgpu_resample_wav_ex(C.byref(cctx), data, len(data), output_rate, atten, delta, flag, name,
                                            C.byref(out), C.byref(n), err, _ERRCAP), err)
        return _take_bytes(out, n.value)
    _check(lib().aptgpu_resample_wav_file_ex(C.byref(cctx), os.fsencode(input_wav), os.fsencode(output_filename),
                                             output_rate, atten, delta, flag, err, _ERRCAP), err)
    return None


# ------------------------------------------------------------------ consumers of the rows
class Contrast:
    """noaa_apt::Contrast (noaa_apt.rs:25-37).  Histogram's equalisation is host-side and out
    of scope; it takes MinMax limits first (noaa_apt.rs:158)."""
    TELEMETRY, MINMAX = ("telemetry",), ("minmax",)

    @staticmethod
    def Percent(p):  # noqa: N802 - the reference's variant name
        return ("percent", float(p))

    @staticmethod
    def _c(contrast):
        kind = contrast[0]
        return ({"telemetry": 0, "percent": 1, "minmax": 2}[kind],
                contrast[1] if kind == "percent" else 0.0)


class Rotate:
    """noaa_apt::Rotate (noaa_apt.rs:52-60); Orbit needs orbit propagation (out of scope)."""
    NO, YES, ORBIT = 0, 1, 2


class Telemetry:
    """telemetry::Telemetry (telemetry.rs:19-121): wedge values of both bands."""

    def __init__(self, values_a, values_b, row=0, quality=0.0, channels=(-1, -1)):
        self.values_a = np.asarray(values_a, np.float32)
        self.values_b = np.asarray(values_b, np.float32)
        self.row, self.quality, self._channels = int(row), np.float32(quality), channels

    @staticmethod
    def _from(r: ImageResult):
        return Telemetry(list(r.values_a), list(r.values_b), r.telemetry_row, r.telemetry_quality,
                         (r.channel_a, r.channel_b))

    def get_wedge_value(self, wedge, channel=None):
        if channel == "A":
            return self.values_a[wedge - 1]
        if channel == "B":
            return self.values_b[wedge - 1]
        return np.float32((self.values_a[wedge - 1] + self.values_b[wedge - 1]) / np.float32(2.))

    def get_channel_name(self, channel):
        i = self._channels[{"A": 0, "B": 1}[channel]]
        if i < 0:
            raise InternalError("Can't compare values")
        return lib().aptgpu_channel_name(i).decode()


def _reduce(fn, context, signal):
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    out = C.c_float()
    err = C.create_string_buffer(_ERRCAP)
    _check(fn(C.byref(cctx), xp, x.size, C.byref(out), err, _ERRCAP), err)
    return np.float32(out.value)


def get_min(signal, context=None):   # dsp.rs:38
    return _reduce(lib().aptgpu_get_min, context, signal)


def get_max(signal, context=None):   # dsp.rs:20
    return _reduce(lib().aptgpu_get_max, context, signal)


def percent(signal, percent_value, context=None):  # misc.rs:119
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    lo, hi = C.c_float(), C.c_float()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_percent(C.byref(cctx), xp, x.size, percent_value, C.byref(lo), C.byref(hi),
                                err, _ERRCAP), err)
    return np.float32(lo.value), np.float32(hi.value)


def map_signal_u8(signal, low, high, context=None):  # noaa_apt.rs:249
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    out = _u8p()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_map_signal_u8(C.byref(cctx), xp, x.size, low, high, C.byref(out), err,
                                      _ERRCAP), err)
    return _take(out, x.size, np.uint8)


def read_telemetry(context, signal):  # telemetry.rs:125
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    r = ImageResult()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_read_telemetry(C.byref(cctx), xp, x.size, C.byref(r), err, _ERRCAP), err)
    return Telemetry._from(r)


def process(context, signal, contrast_adjustment, rotate=Rotate.NO, color=None, orbit=None,
            return_info=False):
    """noaa_apt::process (noaa_apt.rs:132-235) up to the grayscale image: returns the
    height x 2080 u8 array.  False colour, histogram equalisation and the map overlay are
    host-side features of the reference that this path does not offer."""
    if color is not None or orbit is not None:
        raise UnsupportedError("false colour / map overlay are not part of the GPU path")
    cctx = (context or Context())._c()
    x, xp = _as_f32(signal)
    kind, p = Contrast._c(contrast_adjustment)
    img, n, info = _u8p(), C.c_size_t(), ImageResult()
    err = C.create_string_buffer(_ERRCAP)
    _check(lib().aptgpu_process_gray(C.byref(cctx), xp, x.size, kind, p, int(rotate), C.byref(img),
                                     C.byref(n), C.byref(info), err, _ERRCAP), err)
    out = _take(img, n.value, np.uint8).reshape(-1, PX_PER_ROW)
    return (out, info) if return_info else out


# ------------------------------------------------------------------ host-fed batch over GPUs
def decode_batch(context: Optional[Context], settings: Settings, inputs, input_rate: Rate, sync: bool,
                 devices: Sequence[int] = (), recordings_per_call: int = 0, return_stats=False):
    """aptgpu_decode_batch / aptgpu_decode_batch_wav: independent recordings — f32 Signals (numpy arrays) or
    WAV file images (bytes) — decoded on `devices` (ordinals, repeats allowed; empty = the context's
    device), no collectives.  Returns a list with one entry per recording: the rows (flat f32 array), or
    the AptError decode() would have raised for it; optionally also the result records and BatchStats."""
    cctx = (context or Context())._c()
    cs = settings._c()
    k = len(inputs)
    wav = k > 0 and isinstance(inputs[0], (bytes, bytearray, memoryview))
    keep, ptrs, sizes = [], (C.c_void_p * max(k, 1))(), (C.c_size_t * max(k, 1))()
    for i, x in enumerate(inputs):
        if wav:
            b = bytes(x)
            keep.append(b)
            ptrs[i] = C.cast(C.c_char_p(b), C.c_void_p)
            sizes[i] = len(b)
        else:
            a, ap = _as_f32(x)
            keep.append(a)
            ptrs[i] = C.cast(ap, C.c_void_p)
            sizes[i] = a.size
    devs = (C.c_int32 * max(len(devices), 1))(*devices)
    rows = (_f32p * max(k, 1))()
    n_out = (C.c_size_t * max(k, 1))()
    status = (C.c_int32 * max(k, 1))()
    results = (Result * max(k, 1))()
    stats = BatchStats()
    stats.struct_size = C.sizeof(BatchStats)
    err = C.create_string_buffer(_ERRCAP)
    fn = lib().aptgpu_decode_batch_wav if wav else lib().aptgpu_decode_batch
    rc = fn(C.byref(cctx), C.byref(cs), input_rate.get_hz(), int(sync), k, ptrs, sizes, devs, len(devices),
            int(recordings_per_call), rows, n_out, status, results, C.byref(stats), err, _ERRCAP)
    if rc != 0:
        # a worker-level failure (HIP error, bad argument): nothing of what finished before it is leaked
        for i in range(k):
            if rows[i]:
                lib().aptgpu_free(C.cast(rows[i], C.c_void_p))
        _check(rc, err)
    out = []
    for i in range(k):
        if status[i] == 0:
            out.append(_take(rows[i], n_out[i]))
            continue
        reason = {1: "Got less than 10 rows of samples, audio file is too short",
                  2: "Found less than 5 sync frames, audio file is too short or too noisy",
                  3: "work_rate is not multiple of FINAL_RATE"}.get(results[i].reason, "")
        if not reason and wav:
            # rejected before it reached a worker: the reference's message for this file (wav.rs / err.rs:72-83),
            # or the one thing a batch adds — every recording of a batch has the batch's input rate
            try:
                spec = wav_parse(keep[i])
                if spec.sample_rate != input_rate.get_hz():
                    reason = (f"recording {i}: WAV sample rate {spec.sample_rate} Hz differs from the batch's "
                              f"input rate {input_rate.get_hz()} Hz")
            except AptError as e:
                reason = str(e)
        elif not reason and sizes[i] and not ptrs[i]:
            reason = f"recording {i}: null input"
        out.append(_ERRORS.get(status[i], AptError)(reason or f"recording {i}: status {status[i]}"))
    return (out, list(results)[:k], stats) if return_stats else out


def host_alloc_f32(n: int) -> np.ndarray:
    """A pinned (hipHostMalloc) f32 buffer of n samples as a numpy array; free with host_free(array)."""
    p = lib().aptgpu_host_alloc(int(n) * 4)
    if not p:
        raise HipError("aptgpu_host_alloc failed")
    arr = np.ctypeslib.as_array(C.cast(p, _f32p), shape=(int(n),))
    _PINNED[arr.ctypes.data] = p
    return arr


def host_free(arr: np.ndarray):
    p = _PINNED.pop(arr.ctypes.data, None)
    if p:
        lib().aptgpu_host_free(p)


_PINNED = {}


# ------------------------------------------------------------------ plans (device-resident)
class Plan:
    """aptgpu_plan: device-resident / batched decode().  Device pointers are plain ints
    (e.g. torch.Tensor.data_ptr()); torch is only the allocator, never on the compute path."""

    def __init__(self, settings: Settings, input_rate: Rate, sync=True, max_samples=0, max_batch=1,
                 device=0, mode=MODE_STRICT, stream=0):
        self._ctx = Context(device=device, mode=mode, stream=stream)
        cctx, cs = self._ctx._c(), settings._c()
        self._p = C.c_void_p()
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().aptgpu_plan_create(C.byref(cctx), C.byref(cs), input_rate.get_hz(),
                                        1 if sync else 0, int(max_samples), int(max_batch),
                                        C.byref(self._p), err, _ERRCAP), err)
        self.info = PlanInfo()
        _check(lib().aptgpu_plan_get_info(self._p, C.byref(self.info)))

    def close(self):
        if self._p:
            lib().aptgpu_plan_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_device(self, d_signals: Sequence[int], n: Sequence[int], d_rows: Sequence[int],
                      rows_cap: Sequence[int]):
        k = len(d_signals)
        sig = (C.c_void_p * k)(*d_signals)
        rows = (C.c_void_p * k)(*d_rows)
        nn = (C.c_size_t * k)(*n)
        cap = (C.c_size_t * k)(*rows_cap)
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().aptgpu_plan_decode_device(self._p, k, sig, nn, rows, cap, err, _ERRCAP), err)

    def decode_device_wav(self, d_data: Sequence[int], specs: Sequence[WavSpec], d_rows: Sequence[int],
                          rows_cap: Sequence[int]):
        """As decode_device, but every recording is the payload of a WAV data chunk in HBM."""
        k = len(d_data)
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().aptgpu_plan_decode_device_wav(self._p, k, (C.c_void_p * k)(*d_data),
                                                   (WavSpec * k)(*specs), (C.c_void_p * k)(*d_rows),
                                                   (C.c_size_t * k)(*rows_cap), err, _ERRCAP), err)

    def process_device(self, d_rows: Sequence[int], rows_cap: Sequence[int], contrast_adjustment,
                       d_images: Sequence[int], rotate=Rotate.NO):
        """Contrast limits -> u8 image (and telemetry) of the recordings of the last
        decode_device call, chained on the device behind their decode."""
        k = len(d_rows)
        kind, p = Contrast._c(contrast_adjustment)
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().aptgpu_plan_process_device(self._p, k, (C.c_void_p * k)(*d_rows),
                                                (C.c_size_t * k)(*rows_cap), kind, p, int(rotate),
                                                (C.c_void_p * k)(*d_images), err, _ERRCAP), err)

    def image_results(self, count=1) -> List[ImageResult]:
        arr = (ImageResult * count)()
        _check(lib().aptgpu_plan_image_results(self._p, count, arr))
        return list(arr)

    def results(self, count=1) -> List[Result]:
        arr = (Result * count)()
        _check(lib().aptgpu_plan_results(self._p, count, arr))
        return list(arr)

    def sync_positions(self, i=0, cap=1 << 20):
        buf = (C.c_uint64 * cap)()
        n = C.c_size_t()
        _check(lib().aptgpu_plan_sync_positions(self._p, i, buf, cap, C.byref(n)))
        return np.array(buf[:min(cap, n.value)], dtype=np.uint64)

    def read_internal(self, name, dtype, count, i=0):
        """Download `count` elements of an internal HBM buffer (see aptgpu_plan_read_internal)."""
        out = np.zeros(int(count), dtype=dtype)
        size = C.c_size_t()
        _check(lib().aptgpu_plan_read_internal(self._p, i, name.encode(), out.ctypes.data_as(C.c_void_p),
                                               out.nbytes, C.byref(size)))
        return out

    def synchronize(self):
        _check(lib().aptgpu_plan_synchronize(self._p))

    def join(self):
        """Order the caller's stream (given at creation) after everything enqueued so far."""
        _check(lib().aptgpu_plan_join(self._p))

    def enable_timing(self, mode=2):
        """0/False off, 1 dominant kernel only, 2/True every kernel launch."""
        mode = 2 if mode is True else (0 if mode is False else int(mode))
        _check(lib().aptgpu_plan_enable_timing(self._p, mode))

    def collect_timing(self):
        arr = (KernelTime * 32)()
        n = C.c_size_t()
        _check(lib().aptgpu_plan_collect_timing(self._p, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): (arr[i].avg_ms, int(arr[i].launches))
                for i in range(min(32, n.value))}
