"""ctypes binding to oracle/libaptoracle.so (TEST INFRASTRUCTURE ONLY).

The oracle is the CPU restatement of the reference's decode() path
(/root/reference/src/{decode,dsp,filters,misc,frequency}.rs); see apt_oracle.h
for the per-function citations and the pinning status ("parity unpinned" for
decode()'s numeric output).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libaptoracle.so")

OK, ERR_INTERNAL, ERR_RATE_OVERFLOW = 0, 1, 2
NOFILTER, LOWPASS, LOWPASS_DC_REMOVAL = 0, 1, 2

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_i8p = C.POINTER(C.c_int8)


class Settings(C.Structure):
    _fields_ = [
        ("work_rate", C.c_uint32),
        ("resample_atten", C.c_float),
        ("resample_delta_freq", C.c_float),
        ("resample_cutout", C.c_float),
        ("demodulation_atten", C.c_float),
    ]


class FilterSpec(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("cutout_pi_rad", C.c_float),
        ("atten", C.c_float),
        ("delta_w_pi_rad", C.c_float),
    ]


class Steps(C.Structure):
    _fields_ = [
        ("resample_filter", _f32p), ("n_resample_filter", C.c_size_t),
        ("resampled", _f32p), ("n_resampled", C.c_size_t),
        ("demodulated", _f32p), ("n_demodulated", C.c_size_t),
        ("filter_filter", _f32p), ("n_filter_filter", C.c_size_t),
        ("filtered", _f32p), ("n_filtered", C.c_size_t),
        ("correlation", _f32p), ("n_correlation", C.c_size_t),
        ("sync_pos", _u64p), ("n_sync_pos", C.c_size_t),
        ("aligned", _f32p), ("n_aligned", C.c_size_t),
        ("t_resample", C.c_double), ("t_demod", C.c_double), ("t_filter", C.c_double),
        ("t_sync", C.c_double), ("t_gather", C.c_double),
    ]


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if not os.path.exists(_LIB_PATH) or any(os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in srcs):
        build()
    L = C.CDLL(_LIB_PATH)
    L.apt_oracle_freq_hz.restype = C.c_float
    L.apt_oracle_freq_hz.argtypes = [C.c_float, C.c_uint32]
    L.apt_oracle_freq_rad.restype = C.c_float
    L.apt_oracle_freq_rad.argtypes = [C.c_float]
    L.apt_oracle_freq_get_rad.restype = C.c_float
    L.apt_oracle_freq_get_rad.argtypes = [C.c_float]
    L.apt_oracle_freq_get_hz.restype = C.c_float
    L.apt_oracle_freq_get_hz.argtypes = [C.c_float, C.c_uint32]
    L.apt_oracle_bessel_i0.restype = C.c_float
    L.apt_oracle_bessel_i0.argtypes = [C.c_float]
    L.apt_oracle_kaiser.restype = _f32p
    L.apt_oracle_kaiser.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_size_t)]
    L.apt_oracle_filter_design.restype = _f32p
    L.apt_oracle_filter_design.argtypes = [C.POINTER(FilterSpec), C.POINTER(C.c_size_t)]
    L.apt_oracle_filter_resample.restype = None
    L.apt_oracle_filter_resample.argtypes = [C.POINTER(FilterSpec), C.c_uint32, C.c_uint32]
    L.apt_oracle_fast_resampling.restype = _f32p
    L.apt_oracle_fast_resampling.argtypes = [_f32p, C.c_size_t, C.c_uint32, C.c_uint32, _f32p,
                                             C.c_size_t, C.POINTER(C.c_size_t)]
    L.apt_oracle_decimate.restype = _f32p
    L.apt_oracle_decimate.argtypes = [_f32p, C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t)]
    L.apt_oracle_demodulate.restype = _f32p
    L.apt_oracle_demodulate.argtypes = [_f32p, C.c_size_t, C.c_float]
    L.apt_oracle_fir.restype = _f32p
    L.apt_oracle_fir.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t]
    L.apt_oracle_resample_with_filter.restype = C.c_int
    L.apt_oracle_resample_with_filter.argtypes = [
        _f32p, C.c_size_t, C.c_uint32, C.c_uint32, FilterSpec, C.POINTER(_f32p),
        C.POINTER(C.c_size_t), C.POINTER(_f32p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.apt_oracle_resample.restype = C.c_int
    L.apt_oracle_resample.argtypes = [
        _f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.POINTER(_f32p),
        C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.apt_oracle_generate_sync_frame.restype = C.c_int
    L.apt_oracle_generate_sync_frame.argtypes = [C.c_uint32, C.POINTER(_i8p),
                                                 C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.apt_oracle_find_sync.restype = C.c_int
    L.apt_oracle_find_sync.argtypes = [
        _f32p, C.c_size_t, C.c_uint32, C.POINTER(_u64p), C.POINTER(C.c_size_t),
        C.POINTER(_f32p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.apt_oracle_decode.restype = C.c_int
    L.apt_oracle_decode.argtypes = [
        C.POINTER(Settings), _f32p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(_f32p),
        C.POINTER(C.c_size_t), C.POINTER(Steps), C.c_char_p, C.c_size_t]
    L.apt_oracle_fast_resampling_export.restype = _f32p
    L.apt_oracle_fast_resampling_export.argtypes = [_f32p, C.c_size_t, C.c_uint32, C.c_uint32, _f32p, C.c_size_t,
                                                    C.POINTER(C.c_size_t), C.POINTER(_f32p), C.POINTER(C.c_size_t)]
    L.apt_oracle_resample_ex.restype = C.c_int
    L.apt_oracle_resample_ex.argtypes = [
        _f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_int, C.POINTER(_f32p),
        C.POINTER(C.c_size_t), C.POINTER(_f32p), C.POINTER(C.c_size_t), C.POINTER(_f32p), C.POINTER(C.c_size_t),
        C.c_char_p, C.c_size_t]
    L.apt_oracle_decode_ex.restype = C.c_int
    L.apt_oracle_decode_ex.argtypes = [
        C.POINTER(Settings), _f32p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.POINTER(_f32p),
        C.POINTER(C.c_size_t), C.POINTER(Steps), C.POINTER(_f32p), C.POINTER(C.c_size_t), C.POINTER(_f32p),
        C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.apt_oracle_free.restype = None
    L.apt_oracle_free.argtypes = [C.c_void_p]
    L.apt_oracle_free_steps.restype = None
    L.apt_oracle_free_steps.argtypes = [C.POINTER(Steps)]
    _lib = L
    return L


def _as_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _take(ptr, n, dtype=np.float32):
    """Copy n elements out of a malloc'd oracle buffer and free it."""
    n = int(n)
    if n == 0:
        out = np.zeros(0, dtype=dtype)
    else:
        out = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    lib().apt_oracle_free(C.cast(ptr, C.c_void_p))
    return out


def _copy(ptr, n, dtype=np.float32):
    n = int(n)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def freq_hz(f, rate):
    return float(lib().apt_oracle_freq_hz(f, rate))


def freq_rad(f):
    return float(lib().apt_oracle_freq_rad(f))


def freq_get_rad(pi_rad):
    return float(lib().apt_oracle_freq_get_rad(pi_rad))


def freq_get_hz(pi_rad, rate):
    return float(lib().apt_oracle_freq_get_hz(pi_rad, rate))


def bessel_i0(x):
    return float(lib().apt_oracle_bessel_i0(x))


def kaiser(atten, delta_w_pi_rad):
    n = C.c_size_t()
    p = lib().apt_oracle_kaiser(atten, delta_w_pi_rad, C.byref(n))
    return _take(p, n.value)


def filter_design(kind, cutout_pi_rad=0.0, atten=0.0, delta_w_pi_rad=0.0):
    spec = FilterSpec(kind, cutout_pi_rad, atten, delta_w_pi_rad)
    n = C.c_size_t()
    p = lib().apt_oracle_filter_design(C.byref(spec), C.byref(n))
    return _take(p, n.value)


def filter_resample(kind, cutout_pi_rad, atten, delta_w_pi_rad, in_rate, out_rate):
    spec = FilterSpec(kind, cutout_pi_rad, atten, delta_w_pi_rad)
    lib().apt_oracle_filter_resample(C.byref(spec), in_rate, out_rate)
    return spec.kind, float(spec.cutout_pi_rad), float(spec.atten), float(spec.delta_w_pi_rad)


def fast_resampling(x, l, m, coeff):
    x, xp = _as_f32(x)
    c, cp = _as_f32(coeff)
    n = C.c_size_t()
    p = lib().apt_oracle_fast_resampling(xp, x.size, l, m, cp, c.size, C.byref(n))
    return _take(p, n.value)


def fast_resampling_export(x, l, m, coeff):
    """fast_resampling with context.export_resample_filtered set (dsp.rs:265-273): (output, expanded)."""
    x, xp = _as_f32(x)
    c, cp = _as_f32(coeff)
    n, ne = C.c_size_t(), C.c_size_t()
    ex = _f32p()
    p = lib().apt_oracle_fast_resampling_export(xp, x.size, l, m, cp, c.size, C.byref(n), C.byref(ex), C.byref(ne))
    return _take(p, n.value), _take(ex, ne.value)


def decimate(x, m):
    x, xp = _as_f32(x)
    n = C.c_size_t()
    p = lib().apt_oracle_decimate(xp, x.size, m, C.byref(n))
    return _take(p, n.value)


def demodulate(x, carrier_pi_rad):
    x, xp = _as_f32(x)
    p = lib().apt_oracle_demodulate(xp, x.size, carrier_pi_rad)
    return _take(p, x.size)


def fir(x, coeff):
    x, xp = _as_f32(x)
    c, cp = _as_f32(coeff)
    p = lib().apt_oracle_fir(xp, x.size, cp, c.size)
    return _take(p, x.size)


def _check(rc, err):
    if rc != OK:
        raise OracleError(rc, err.value.decode("utf-8", "replace"))


def resample_with_filter(x, in_rate, out_rate, kind, cutout_pi_rad=0.0, atten=0.0,
                         delta_w_pi_rad=0.0, return_coeff=False):
    x, xp = _as_f32(x)
    out, n = _f32p(), C.c_size_t()
    co, nco = _f32p(), C.c_size_t()
    err = C.create_string_buffer(1024)
    spec = FilterSpec(kind, cutout_pi_rad, atten, delta_w_pi_rad)
    rc = lib().apt_oracle_resample_with_filter(xp, x.size, in_rate, out_rate, spec, C.byref(out),
                                               C.byref(n), C.byref(co), C.byref(nco), err, 1024)
    _check(rc, err)
    res = _take(out, n.value)
    coeff = _take(co, nco.value)
    return (res, coeff) if return_coeff else res


def resample(x, in_rate, out_rate, atten, delta_w_pi_rad):
    x, xp = _as_f32(x)
    out, n = _f32p(), C.c_size_t()
    err = C.create_string_buffer(1024)
    rc = lib().apt_oracle_resample(xp, x.size, in_rate, out_rate, atten, delta_w_pi_rad,
                                   C.byref(out), C.byref(n), err, 1024)
    _check(rc, err)
    return _take(out, n.value)


def resample_ex(x, in_rate, out_rate, atten, delta_w_pi_rad, export_resample_filtered=False):
    """dsp::resample under Context::resample(export_wav, export_resample_filtered): (output, dict of the steps
    "resample_filter" / "resample_filtered" as the reference would export them)."""
    x, xp = _as_f32(x)
    out, n = _f32p(), C.c_size_t()
    co, nco, ex, nex = _f32p(), C.c_size_t(), _f32p(), C.c_size_t()
    err = C.create_string_buffer(1024)
    rc = lib().apt_oracle_resample_ex(xp, x.size, in_rate, out_rate, atten, delta_w_pi_rad,
                                      1 if export_resample_filtered else 0, C.byref(out), C.byref(n), C.byref(co),
                                      C.byref(nco), C.byref(ex), C.byref(nex), err, 1024)
    _check(rc, err)
    steps = dict(resample_filter=_take(co, nco.value),
                 resample_filtered=_take(ex, nex.value) if ex else np.zeros(0, np.float32))
    return _take(out, n.value), steps


def generate_sync_frame(work_rate):
    out, n = _i8p(), C.c_size_t()
    err = C.create_string_buffer(1024)
    rc = lib().apt_oracle_generate_sync_frame(work_rate, C.byref(out), C.byref(n), err, 1024)
    _check(rc, err)
    return _take(out, n.value, dtype=np.int8)


def find_sync(x, work_rate, return_correlation=False):
    x, xp = _as_f32(x)
    pos, npos = _u64p(), C.c_size_t()
    corr, ncorr = _f32p(), C.c_size_t()
    err = C.create_string_buffer(1024)
    rc = lib().apt_oracle_find_sync(xp, x.size, work_rate, C.byref(pos), C.byref(npos),
                                    C.byref(corr) if return_correlation else None,
                                    C.byref(ncorr), err, 1024)
    _check(rc, err)
    p = _take(pos, npos.value, dtype=np.uint64)
    if return_correlation:
        return p, _take(corr, ncorr.value)
    return p


STANDARD = dict(work_rate=12480, resample_atten=30.0, resample_delta_freq=1000.0,
                resample_cutout=4800.0, demodulation_atten=25.0)
FAST = dict(work_rate=16640, resample_atten=30.0, resample_delta_freq=3000.0,
            resample_cutout=4800.0, demodulation_atten=23.0)
SLOW = dict(work_rate=20800, resample_atten=40.0, resample_delta_freq=500.0,
            resample_cutout=4800.0, demodulation_atten=25.0)


def decode(x, input_rate, sync=True, settings=None, want_steps=False, export_resample_filtered=False):
    """Oracle decode(): returns rows (flat f32, len rows*2080) [and a dict of steps].  export_resample_filtered is
    Context.export_resample_filtered (context.rs:113): it moves the decimation phase of fast_resampling (dsp.rs:265-273);
    with want_steps the dict then also holds "expanded1" / "expanded2", the two "resample_filtered" steps."""
    s = Settings(**(settings or STANDARD))
    x, xp = _as_f32(x)
    out, n = _f32p(), C.c_size_t()
    steps = Steps()
    e1, ne1, e2, ne2 = _f32p(), C.c_size_t(), _f32p(), C.c_size_t()
    want_ex = bool(want_steps and export_resample_filtered)
    err = C.create_string_buffer(1024)
    rc = lib().apt_oracle_decode_ex(C.byref(s), xp, x.size, input_rate, 1 if sync else 0,
                                    1 if export_resample_filtered else 0, C.byref(out), C.byref(n),
                                    C.byref(steps) if want_steps else None,
                                    C.byref(e1) if want_ex else None, C.byref(ne1) if want_ex else None,
                                    C.byref(e2) if want_ex else None, C.byref(ne2) if want_ex else None, err, 1024)
    expanded = {}
    if want_ex:
        expanded = dict(expanded1=_take(e1, ne1.value) if e1 else np.zeros(0, np.float32),
                        expanded2=_take(e2, ne2.value) if e2 else np.zeros(0, np.float32))
    if rc != OK:
        if want_steps:
            lib().apt_oracle_free_steps(C.byref(steps))
        _check(rc, err)
    rows = _take(out, n.value)
    if not want_steps:
        return rows
    d = dict(
        **expanded,
        resample_filter=_copy(steps.resample_filter, steps.n_resample_filter),
        resampled=_copy(steps.resampled, steps.n_resampled),
        demodulated=_copy(steps.demodulated, steps.n_demodulated),
        filter_filter=_copy(steps.filter_filter, steps.n_filter_filter),
        filtered=_copy(steps.filtered, steps.n_filtered),
        correlation=_copy(steps.correlation, steps.n_correlation),
        sync_pos=_copy(steps.sync_pos, steps.n_sync_pos, dtype=np.uint64),
        aligned=_copy(steps.aligned, steps.n_aligned),
        t_resample=steps.t_resample, t_demod=steps.t_demod, t_filter=steps.t_filter,
        t_sync=steps.t_sync, t_gather=steps.t_gather,
    )
    lib().apt_oracle_free_steps(C.byref(steps))
    return rows, d
