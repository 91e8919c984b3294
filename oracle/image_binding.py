"""ctypes binding to the image-side oracle functions (apt_oracle_image.c) — TEST INFRASTRUCTURE ONLY.

Restates /root/reference/src/{misc.rs:119-175, dsp.rs:20-54, noaa_apt.rs:132-192,249-259,
telemetry.rs} — the consumers of decode()'s pixel rows.
"""
import ctypes as C

import numpy as np

from . import binding as _b

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_F16 = C.c_float * 16

CONTRAST_TELEMETRY, CONTRAST_PERCENT, CONTRAST_MINMAX = 0, 1, 2
CHANNEL_NAMES = ["1", "2", "3a", "4", "5", "3b", "Unknown", "Unknown", "Unknown"]

_ready = False


def lib():
    global _ready
    L = _b.lib()
    if _ready:
        return L
    L.apt_oracle_get_max.restype = C.c_int
    L.apt_oracle_get_max.argtypes = [_f32p, C.c_size_t, _f32p, C.c_char_p, C.c_size_t]
    L.apt_oracle_get_min.restype = C.c_int
    L.apt_oracle_get_min.argtypes = [_f32p, C.c_size_t, _f32p, C.c_char_p, C.c_size_t]
    L.apt_oracle_percent.restype = C.c_int
    L.apt_oracle_percent.argtypes = [_f32p, C.c_size_t, C.c_float, _f32p, _f32p, _u32p, C.c_char_p,
                                     C.c_size_t]
    L.apt_oracle_map_signal_u8.restype = None
    L.apt_oracle_map_signal_u8.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float, _u8p]
    L.apt_oracle_telemetry_from_bands.restype = None
    L.apt_oracle_telemetry_from_bands.argtypes = [_f32p, _f32p, C.c_size_t, C.c_size_t, _F16, _F16]
    L.apt_oracle_telemetry_wedge_value.restype = C.c_float
    L.apt_oracle_telemetry_wedge_value.argtypes = [_F16, _F16, C.c_uint32, C.c_int]
    L.apt_oracle_telemetry_channel_index.restype = C.c_int
    L.apt_oracle_telemetry_channel_index.argtypes = [_F16, _F16, C.c_int]
    L.apt_oracle_read_telemetry.restype = C.c_int
    L.apt_oracle_read_telemetry.argtypes = [
        _f32p, C.c_size_t, _F16, _F16, C.POINTER(C.c_uint64), _f32p, C.POINTER(_f32p),
        C.POINTER(_f32p), C.POINTER(_f32p), C.POINTER(_f32p), C.POINTER(_f32p),
        C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.apt_oracle_process_gray.restype = C.c_int
    L.apt_oracle_process_gray.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_float, C.POINTER(_u8p),
                                          C.POINTER(C.c_size_t), _f32p, _f32p, C.c_char_p, C.c_size_t]
    _ready = True
    return L


def _check(rc, err):
    if rc != 0:
        raise _b.OracleError(rc, err.value.decode())


def _reduce(fn, x):
    a, p = _b._as_f32(x)
    out = C.c_float()
    err = C.create_string_buffer(512)
    _check(fn(p, a.size, C.byref(out), err, 512), err)
    return np.float32(out.value)


def get_max(x):
    return _reduce(lib().apt_oracle_get_max, x)


def get_min(x):
    return _reduce(lib().apt_oracle_get_min, x)


def percent(x, p, want_buckets=False):
    a, ptr = _b._as_f32(x)
    lo, hi = C.c_float(), C.c_float()
    buckets = np.zeros(1000, np.uint32)
    err = C.create_string_buffer(512)
    _check(lib().apt_oracle_percent(ptr, a.size, p, C.byref(lo), C.byref(hi),
                                    buckets.ctypes.data_as(_u32p), err, 512), err)
    res = (np.float32(lo.value), np.float32(hi.value))
    return res + (buckets,) if want_buckets else res


def map_signal_u8(x, low, high):
    a, ptr = _b._as_f32(x)
    out = np.zeros(a.size, np.uint8)
    lib().apt_oracle_map_signal_u8(ptr, a.size, low, high, out.ctypes.data_as(_u8p))
    return out


class Telemetry:
    """values_a / values_b as in telemetry.rs:19-23, plus the chosen frame row."""

    def __init__(self, values_a, values_b, row=0, quality=0.0, steps=None):
        self.values_a = np.asarray(values_a, np.float32)
        self.values_b = np.asarray(values_b, np.float32)
        self.row = int(row)
        self.quality = np.float32(quality)
        self.steps = steps or {}

    def _arrs(self):
        return _F16(*self.values_a.tolist()), _F16(*self.values_b.tolist())

    def get_wedge_value(self, wedge, channel=None):
        a, b = self._arrs()
        ch = {None: -1, "A": 0, "B": 1}[channel]
        return np.float32(lib().apt_oracle_telemetry_wedge_value(a, b, wedge, ch))

    def get_channel_name(self, channel):
        a, b = self._arrs()
        i = lib().apt_oracle_telemetry_channel_index(a, b, {"A": 0, "B": 1}[channel])
        if i < 0:
            raise _b.OracleError(1, "Can't compare values")
        return CHANNEL_NAMES[i]


def telemetry_from_bands(means_a, means_b, row):
    a, pa = _b._as_f32(means_a)
    b, pb = _b._as_f32(means_b)
    va, vb = _F16(), _F16()
    lib().apt_oracle_telemetry_from_bands(pa, pb, a.size, row, va, vb)
    return Telemetry(list(va), list(vb), row)


def read_telemetry(signal):
    a, ptr = _b._as_f32(signal)
    va, vb = _F16(), _F16()
    row, q = C.c_uint64(), C.c_float()
    ptrs = [_f32p() for _ in range(5)]
    rows = C.c_size_t()
    err = C.create_string_buffer(512)
    rc = lib().apt_oracle_read_telemetry(ptr, a.size, va, vb, C.byref(row), C.byref(q),
                                         *[C.byref(p) for p in ptrs], C.byref(rows), err, 512)
    r = rows.value
    nc = max(r - 200, 0)
    names = ["telemetry_a", "telemetry_b", "telemetry_variance", "telemetry_correlation",
             "telemetry_quality"]
    sizes = [r, r, r, nc, nc]
    steps = {}
    for name, p, n in zip(names, ptrs, sizes):
        if p:
            steps[name] = _b._take(p, n)
    _check(rc, err)
    return Telemetry(list(va), list(vb), row.value, q.value, steps)


def process_gray(signal, contrast, percent_value=0.98):
    a, ptr = _b._as_f32(signal)
    img = _u8p()
    n = C.c_size_t()
    lo, hi = C.c_float(), C.c_float()
    err = C.create_string_buffer(512)
    _check(lib().apt_oracle_process_gray(ptr, a.size, contrast, percent_value, C.byref(img),
                                         C.byref(n), C.byref(lo), C.byref(hi), err, 512), err)
    out = _b._take(img, n.value, np.uint8)
    return out, np.float32(lo.value), np.float32(hi.value)
