"""ctypes binding to the WAV-ingest oracle (apt_oracle_wav.c) — TEST INFRASTRUCTURE ONLY.

Restates /root/reference/src/wav.rs:11-57 on top of hound 3.5.1's reader semantics.
"""
import ctypes as C

import numpy as np

from . import binding as _b

ERR_WAV_OPEN, ERR_IO = 6, 7


class WavSpec(C.Structure):
    _fields_ = [("channels", C.c_uint16), ("bits_per_sample", C.c_uint16),
                ("bytes_per_sample", C.c_uint16), ("sample_format", C.c_uint16),
                ("sample_rate", C.c_uint32), ("data_offset", C.c_uint64), ("data_len", C.c_uint64),
                ("n_samples", C.c_uint64)]


_ready = False


def lib():
    global _ready
    L = _b.lib()
    if not _ready:
        L.apt_oracle_load_wav.restype = C.c_int
        L.apt_oracle_load_wav.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_float)),
                                          C.POINTER(C.c_size_t), C.POINTER(WavSpec), C.c_char_p, C.c_size_t]
        L.apt_oracle_write_wav_i16.restype = C.c_int
        L.apt_oracle_write_wav_i16.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_uint32,
                                               C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                               C.c_char_p, C.c_size_t]
        _ready = True
    return L


def load_wav(file_bytes: bytes):
    """(signal f32, spec) — wav::load_wav on an in-memory file image."""
    out, n, spec = C.POINTER(C.c_float)(), C.c_size_t(), WavSpec()
    err = C.create_string_buffer(512)
    rc = lib().apt_oracle_load_wav(file_bytes, len(file_bytes), C.byref(out), C.byref(n), C.byref(spec), err, 512)
    if rc != 0:
        raise _b.OracleError(rc, err.value.decode())
    return _b._take(out, n.value), spec


def write_wav_i16(signal, rate) -> bytes:
    """wav::write_wav with the resample tool's spec: the file image."""
    a, p = _b._as_f32(signal)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    err = C.create_string_buffer(512)
    rc = lib().apt_oracle_write_wav_i16(p, a.size, rate, C.byref(out), C.byref(n), err, 512)
    if rc != 0:
        raise _b.OracleError(rc, err.value.decode())
    return _b._take(out, n.value, np.uint8).tobytes()


def resample_wav(file_bytes: bytes, output_rate, atten, delta_w_pi_rad, export_resample_filtered=False,
                 return_steps=False):
    """resample::resample (resample.rs:17-71) between file images; return_steps: also the dict of what
    Context::resample would export ("input", "resample_filter", "resample_filtered", "resample_decimated")."""
    sig, spec = load_wav(file_bytes)
    res, steps = _b.resample_ex(sig, spec.sample_rate, output_rate, atten, delta_w_pi_rad, export_resample_filtered)
    if res.size == 0:
        raise _b.OracleError(1, "Got zero samples after resampling, audio file too short or output "
                                "sampling frequency too low")
    out = write_wav_i16(res, output_rate)
    if not return_steps:
        return out
    steps.update(input=sig, resample_decimated=res)
    return out, steps
