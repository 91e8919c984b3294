"""noaa_apt_amd — MI355X (gfx950) implementation of noaa-apt's decode() hot path.

This package is a thin ctypes mirror of the reference's Rust interface for the path
(`noaa_apt::decode` and the dsp/filters functions under it) on top of the C ABI in
include/aptgpu.h (libaptgpu.so: hand-written HIP kernels + host-side FIR design).
Names, argument meaning and error behaviour follow the reference:

    decode(context, settings, signal, input_rate, sync)   /root/reference/src/decode.rs:43-49
    Context.status / Context.step                          /root/reference/src/context.rs:127-211
    Settings (the 5 fields decode() reads)                 /root/reference/src/config.rs:76-106
    Rate, Freq                                             /root/reference/src/frequency.rs:30-117
    Lowpass, LowpassDcRemoval, NoFilter                    /root/reference/src/filters.rs:22-138
    resample_with_filter, resample, demodulate, filter     /root/reference/src/dsp.rs:62,132,350,386
    find_sync, generate_sync_frame                         /root/reference/src/decode.rs:171,204
    load (WAV ingest), decode_wav                          /root/reference/src/noaa_apt.rs:114-130, wav.rs:11-57
    resample_wav (WAV->WAV tool), write_wav                /root/reference/src/resample.rs:17-71, wav.rs:59-98
    process (grayscale part), Contrast, Rotate             /root/reference/src/noaa_apt.rs:25-60,132-235
    percent, get_min, get_max, map_signal_u8               /root/reference/src/misc.rs:119, dsp.rs:20-54
    read_telemetry, Telemetry                              /root/reference/src/telemetry.rs:19-243

There is no CPU fallback: if libaptgpu.so is missing or no GPU is present the calls
raise.  (The CPU oracle lives in oracle/ and is test infrastructure only.)
"""
from .api import (  # noqa: F401
    FINAL_RATE, PX_PER_ROW, CARRIER_FREQ,
    AptError, InternalError, RateOverflowError, HipError, InvalidError, UnsupportedError,
    WavOpenError, IoError, WavSpec, wav_parse, load, decode_wav, write_wav, resample_wav,
    Rate, Freq, Settings, Context, Stats,
    NoFilter, Lowpass, LowpassDcRemoval,
    decode, resample_with_filter, resample, demodulate, filter, find_sync, generate_sync_frame,
    Contrast, Rotate, Telemetry, ImageResult,
    get_min, get_max, percent, map_signal_u8, read_telemetry, process,
    Plan, PlanInfo, Result, KernelTime, decode_batch, BatchStats, host_alloc_f32, host_free,
    lib, lib_path, use_library, build, device_count, version, abi_version, cache_clear, cache_info, host_affinity, host_affinity_from_sysfs,
    MODE_STRICT, MODE_GENERIC, MODE_FP16_TAPS, MODE_FAST,
)
