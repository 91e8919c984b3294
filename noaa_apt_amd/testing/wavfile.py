"""Builds WAV file images for the ingest tests (test tooling only).

Covers what hound 3.5.1 — the reader behind wav::load_wav (/root/reference/src/wav.rs:11-57) —
accepts: PCMWAVEFORMAT (16-byte fmt), WAVEFORMATEX (18), 40-byte PCM, WAVE_FORMAT_EXTENSIBLE,
IEEE float, 8/16/24/32-bit integers, 24-bit in 4-byte containers, any channel count, extra
chunks before the data chunk.
"""
import struct

import numpy as np

_GUID_TAIL = bytes([0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xaa, 0x00, 0x38, 0x9b, 0x71])


def encode_samples(samples, bits, container_bytes=None, is_float=False):
    """Interleaved sample values -> little-endian bytes."""
    a = np.asarray(samples)
    if is_float:
        return a.astype("<f4").tobytes()
    nbytes = container_bytes or bits // 8
    v = a.astype(np.int64)
    if bits == 8 and nbytes == 1:
        return (v + 128).astype(np.uint8).tobytes()
    if nbytes == 2:
        return v.astype("<i2").tobytes()
    if nbytes == 3:
        u = (v & 0xFFFFFF).astype(np.uint32)
        out = np.empty((u.size, 3), np.uint8)
        out[:, 0], out[:, 1], out[:, 2] = u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF
        return out.tobytes()
    if nbytes == 4 and bits == 24:
        # the top byte of the container is junk on purpose: only 24 bits are valid
        u = (v & 0xFFFFFF).astype(np.uint32) | np.uint32(0x5A000000)
        return u.astype("<u4").tobytes()
    if nbytes == 4:
        return v.astype("<i4").tobytes()
    raise ValueError((bits, nbytes))


def make_wav(samples, rate, channels=1, bits=16, is_float=False, container_bytes=None, fmt_len=16,
             extensible=False, valid_bits=None, extra_chunks=(), fact=False, data_len_override=None,
             truncate=None, format_tag=None, byte_rate_override=None):
    """samples: interleaved values (frames * channels).  Returns the file image (bytes)."""
    if is_float and bits == 16:
        bits = 32  # the default width of a float file
    nbytes = container_bytes or (4 if is_float else bits // 8)
    data = encode_samples(samples, bits, nbytes, is_float)
    block_align = nbytes * channels
    container_bits = nbytes * 8
    tag = format_tag if format_tag is not None else (0xFFFE if extensible else (3 if is_float else 1))
    byte_rate = byte_rate_override if byte_rate_override is not None else block_align * rate
    fmt = struct.pack("<HHIIHH", tag, channels, rate, byte_rate, block_align,
                      container_bits if extensible else bits)
    if extensible:
        fmt += struct.pack("<HHI", 22, bits if valid_bits is None else valid_bits, 0)
        fmt += bytes([3 if is_float else 1, 0]) + _GUID_TAIL
    elif fmt_len == 18:
        fmt += struct.pack("<H", 0)
    elif fmt_len == 40:
        fmt += struct.pack("<H", 22) + bytes(22)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if fact:
        chunks += b"fact" + struct.pack("<II", 4, len(data) // max(block_align, 1))
    for name, payload in extra_chunks:
        chunks += name + struct.pack("<I", len(payload)) + payload
    dlen = len(data) if data_len_override is None else data_len_override
    chunks += b"data" + struct.pack("<I", dlen) + data
    out = b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks
    return out if truncate is None else out[:truncate]


def first_channel_f32(samples, channels, bits=16, is_float=False):
    """What wav::load_wav returns for these interleaved values: channel 0, `as f32`, unscaled."""
    a = np.asarray(samples)[::channels]
    if is_float:
        return a.astype(np.float32)
    return a.astype(np.int64).astype(np.float32)
