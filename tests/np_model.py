"""Independent numpy-f32 re-derivation of the decode() stages (test tooling).

Written from the reference's formulas (not from oracle/apt_oracle.c), in the closed
forms the HIP kernels use, so it checks BOTH the oracle restatement and the
reformulations:

* polyphase indexing of fast_resampling (/root/reference/src/dsp.rs:186-289):
  output k reads x[x0+i]*h[p+i*l], x0 = ceil(k*m/l), p = x0*l - k*m, ascending i;
* the peak picker of find_sync (/root/reference/src/decode.rs:239-253) as
  "terminals + orbit": T[i] <=> no corr[j] > corr[i] for j in (i, i+min_distance];
  every tracking phase started at s ends on the first terminal >= s.
"""
import numpy as np

import ctypes

f32 = np.float32
PI = f32(np.pi)

# Rust's f32::{sin,cos,powf} lower to the platform libm (glibc here); numpy's float32
# sin/cos/power use numpy's own SIMD kernels, which differ by an ulp now and then.
_libm = ctypes.CDLL("libm.so.6")
for _n in ("sinf", "cosf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]


def sinf(x):
    x = np.asarray(x, f32)
    return np.array([_libm.sinf(float(v)) for v in x.reshape(-1)], f32).reshape(x.shape)


def cosf(x):
    return f32(_libm.cosf(float(f32(x))))


def powf(a, b):
    return f32(_libm.powf(float(f32(a)), float(f32(b))))


def bessel_i0(x):
    x = f32(x)
    table = [f32(v) for v in (1.0, 0.25, 0.015625, 0.00043402777777777775, 6.781684027777777e-06,
                              6.781684027777778e-08, 4.709502797067901e-10,
                              2.4028075495244395e-12, 9.385966990329842e-15)]
    r = f32(0)
    for k in range(8, 0, -1):
        r = f32(r + table[k])
        r = f32(r * f32(x * x))
    return f32(r + f32(1))


def kaiser(atten, delta_w_pi_rad):
    atten = f32(atten)
    if atten > 50:
        beta = f32(f32(0.1102) * f32(atten - f32(8.7)))
    elif atten < 21:
        beta = f32(0)
    else:
        a = f32(atten - f32(21))
        # powf in f32 via libm: numpy's float32 power calls powf
        beta = f32(f32(f32(0.5842) * powf(a, f32(0.4))) + f32(f32(0.07886) * a))
    rad = f32(f32(delta_w_pi_rad) * PI)
    length = int(np.ceil(f32(f32(atten - f32(8)) / f32(f32(2.285) * rad)))) + 1
    if length % 2 == 0:
        length += 1
    h = (length - 1) // 2
    m = f32(length)
    out = []
    den = bessel_i0(beta)
    for ni in range(-h, h + 1):
        q = f32(f32(ni) / f32(m / f32(2)))
        arg = f32(beta * np.sqrt(f32(f32(1) - f32(q * q)), dtype=f32))
        out.append(f32(bessel_i0(arg) / den))
    return np.array(out, dtype=f32)


def design(kind, cutout, atten, delta_w):
    """kind: 'lowpass' | 'dcremoval'; frequencies in pi rad."""
    w = kaiser(atten, delta_w)
    h = (w.size - 1) // 2
    cutout = f32(cutout)
    half = f32(f32(delta_w) / f32(2))
    n = np.arange(-h, h + 1).astype(f32)
    npi = (n * PI).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        a = (sinf((npi * cutout).astype(f32)) / npi).astype(f32)
        if kind == "lowpass":
            filt = a
            filt[h] = cutout
        else:
            b = (sinf((npi * half).astype(f32)) / npi).astype(f32)
            filt = (a - b).astype(f32)
            filt[h] = f32(cutout - half)
    return (filt * w).astype(f32)


def resample_poly(x, l, m, coeff):
    """fast_resampling in polyphase closed form, vectorised over outputs."""
    x = np.asarray(x, f32)
    n = x.size
    t_len = coeff.size
    off = (t_len - 1) // 2
    if n * l <= off:
        return np.zeros(0, f32)
    w = -(-(n * l - off) // m)
    k = np.arange(w, dtype=np.int64)
    x0 = -(-(k * m) // l)
    p = x0 * l - k * m
    tp = -(-t_len // l)
    xpad = np.concatenate([x, np.zeros(tp + 1, f32)])
    cpad = np.concatenate([coeff.astype(f32), np.zeros(l + 1, f32)])
    s = np.zeros(w, f32)
    for i in range(tp):
        j = p + i * l
        valid = (j < t_len) & (x0 + i < n)
        prod = (cpad[np.minimum(j, t_len)] * xpad[np.minimum(x0 + i, n)]).astype(f32)
        s = np.where(valid, (s + prod).astype(f32), s)
    return s


def demodulate(x, work_rate):
    x = np.asarray(x, f32)
    pi_rad = f32(f32(f32(2) * f32(2400)) / f32(work_rate))
    phi = f32(f32(2) * f32(pi_rad * PI))
    cosphi2 = f32(cosf(phi) * f32(2))
    sinphi = f32(sinf(phi))
    sq = (x * x).astype(f32)
    y = np.zeros_like(x)
    inner = ((sq[:-1] + sq[1:]).astype(f32) - ((x[:-1] * x[1:]).astype(f32) * cosphi2).astype(f32)).astype(f32)
    with np.errstate(invalid="ignore"):
        y[1:] = (np.sqrt(inner, dtype=f32) / sinphi).astype(f32)
    return y


def fir_causal(x, h):
    """filter(): out[i] = sum_{j < min(i, T)} x[i-j]*h[j], ascending j."""
    x = np.asarray(x, f32)
    n = x.size
    s = np.zeros(n, f32)
    i = np.arange(n)
    for j in range(h.size):
        valid = i > j
        xs = np.where(valid, x[np.maximum(i - j, 0)], f32(0))
        s = np.where(valid, (s + (xs * h[j]).astype(f32)).astype(f32), s)
    return s


def sync_template(work_rate):
    pw = work_rate // 4160
    return np.array([-1] * (2 * pw) + ([-1] * (2 * pw) + [1] * (2 * pw)) * 7 + [-1] * (8 * pw),
                    dtype=np.int8)


def correlate(f, g):
    f = np.asarray(f, f32)
    nc = f.size - g.size
    c = np.zeros(max(nc, 0), f32)
    for j in range(g.size):
        seg = f[j:j + nc]
        c = (c + seg).astype(f32) if g[j] == 1 else (c - seg).astype(f32)
    return c


def terminals(corr, md):
    """T[i] = not any(corr'[j] > corr'[i] for j in (i, i+md]); corr'[0] = max(corr[0], 0)."""
    c = np.array(corr, f32, copy=True)
    n = c.size
    if n == 0:
        return np.zeros(0, bool)
    if not (c[0] > 0):
        c[0] = f32(0)
    # windowed max over (i, i+md] by brute force in chunks
    t = np.ones(n, bool)
    wmax = np.full(n, -np.inf, dtype=f32)
    # van Herk / Gil-Werman with block size md
    nb = -(-n // md) + 1
    pad = np.full(nb * md + md + 1, -np.inf, dtype=f32)
    pad[:n] = c
    blocks = pad[:nb * md].reshape(nb, md)
    pre = np.maximum.accumulate(blocks, axis=1).reshape(-1)           # prefix max inside block
    suf = np.maximum.accumulate(blocks[:, ::-1], axis=1)[:, ::-1].reshape(-1)  # suffix max
    i = np.arange(n)
    lo = i + 1
    hi = i + md
    # window [lo, hi] has length md: spans at most two blocks
    same = (lo // md) == (hi // md)
    wmax = np.where(same, suf[np.minimum(lo, nb * md - 1)], np.maximum(suf[np.minimum(lo, nb * md - 1)], pre[np.minimum(hi, nb * md - 1)]))
    # when lo is a block start and same block: suf[lo] covers the whole block = window  (ok)
    t = ~(wmax > c)
    return t


def find_sync_orbit(corr, spr, md):
    """Peak list of find_sync() computed through terminals + orbit."""
    n = corr.size
    if n == 0:
        return [0]
    t = terminals(corr, md)
    tpos = np.flatnonzero(t)

    def first_t(s):
        return int(tpos[np.searchsorted(tpos, s, side="left")])

    u = first_t(0)
    peaks = [u]
    length = 1
    while True:
        s = max(u + md + 1, (length + 1) * spr)
        if s >= n:
            break
        c = s // spr
        peaks += [s] * (c - length - 1)
        u = first_t(s)
        peaks.append(u)
        length = c
    return peaks


def gather_rows(f, peaks, spr):
    """decode.rs:120-134 + the final NoFilter /3 decimation (px(0,0) = 0)."""
    f = np.asarray(f, f32)
    pw = spr // 2080
    rows = []
    for p in peaks[:-1]:
        if p + spr < f.size:
            rows.append(f[p:p + spr:pw].copy())
    if not rows:
        return np.zeros(0, f32)
    out = np.concatenate(rows)
    out[0] = f32(0) * out[0] if np.isfinite(out[0]) else out[0]
    out[0] = f32(0)
    return out
