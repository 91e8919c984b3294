"""aptgpu_decode_batch: host-fed batches sharded over device entries (SURVEY.md §8(b) batch variant,
§8(e)): one host thread + plan per entry, no collective, every recording bit-identical to the oracle.
On the one-GPU test box the entries are {0}, {0, 0} (two workers on the same GPU); with more GPUs
visible, distinct devices too."""
import numpy as np
import pytest

import noaa_apt_amd as apt
from noaa_apt_amd.testing.synth import synth_apt
from noaa_apt_amd.testing.wavfile import make_wav

pytestmark = pytest.mark.gpu
f32 = np.float32


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module")
def recordings():
    # ragged lengths, different rate errors / noise / images, one far too short, one pure DC
    recs = [synth_apt(48000, 11 + (3 * i) % 7, 300 + i, ppm=7.0 * i - 20.0) for i in range(11)]
    recs.append(np.zeros(48000, f32))                 # < 10 rows of samples
    recs.append(np.full(48000 * 9, 1000.0, f32))      # constant: decodes, every position a terminal
    return recs


@pytest.mark.parametrize("devices,per_call", [((), 0), ((0,), 1), ((0, 0), 3), ((0, 0, 0), 4)])
def test_decode_batch_matches_oracle(oracle, recordings, devices, per_call):
    got, results, stats = apt.decode_batch(apt.Context(device=0), apt.Settings(), recordings, apt.Rate.hz(48000), True,
                                           devices=devices, recordings_per_call=per_call, return_stats=True)
    assert len(got) == len(recordings)
    assert stats.workers == max(1, len(devices)) and stats.samples == sum(r.size for r in recordings)
    assert stats.h2d_bytes == 4 * stats.samples
    for i, x in enumerate(recordings):
        try:
            want, st = oracle.decode(x, 48000, True, want_steps=True)
        except Exception as e:  # noqa: BLE001 - the oracle's error is the reference's
            assert isinstance(got[i], apt.InternalError) and str(got[i]) == str(e), (i, got[i], e)
            continue
        assert not isinstance(got[i], Exception), (i, got[i])
        assert _same(got[i], want), f"recording {i}"
        assert results[i].n_sync == st["sync_pos"].size and results[i].n_rows == want.size // 2080


def test_decode_batch_nosync_and_fast_mode(oracle, recordings):
    recs = recordings[:5]
    got = apt.decode_batch(None, apt.Settings(), recs, apt.Rate.hz(48000), False, devices=(0, 0))
    for x, g in zip(recs, got):
        assert _same(g, oracle.decode(x, 48000, False))
    from test_gpu_fast import PX_TOL
    got = apt.decode_batch(apt.Context(device=0, mode=apt.MODE_FAST), apt.Settings(), recs, apt.Rate.hz(48000), True,
                           devices=(0, 0), recordings_per_call=2)
    for x, g in zip(recs, got):
        want = oracle.decode(x, 48000, True)
        assert g.size == want.size
        assert float(np.max(np.abs(g - want))) <= PX_TOL * float(np.max(np.abs(want)))


def test_decode_batch_wav_images(oracle):
    """WAV file images: PCM16 payloads go over PCIe as they are (2 bytes per sample)."""
    recs = [synth_apt(48000, 11 + i, 400 + i) for i in range(5)]
    files = [make_wav(r.astype(np.int16), 48000) for r in recs]
    files.append(make_wav(recs[0].astype(np.int16), 44100))   # wrong rate for this batch
    files.append(b"RIFFxxxxWAVE")                              # not a WAV the reference would open
    got, results, stats = apt.decode_batch(None, apt.Settings(), files, apt.Rate.hz(48000), True, devices=(0, 0),
                                           recordings_per_call=2, return_stats=True)
    for r, g in zip(recs, got):
        assert _same(g, oracle.decode(r, 48000, True))
    assert isinstance(got[5], apt.InvalidError)
    assert isinstance(got[6], (apt.WavOpenError, apt.IoError))
    assert stats.h2d_bytes == sum(2 * r.size for r in recs)


def test_decode_batch_pinned_inputs(oracle):
    x = synth_apt(48000, 12, 77)
    pinned = apt.host_alloc_f32(x.size)
    pinned[:] = x
    try:
        got = apt.decode_batch(None, apt.Settings(), [pinned, x], apt.Rate.hz(48000), True, devices=(0,))
        want = oracle.decode(x, 48000, True)
        assert _same(got[0], want) and _same(got[1], want)
    finally:
        apt.host_free(pinned)


def test_decode_batch_on_every_visible_device(oracle):
    n = apt.device_count()
    if n < 2:
        pytest.skip("one GPU visible")
    recs = [synth_apt(48000, 11 + i % 3, 500 + i) for i in range(2 * n)]
    got = apt.decode_batch(None, apt.Settings(), recs, apt.Rate.hz(48000), True, devices=tuple(range(n)))
    for x, g in zip(recs, got):
        assert _same(g, oracle.decode(x, 48000, True))


def test_plans_on_two_host_threads_share_a_device(oracle):
    """Two plans, two host threads, one GPU — what aptgpu_decode_batch does with devices = {0, 0}, spelled
    out through the plan API (the per-device kernel attributes and the divide check are process-wide
    caches touched from both threads)."""
    import threading
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    xs = [synth_apt(48000, 12, 600 + i) for i in range(2)]
    wants = [oracle.decode(x, 48000, True) for x in xs]
    outs, errs = [None, None], []

    def run(i):
        try:
            plan = apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=xs[i].size)
            d_in = torch.from_numpy(xs[i]).to(dev)
            cap = int(plan.info.max_rows)
            d_out = torch.empty(cap * 2080, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            for _ in range(5):
                plan.decode_device([d_in.data_ptr()], [xs[i].size], [d_out.data_ptr()], [cap])
            res = plan.results(1)[0]
            outs[i] = d_out[:res.n_out].cpu().numpy()
            plan.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for o, w in zip(outs, wants):
        assert _same(o, w)
