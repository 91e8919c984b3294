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


def test_session_cache_serves_the_one_shot_decode(oracle):
    """aptgpu_decode() takes its plan and device buffers from the process-wide session cache (threading
    row): a second call with the same key reuses the session — also for a SHORTER recording and, within the headroom
    a session is built with, a slightly longer one — a longer one gets a larger session, every result stays
    bit-identical to the oracle, and aptgpu_cache_clear() leaves nothing idle."""
    apt.cache_clear()
    assert apt.cache_info() == (0, 0)
    ctx, st, rate = apt.Context(device=0), apt.Settings(), apt.Rate.hz(48000)
    xs = [synth_apt(48000, s, 900 + i) for i, s in enumerate((12, 12, 9, 12.5, 30))]
    entries = []
    for x in xs:
        assert _same(apt.decode(ctx, st, x, rate, True), oracle.decode(x, 48000, True))
        entries.append(apt.cache_info()[0])
    # one session serves the first four (same key, lengths within its capacity); the 30 s recording needs another
    assert entries[:4] == [1, 1, 1, 1] and entries[4] == 2, entries
    assert apt.cache_info()[1] > 0
    # another key (no sync, fast mode, another rate): separate sessions, same answers as uncached plans give
    x = xs[0]
    assert _same(apt.decode(ctx, st, x, rate, False), oracle.decode(x, 48000, False))
    y = synth_apt(11025, 20, 77)
    assert _same(apt.decode(ctx, st, y, apt.Rate.hz(11025), True), oracle.decode(y, 11025, True))
    assert apt.cache_info()[0] == 4
    # errors do not poison later calls
    with pytest.raises(apt.InternalError):
        apt.decode(ctx, st, np.zeros(48000, f32), rate, True)
    assert _same(apt.decode(ctx, st, xs[2], rate, True), oracle.decode(xs[2], 48000, True))
    apt.cache_clear()
    assert apt.cache_info() == (0, 0)


def test_session_cache_concurrent_callers_get_their_own_session(oracle):
    import threading
    apt.cache_clear()
    x = synth_apt(48000, 14, 4242)
    want = oracle.decode(x, 48000, True)
    out = [None] * 4

    def run(k):
        out[k] = apt.decode(apt.Context(device=0), apt.Settings(), x, apt.Rate.hz(48000), True)

    for _ in range(2):
        ts = [threading.Thread(target=run, args=(k,)) for k in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert all(_same(o, want) for o in out)
    assert 1 <= apt.cache_info()[0] <= 4
    apt.cache_clear()


def test_decode_batch_many_calls_in_flight_and_reused_session(oracle):
    """More calls than buffer sets (three): every set is reused, uploads run two calls ahead of the decode, rows are
    DMA'd into the returned buffers — twice over the same cached session."""
    recs = [synth_apt(48000, 10.5 + 0.5 * (i % 3), 5000 + i) for i in range(14)]
    want = [oracle.decode(x, 48000, True) for x in recs]
    for _ in range(2):
        got = apt.decode_batch(apt.Context(device=0), apt.Settings(), recs, apt.Rate.hz(48000), True, devices=(0,),
                               recordings_per_call=2)
        assert all(_same(g, w) for g, w in zip(got, want))
    wavs = [make_wav(x.astype(np.int16), 48000) for x in recs]
    got = apt.decode_batch(apt.Context(device=0), apt.Settings(), wavs, apt.Rate.hz(48000), True, devices=(0, 0),
                           recordings_per_call=3)
    assert all(_same(g, w) for g, w in zip(got, want))


def test_decode_batch_says_why_a_wav_was_rejected():
    good = make_wav(synth_apt(48000, 11, 1).astype(np.int16), 48000)
    other_rate = make_wav(synth_apt(11025, 11, 2).astype(np.int16), 11025)
    broken = bytes(good[:20])
    got = apt.decode_batch(apt.Context(device=0), apt.Settings(), [good, other_rate, broken], apt.Rate.hz(48000), True)
    assert not isinstance(got[0], Exception)
    assert isinstance(got[1], apt.AptError) and "11025" in str(got[1]) and "48000" in str(got[1])
    assert isinstance(got[2], apt.AptError) and str(got[2]) and "status" not in str(got[2])
    # ... the same text load() / decode_wav() give for that file
    with pytest.raises(apt.AptError) as e:
        apt.wav_parse(broken)
    assert str(got[2]) == str(e.value)


def test_decode_batch_one_per_call_neighbours_differ(oracle):
    """recordings_per_call == 1 with three calls in flight: the decode of call c+1 may not touch the result record the
    download of call c is still copying.  Batch-worker plans therefore get one stream and slot set per call in flight
    whatever `recordings_per_call` is (a depth-1 plan has a single record: a recording could be reported with its
    neighbour's status / row count).  Every recording here differs from its neighbours in the worker's (longest-first)
    order in length and row count, too-short ones (whose record is written by a tiny kernel right at the head of
    their stream) follow the last good one; repeated, one and two workers: the window is a few microseconds wide."""
    base = synth_apt(48000, 24, 700)
    recs = [np.ascontiguousarray(base[:int(48000 * (11.0 + 0.53 * i))]) for i in range(20)]
    wants = [oracle.decode(x, 48000, True) for x in recs]
    assert len({w.size for w in wants}) >= 10
    recs += [np.zeros(20000 + 100 * i, f32) for i in range(6)]
    for rep in range(6):
        got, results, _ = apt.decode_batch(apt.Context(device=0), apt.Settings(), recs, apt.Rate.hz(48000), True,
                                           devices=(0,) if rep % 2 == 0 else (0, 0), recordings_per_call=1, return_stats=True)
        for i, g in enumerate(got):
            if i >= len(wants):
                assert isinstance(g, apt.InternalError), (rep, i, g)
                continue
            assert not isinstance(g, Exception), (rep, i, g)
            assert _same(g, wants[i]), (rep, i)
            assert results[i].n_rows == wants[i].size // 2080


def test_worker_affinity_lookup_on_this_host():
    """The real lookup: the GPU has a PCI address; whatever sysfs says about its NUMA node, a worker that pins itself
    (or finds nothing to pin to) still decodes — the batch tests above ran that way."""
    bdf, node, cpus = apt.host_affinity(0)
    assert len(bdf.split(":")) == 3, bdf
    assert (node < 0 and cpus == "") or (node >= 0 and cpus != "")


def _rank_decode(rank, world, port, recs, out_dir):
    """One rank of a two-process job on ONE GPU (the torchrun shape minus the second device): its shard of the
    recordings through aptgpu_decode_batch with two worker entries on device 0, its own process, HIP context and
    session cache; rows to files, bookkeeping over gloo."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from noaa_apt_amd import shard
        mine = shard.my_shard([r.size for r in recs], rank, world)
        dist.barrier()
        got, results, stats = apt.decode_batch(apt.Context(device=0), apt.Settings(), [recs[i] for i in mine],
                                               apt.Rate.hz(48000), True, devices=(0, 0), recordings_per_call=2,
                                               return_stats=True)
        for i, g in zip(mine, got):
            assert not isinstance(g, Exception), (rank, i, g)
            np.save(os.path.join(out_dir, f"rows_{i}.npy"), g)
        elapsed, total = shard.reduce_job(1.0 + rank, float(stats.samples))
        counts = shard.gather_counts(len(mine))
        info = apt.cache_info()
        np.save(os.path.join(out_dir, f"meta_{rank}.npy"),
                np.array([elapsed, total, counts[0], counts[1], stats.workers, stats.sessions_created, len(mine)], np.float64))
        del info
    finally:
        dist.destroy_process_group()


def test_two_processes_share_one_gpu(oracle, tmp_path):
    """SURVEY.md §8(e) on a one-GPU box: two spawned ranks, each a process with its own HIP context and session cache,
    decode disjoint shards on the same device at the same time; every recording's rows bit-identical to the oracle's."""
    import socket
    import torch.multiprocessing as mp
    recs = [synth_apt(48000, 11 + (5 * i) % 6, 700 + i, ppm=5.0 * i - 10.0) for i in range(7)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_decode, args=(r, 2, port, recs, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(2)]
    for m in metas:
        assert m[0] == 2.0                                   # MAX over ranks of (1.0, 2.0)
        assert m[1] == float(sum(r.size for r in recs))      # whole-job samples
        assert m[2] + m[3] == len(recs) and m[4] == 2        # both shards counted; two workers per rank
    for i, x in enumerate(recs):
        got = np.load(tmp_path / f"rows_{i}.npy")
        assert _same(got, oracle.decode(x, 48000, True)), f"recording {i}"
