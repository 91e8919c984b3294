#!/usr/bin/env python3
"""bench.py — Msamples/s of WAV->APT-line decode on MI355X, and % of the HBM roofline.

One "step" = one pass of the decode() hot path (resample -> AM envelope -> low-pass -> sync
correlation + peak picker -> row gather) over one batch of synthetic input already resident in
HBM: `--batch` (default 16) independent recordings of BASELINE.json configs[1] — synthetic 48 kHz
APT, 10 min (28.8 M samples), `standard` profile — decoded by ONE aptgpu_plan_decode_device call,
i.e. one launch per stage over the sixteen recordings (1.8 GB of input per step).  Every GPU works on its own batch (weak
scaling, no collective on the data path; recordings never talk to each other).  `--batch 1` is
the recording-by-recording shape of round 1.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (HIP events recorded on
the plan's stream inside the timed region); `pipeline` is the same algorithmic-bytes figure
over the whole step; `cpu_baseline` is the CPU oracle (the C restatement of the reference's
scalar loops, 1 thread — the reference decode is single-threaded) timed on this host on the
same recording, and it doubles as a bit-exact parity check of the GPU output.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)


class _QuietStdout:
    """Rank 0 prints ONE JSON line on stdout — and nothing else may: RCCL's version banner and gloo's connection
    notice are written to file descriptor 1 by native code.  Everything between construction and emit() goes to
    stderr instead."""

    def __init__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, text):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        print(text, flush=True)
        os.dup2(2, 1)


def config4_recordings(count, rate=48000, seconds=900.0, distinct=4):
    """The recording list of BASELINE config 4: `count` independent 15-minute recordings, seeds
    1000..., per-recording sample-rate error uniform in +-50 ppm and a random start phase so that lengths and sync
    positions differ.  Synthesising 256 x 43.2 M samples would take a quarter of an hour of numpy: `distinct` base
    recordings (each with its own seed and rate error) are generated, and recording i is base i % distinct started
    at a random sample (a rotation: another start phase, another row alignment) and cut to its own length.
    Returns (lengths, make(i) -> f32 array)."""
    from noaa_apt_amd.testing.synth import synth_apt
    rng = np.random.default_rng(4)
    ppm = rng.uniform(-50.0, 50.0, size=count)
    n_nom = int(round(rate * seconds))
    lengths = [int(round(n_nom * (1.0 + p * 1e-6))) for p in ppm]
    shift = rng.integers(0, n_nom, size=count)
    slack = int(n_nom * 60e-6) + 8
    bases = {}

    def make(i):
        j = i % distinct
        if j not in bases:
            bases[j] = synth_apt(rate, seconds + slack / rate, seed=1000 + j, ppm=float(ppm[j]))
        b = bases[j]
        return np.ascontiguousarray(np.roll(b, -int(shift[i]))[:lengths[i]])

    return lengths, make


def sustained_front_end(args, batch):
    """The dominant kernel launched back to back with NOTHING else on the GPU: the probe build of the library
    (`make -C noaa_apt_amd/csrc probe-lib`: libaptgpu_probe.so, in which APTGPU_DEBUG_SKIP=7 leaves the picker and the
    gather out of a call — the rows are then garbage, so this runs in a child process of its own) through tools/sweep.py,
    three calls in flight, front ends ordered by events as in the product.  Returns ms per launch (the period of the
    front ends: their duration plus the hand-over between two of them), or None when the probe library is not there."""
    import subprocess
    probe = os.path.join(ROOT, "noaa_apt_amd", "libaptgpu_probe.so")
    if not os.path.exists(probe) or args.profile != "standard" or args.mode not in ("strict", "fast"):
        return None
    env = dict(os.environ, APTGPU_PROBE_LIB=probe, APTGPU_DEBUG_SKIP="7")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "sweep.py"), "--configs", f"{args.mode}:{batch}:3", "--steps", "80",
           "--warmup", "10", "--inputs", str(batch), "--rate", str(args.rate), "--seconds", str(args.seconds)]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240).stdout
        for line in out.splitlines():
            d = json.loads(line)
            if "config" in d:
                return {"ms_per_launch": round(d["ms_per_recording"] * batch, 5),
                        "kernel_ms_one_at_a_time_in_that_process": d["alone_ms_per_call"].get("fused_front_end"),
                        "how": "libaptgpu_probe.so, APTGPU_DEBUG_SKIP=7 (front ends only), tools/sweep.py, 80 calls, three in flight"}
    except Exception as e:  # noqa: BLE001 - informational leg
        return {"error": str(e)}
    return None


def run_config4(args):
    """bench.py --config4: see the argument's help.  One JSON line on rank 0."""
    json_out = _QuietStdout()
    import torch
    import noaa_apt_amd as apt
    from noaa_apt_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    visible = torch.cuda.device_count()
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    G = max(1, args.node_gpus)
    rate_hz, seconds = 48000, 900.0
    lengths, make = config4_recordings(args.recordings, rate_hz, seconds)
    shares = shard.assign(lengths, G)  # the node's sharding, whatever is visible here
    # which of the G shares this process measures: one per rank under torchrun, one per visible device otherwise
    if dist is not None:
        mine = [(rank, local_rank)] if rank < G else []
    else:
        mine = [(g, g) for g in range(min(G, visible))]
    settings = tuned_settings(apt, args)
    rate = apt.Rate.hz(rate_hz)
    mode = {"strict": apt.MODE_STRICT, "generic": apt.MODE_GENERIC, "fp16taps": apt.MODE_FP16_TAPS, "fast": apt.MODE_FAST}[args.mode]
    B = max(1, args.batch)
    per_device = []
    for share_idx, dev_idx in mine:
        idx = shares[share_idx]
        recs = [make(i) for i in idx]
        samples = float(sum(r.size for r in recs))
        torch.cuda.set_device(dev_idx)
        dev = torch.device("cuda", dev_idx)
        # (i) device-resident: the share's inputs in HBM, decoded in calls of B recordings
        d_x = [torch.from_numpy(r).to(dev) for r in recs]
        n_max = max(r.size for r in recs)
        plan = apt.Plan(settings, rate, True, max_samples=n_max, max_batch=B, device=dev_idx, mode=mode)
        cap = int(plan.info.max_rows)
        depth = 3
        outs = [[torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in range(B)] for _ in range(depth)]
        torch.cuda.synchronize()

        def run_share():
            k = 0
            for a in range(0, len(recs), B):
                b = min(len(recs), a + B)
                plan.decode_device([t.data_ptr() for t in d_x[a:b]], [t.numel() for t in d_x[a:b]],
                                   [t.data_ptr() for t in outs[k % depth][:b - a]], [cap] * (b - a))
                k += 1

        for _ in range(2):
            run_share()
        torch.cuda.synchronize()
        reps = max(1, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(reps):
            run_share()
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / reps
        res0 = plan.results(1)[0]
        plan.close()
        del d_x, outs
        torch.cuda.empty_cache()
        # (ii) host-fed: the same recordings from pageable host memory through aptgpu_decode_batch
        ctx = apt.Context(device=dev_idx, mode=mode)
        apt.decode_batch(ctx, settings, recs[:min(len(recs), 2 * B)], rate, True, devices=(dev_idx,), recordings_per_call=B)
        got, hres, hst = apt.decode_batch(ctx, settings, recs, rate, True, devices=(dev_idx,), recordings_per_call=B,
                                          return_stats=True)
        ok = all(not isinstance(g, Exception) for g in got)
        moved = hst.h2d_bytes + hst.d2h_bytes
        per_device.append({
            "share_of_gpu": share_idx, "device": dev_idx, "recordings": len(idx), "samples": samples,
            "device_resident": {"seconds": round(t_dev, 5), "value": round(samples / t_dev / 1e6, 1), "unit": "Msamples/s"},
            "host_fed": {"seconds": round(hst.seconds, 5), "value": round(hst.samples / hst.seconds / 1e6, 1),
                         "unit": "Msamples/s", "pcie_GBps": round(moved / hst.seconds / 1e9, 2),
                         "frac_of_pcie_peak": round(moved / hst.seconds / 63e9, 4), "all_decoded": ok},
            "rows_of_first_recording": int(res0.n_rows),
        })
        apt.cache_clear()
    # whole-job figures: every share runs concurrently on its own GPU, so the job takes as long as its slowest share
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_device)
        per_device = [d for lst in gathered for d in lst]
    if rank == 0:
        measured = sorted(d["share_of_gpu"] for d in per_device)
        t_res = max(d["device_resident"]["seconds"] for d in per_device)
        t_host = max(d["host_fed"]["seconds"] for d in per_device)
        s_meas = sum(d["samples"] for d in per_device)
        line = {
            "metric": "Msamples/sec WAV->APT-line decode", "unit": "Msamples/s", "higher_is_better": True,
            "scaling": "strong", "dtype": "f32", "data": "synthetic", "vs_baseline": None,
            "config": {"workload": f"BASELINE config 4: {args.recordings} independent 15-minute 48 kHz recordings "
                                   f"(+-50 ppm, random start phase), sharded longest-first over {G} GPUs "
                                   f"(noaa_apt_amd.shard.assign), {B} recordings per call, mode={args.mode}",
                       "shares_measured": measured,
                       "shares_not_measured": [g for g in range(G) if g not in measured],
                       "visible_gpus": visible if dist is None else world},
            # the measured shares run side by side: aggregate = their samples over the slowest of them
            "value": round(s_meas / t_res / 1e6, 1), "n_gpus": len(measured),
            "value_host_fed": round(s_meas / t_host / 1e6, 1),
            "note": "value = device-resident aggregate over the measured shares; value_host_fed = the same from pageable "
                    "host memory (PCIe-inclusive, one worker per GPU).  Shares of GPUs that are not visible are not "
                    "extrapolated.",
            "per_device": per_device,
        }
        json_out.emit(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def tuned_settings(apt, args):
    """The profile's Settings with the --set overrides applied."""
    settings = apt.Settings.profile(args.profile)
    for kv in args.set:
        k, _, v = kv.partition("=")
        if not hasattr(settings, k):
            raise SystemExit(f"--set: no Settings field {k!r}")
        setattr(settings, k, type(getattr(settings, k))(float(v)))
    return settings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--pre-steps", type=int, default=48, help="untimed steps before the per-kernel measurement passes")
    ap.add_argument("--settle-steps", type=int, default=40,
                    help="untimed pipelined steps between the per-kernel measurement passes and the W warm-up steps")
    ap.add_argument("--no-power", action="store_true", help="do not sample amd-smi's power / clock metrics beside the timed region")
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--profile", default="standard")
    ap.add_argument("--set", action="append", default=[], metavar="FIELD=VALUE",
                    help="override a Settings field of the profile, e.g. --set resample_atten=31 (default_settings.toml is a "
                         "user-editable file: which kernel a tuned filter lands on is part of the product)")
    ap.add_argument("--mode", default="strict", choices=["strict", "generic", "fp16taps", "fast"],
                    help="strict (default): bit-exact; fast: APTGPU_MODE_FAST, tolerance of SURVEY.md §8(d), "
                         "checked against the oracle after the timed region")
    ap.add_argument("--inputs", type=int, default=4,
                    help="distinct recordings resident in HBM, decoded round-robin; at least --batch of them are "
                         "generated (16 x 115 MB per step: far beyond the 256 MB Infinity Cache, so every step "
                         "reads its input from HBM)")
    ap.add_argument("--batch", type=int, default=16,
                    help="recordings per decode_device call = per step (default 16: one launch per stage covers the "
                         "sixteen recordings; BASELINE config 4's per-GPU share is --seconds 900 --batch 32); `value` "
                         "counts all their samples")
    ap.add_argument("--user-stream", action="store_true",
                    help="experiment: give the plan torch's stream as ctx.stream (every call then waits for "
                         "an event recorded there; the inputs are synchronised before timing anyway)")
    ap.add_argument("--no-sync", action="store_true",
                    help="experiment: decode(sync=false) — front end + final resample only, no peak picker")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed-for-the-headline extra legs (PCM16 ingest, image stage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-launch", action="store_true",
                    help="skip the extra one-recording-per-launch measurement (profiling runs: every front-end "
                         "launch of the process then has the bench's launch shape)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="experiment: leave the per-kernel HIP events out of the timed region")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE config 4 as written: --recordings (default 256) independent 15-minute 48 kHz recordings "
                         "sharded over 8 GPUs with shard.assign(); every visible GPU / rank decodes the share of one of the "
                         "eight, device-resident and host-fed; shares of GPUs that are not there are marked 'not measured'")
    ap.add_argument("--recordings", type=int, default=256)
    ap.add_argument("--node-gpus", type=int, default=8, help="GPUs config 4 shards over (BASELINE.json: 8)")
    args = ap.parse_args()
    if args.config4:
        return run_config4(args)

    json_out = _QuietStdout()
    import torch
    import noaa_apt_amd as apt
    from noaa_apt_amd.testing.synth import synth_apt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # The barriers that bracket the timed region are host-side (a gloo group over the loopback interface): every
    # rank has synchronised its GPU before it arrives, so nothing on a device is left to wait for — and an RCCL
    # barrier (an all-reduce + stream synchronisation through the proxy thread) measured 2.4 ms on this node,
    # 12 % of a 20-step run.  The job's figures are still reduced over RCCL (shard.reduce_job).  Falls back to
    # the default group's barrier if gloo cannot be set up.
    cpu_group = None
    if dist is not None:
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        try:
            cpu_group = dist.new_group(backend="gloo")
        except Exception:
            cpu_group = None

    def job_barrier():
        if dist is not None:
            dist.barrier(group=cpu_group) if cpu_group is not None else dist.barrier()

    # (the group's first collective sets its connections up — milliseconds; done here, not between the warm-up's
    # synchronize and t0, where the GPU would sit drained meanwhile: see the SMU snapshot below)
    job_barrier()
    job_barrier()

    settings = tuned_settings(apt, args)
    rate = apt.Rate.hz(args.rate)

    # ---- synthetic recording (seeded per rank), moved to HBM before any timing
    # recording 0 is the one checked against the oracle; the others differ in seed (noise, image)
    n_inputs = max(1, args.inputs, args.batch)
    xs = [synth_apt(args.rate, args.seconds, seed=2 + rank + 1000 * j) for j in range(n_inputs)]
    x = xs[0]
    n = x.size
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        d_xs = [torch.from_numpy(v).to(dev) for v in xs]
        d_x = d_xs[0]
        mode = {"strict": apt.MODE_STRICT, "generic": apt.MODE_GENERIC, "fp16taps": apt.MODE_FP16_TAPS,
                "fast": apt.MODE_FAST}[args.mode]
        B = max(1, args.batch)
        plan = apt.Plan(settings, rate, not args.no_sync, max_samples=n, max_batch=B, device=local_rank,
                        mode=mode, stream=stream.cuda_stream if args.user_stream else 0)
        cap = int(plan.info.max_rows)
        d_rows_all = [torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in range(B)]
        d_rows = d_rows_all[0]
        torch.cuda.synchronize()  # inputs are resident before anything is enqueued on the plan's streams
        # call j decodes the B recordings j, j+1, ... (mod n_inputs); recording 0 of call 0 is checked
        sigs = [[d_xs[(j + b) % n_inputs].data_ptr() for b in range(B)] for j in range(n_inputs)]
        nn, out, caps = [n] * B, [t.data_ptr() for t in d_rows_all], [cap] * B
        counter = [0]

        def step(j=None):
            if j is None:
                j = counter[0] % n_inputs
                counter[0] += 1
            plan.decode_device(sigs[j], nn, out, caps)

        # (amd-smi is opened here, well before the timed region: its initialisation takes a while, and a pause between
        # the warm-up steps and the timed steps would let the power manager's averages relax)
        smu = None
        if not args.no_power:  # (every rank watches its own device: the first 8-GPU run should explain itself)
            from noaa_apt_amd.testing.smu import SmuSampler
            smu = SmuSampler(pci_bus_id=getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None), device_index=local_rank)

        # ---- per-kernel measurement passes, BEFORE the warm-up steps (they are part of every run; placed
        # here they also bring the GPU out of the idle clocks it fell to while the host generated inputs)
        # (a) the dominant kernel with nothing else on the GPU — one step at a time, a host
        # synchronisation after each, HIP events around every launch.  This per-launch duration
        # is what `roofline` uses (a launch in the timed region below shares the GPU with the
        # launches of the other calls in flight, so its duration there is not GPU time per launch).
        for _ in range(args.pre_steps):  # (the host spent seconds generating inputs: wake the GPU before timing anything)
            step()
        torch.cuda.synchronize()
        plan.enable_timing(2)
        for _ in range(24):
            step()
            torch.cuda.synchronize()
        iso_times = plan.collect_timing()
        # (b) the same kernel launched over ONE recording (the launch shape of round 1), one launch at a time
        single_ms = None
        if B > 1 and not args.no_sync and not args.no_single_launch:
            plan1 = apt.Plan(settings, rate, True, max_samples=n, max_batch=1, device=local_rank, mode=mode)
            plan1.enable_timing(2)
            for j in range(16):
                plan1.decode_device([d_xs[j % n_inputs].data_ptr()], [n], out[:1], caps[:1])
                torch.cuda.synchronize()
            single_ms = plan1.collect_timing().get("fused_front_end", (None, 0))[0]
            plan1.close()
        # (c) pipelined, with events around every kernel, for the per-kernel breakdown.  This pass is also what the
        # power manager sees last before the warm-up steps: passes (a) and (b) leave the package 40 % idle, its power
        # average relaxes, and a pipelined loop started then runs its first ~5 ms at up to 2.4 GHz, the next ~10 ms
        # BELOW the settled clock (the controller overshoots) — `--settle-steps` pipelined steps (default 40: ~35 ms)
        # put the loop into the state a long run is in before W and K are counted
        plan.enable_timing(2)
        for _ in range(10):
            step()
        ktimes = plan.collect_timing()
        plan.enable_timing(0)
        # what the power manager does meanwhile (every rank, its own device): the SMU's accumulators read before and after
        # a window of pipelined steps — no thread beside it — and a sampled loop of the same steps behind it.  The first
        # snapshot is taken HERE, before the settle and warm-up steps, not between the warm-up's synchronize and t0 (round
        # 6: a gpu_metrics read takes milliseconds; with the GPU drained the package's power average relaxes meanwhile and
        # the K timed steps start on the controller's transient — 0.850-0.855 ms per step in the driver's 20-step shape
        # against 0.836-0.839 without the read, profiles/r06_bench_steps20_snapshot_ab.txt).  The window is therefore
        # settle + W + K steps of the same loop (`power.window`), and joules per call is its energy over that many calls.
        snap0 = smu.snapshot() if smu is not None else None
        for _ in range(args.settle_steps):
            step()

        # ---- W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        # timed region: HIP events bracket the dominant kernel of every 8th step (the markers
        # serialise the stream for ~6 us each; sampling keeps the measurement live but cheap)
        plan.enable_timing(0 if args.no_kernel_timing else 1)
        job_barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_enq = time.perf_counter()  # host finished enqueueing (informational)
        torch.cuda.synchronize()
        t_local = time.perf_counter()  # this rank's own K steps (the job's time is taken behind the barrier)
        job_barrier()
        t1 = time.perf_counter()
        dom_times = plan.collect_timing()
        plan.enable_timing(0)
        power = None
        if smu is not None:
            power = {"source": "amd-smi gpu_metrics, per rank (this object: rank 0's device): energy and throttler-residency accumulators read before the settle + "
                               "warm-up steps and after the timed steps (`window`: all of them, the same pipelined loop; nothing is read between the warm-up's "
                               "synchronize and t0); socket power and per-XCD gfx clocks sampled every 2 ms during the loop behind it",
                     "window": smu.between(snap0, smu.snapshot())}
            if power["window"] is not None:
                power["window"]["steps"] = args.settle_steps + args.warmup + args.steps
            if smu.available:
                # the same loop again for ~0.4 s under the sampler, the first 0.1 s left out: what the pipeline looks like
                # to the power manager once it has settled
                k3 = max(args.steps, int(0.4 / max(1e-4, (t1 - t0) / args.steps)))
                with smu:
                    a3 = time.perf_counter()
                    for _ in range(k3):
                        step()
                    torch.cuda.synchronize()
                    b3 = time.perf_counter()
                power["settled"] = smu.summary(skip_s=0.1)
                if power["settled"] is not None:
                    power["settled"]["steps"] = k3
                    power["settled"]["ms_per_step"] = round(1e3 * (b3 - a3) / k3, 5)
            else:
                power["error"] = smu.error
        step(0)  # the recording that is compared with the oracle below
        res = plan.results(1)[0]
        ref_pos = plan.sync_positions(0) if not args.no_sync else np.zeros(0, np.uint64)
        torch.cuda.synchronize()
        ref_rows = d_rows[:res.n_out].clone()

        # ---- extra legs (not part of `value`): the rows either side of the decode path
        extras = {}
        if not args.no_extras and not args.no_sync and args.mode == "strict" and res.status == 0:
            k2 = max(8, min(args.steps, 100))

            def timed_loop(fn):
                for j in range(4):
                    fn(j % n_inputs)
                torch.cuda.synchronize()
                a = time.perf_counter()
                for j in range(k2):
                    fn(j % n_inputs)
                torch.cuda.synchronize()
                return (time.perf_counter() - a) / k2

            # (1) WAV ingest: mono PCM16 payloads resident in HBM, converted inside the front end
            d_pcm = [torch.from_numpy(v.astype(np.int16)).to(dev) for v in xs]
            spec = apt.WavSpec(1, 16, 2, 0, args.rate, 1, 0, 2 * n, n, n)
            pcm_sigs = [[d_pcm[(j + b) % n_inputs].data_ptr() for b in range(B)] for j in range(n_inputs)]
            t_pcm = timed_loop(lambda j: plan.decode_device_wav(pcm_sigs[j], [spec] * B, out, caps)) / B
            plan.decode_device_wav(pcm_sigs[0], [spec] * B, out, caps)
            r2 = plan.results(1)[0]
            torch.cuda.synchronize()
            same = bool(r2.n_out == res.n_out and torch.equal(d_rows[:r2.n_out], ref_rows))
            b_pcm = 2.0 * n + 4.0 * 2080.0 * r2.n_rows
            extras["pcm16_ingest"] = {
                "what": "same recordings as mono PCM16 WAV payloads in HBM (2 B/sample), int16 -> f32 "
                        "inside the fused front end (wav.rs:30-51 + decode())",
                "ms_per_recording": round(1e3 * t_pcm, 5),
                "value": round(n / t_pcm / 1e6, 3), "unit": "Msamples/s",
                "algorithmic_bytes": b_pcm,
                "pipeline_frac_of_hbm_peak": round(b_pcm / t_pcm / 1e9 / HBM_PEAK_GBS, 5),
                "rows_identical_to_f32_input": same,
            }
            # (2) decode + image stage chained on the device: 98 % contrast limits -> u8 image
            d_imgs = [torch.empty(cap * 2080, dtype=torch.uint8, device=dev) for _ in range(B)]
            img_ptrs = [t.data_ptr() for t in d_imgs]

            def decode_and_image(j):
                plan.decode_device(sigs[j], nn, out, caps)
                plan.process_device(out, caps, apt.Contrast.Percent(0.98), img_ptrs)

            t_img = timed_loop(decode_and_image) / B
            plan.enable_timing(2)
            for j in range(8):
                decode_and_image(j % n_inputs)
            itimes = plan.collect_timing()
            plan.enable_timing(0)
            decode_and_image(0)
            ires = plan.image_results(1)[0]
            extras["decode_plus_image"] = {
                "what": "decode() then misc::percent(0.98) + map_signal_u8 on the device (noaa_apt.rs:132-192)",
                "ms_per_recording": round(1e3 * t_img, 5),
                "value": round(n / t_img / 1e6, 3), "unit": "Msamples/s",
                "image_kernels_ms": {k: round(v[0], 5) for k, v in sorted(itimes.items()) if k.startswith("image_")},
                "low": float(ires.low), "high": float(ires.high), "height": int(ires.height),
            }
            step(0)
            plan.results(1)
            # (3) host-fed: BASELINE config 4's per-GPU share (32 recordings of 15 minutes, ragged, +-50 ppm) in ordinary
            # (pageable) host memory -> aptgpu_decode_batch, which keeps uploads, kernels and downloads of consecutive
            # calls in flight together; once as f32 Signals, once as PCM16 WAV file images (half the PCIe bytes).
            # End-to-end, PCIe-inclusive: never part of `value`.
            from noaa_apt_amd.testing.wavfile import make_wav
            from noaa_apt_amd import shard as _shard
            PCIE_GBS = 63.0  # PCIe Gen5 x16 (MI355X_MICROARCH.md)
            if (args.rate, args.profile) == (48000, "standard"):
                c4_len, c4_make = config4_recordings(256)
                c4_idx = _shard.assign(c4_len, 8)[0]
                host_recs = [c4_make(i) for i in c4_idx]
                what = (f"BASELINE config 4's share of one GPU ({len(host_recs)} x 15 min at 48 kHz, +-50 ppm, of 256 "
                        f"sharded over 8) from pageable host memory")
            else:
                host_recs = [xs[j % n_inputs] for j in range(16)]
                what = f"{len(host_recs)} recordings of this run from pageable host memory"
            host_wavs = [make_wav(v.astype(np.int16), args.rate) for v in host_recs]
            ref_host = apt.decode(apt.Context(device=local_rank, mode=mode), settings, host_recs[0], rate, True)
            HOST_RUNS = 5
            for key, inputs, workers in (("host_fed_f32", host_recs, 1), ("host_fed_f32_two_workers", host_recs, 2),
                                         ("host_fed_pcm16_wav", host_wavs, 1), ("host_fed_pcm16_wav_two_workers", host_wavs, 2)):
                # warm-up: the whole list once (host pages touched; every worker has leased — i.e. created — its own
                # session: with a shorter list one worker can finish before the other starts and share its session,
                # and the second session is then built inside the timed call)
                apt.decode_batch(apt.Context(device=local_rank, mode=mode), settings, inputs, rate, True,
                                 devices=(local_rank,) * workers, recordings_per_call=B)
                # HOST_RUNS timed runs: a single pair of runs swung by 2x between boxes (round 3); the median is the figure,
                # the minimum and maximum say how far a run can be from it
                runs = []
                for _ in range(HOST_RUNS):
                    got, hres, hst = apt.decode_batch(apt.Context(device=local_rank, mode=mode), settings, inputs, rate, True,
                                                      devices=(local_rank,) * workers, recordings_per_call=B,
                                                      return_stats=True)
                    ok = all(not isinstance(g, Exception) for g in got)
                    same0 = bool(ok and np.array_equal(got[0].view(np.uint32), ref_host.view(np.uint32)))
                    runs.append({"seconds": hst.seconds, "ok": same0, "gate_wait": hst.gate_wait_seconds, "setup": hst.setup_seconds,
                                 "created": int(hst.sessions_created), "pinned": int(hst.workers_pinned)})
                    moved, n_samples = hst.h2d_bytes + hst.d2h_bytes, hst.samples
                    del got
                secs = sorted(r["seconds"] for r in runs)
                med, best, worst = secs[len(secs) // 2], secs[0], secs[-1]
                extras[key] = {
                    "what": f"{what}, {workers} worker(s) on this GPU, {B} recordings per call; rows DMA'd into the "
                            f"caller's buffers; median of {HOST_RUNS} runs",
                    "seconds": round(med, 5), "seconds_min": round(best, 5), "seconds_max": round(worst, 5),
                    "value": round(n_samples / med / 1e6, 3), "unit": "Msamples/s",
                    "pcie_bytes": int(moved),
                    "pcie_GBps": round(moved / med / 1e9, 2),
                    "frac_of_pcie_peak": round(moved / med / 1e9 / PCIE_GBS, 4),
                    "frac_of_pcie_peak_best_run": round(moved / best / 1e9 / PCIE_GBS, 4),
                    "rows_identical_to_one_shot_decode": all(r["ok"] for r in runs),
                    # what the workers waited for the per-device upload gate / spent leasing their session, per run (summed
                    # over the workers); sessions built inside a timed run (0 after the warm-up); workers pinned to their
                    # GPU's NUMA node
                    "upload_gate_wait_s": [round(r["gate_wait"], 4) for r in runs],
                    "setup_s": [round(r["setup"], 4) for r in runs],
                    "sessions_created_in_timed_runs": sum(r["created"] for r in runs),
                    "workers_pinned_to_numa_node": runs[0]["pinned"],
                    "session_cache_after": list(apt.cache_info()),  # (idle sessions, their device bytes)
                }
            extras["host_affinity"] = dict(zip(("pci", "numa_node", "cpulist"), apt.host_affinity(local_rank)))
            # (4) the one-shot aptgpu_decode() of recording 0 (plan, buffers and staging from the session cache)
            for _ in range(3):
                apt.decode(apt.Context(device=local_rank, mode=mode), settings, x, rate, True)
            shots = []
            for _ in range(11):
                o0 = time.perf_counter()
                apt.decode(apt.Context(device=local_rank, mode=mode), settings, x, rate, True)
                shots.append(1e3 * (time.perf_counter() - o0))
            shots.sort()
            extras["one_shot_decode"] = {
                "what": f"aptgpu_decode() of one {args.seconds:g} s recording in pageable host memory, rows back on the host "
                        f"(PCIe floor of its {4 * n / 1e6:.0f} MB: {4 * n / 57e6:.2f} ms); median of 11 calls",
                "ms": round(shots[len(shots) // 2], 3), "ms_min": round(shots[0], 3), "ms_max": round(shots[-1], 3)}
            del host_recs, host_wavs
            apt.cache_clear()
        pflags = plan.read_internal("picker_flags", np.uint32, 32)
        sustained = None
        if not args.no_extras and not args.no_sync and world == 1:
            torch.cuda.synchronize()
            sustained = sustained_front_end(args, B)

        # ---- the same K steps in APTGPU_MODE_FAST (reported NEXT TO the strict headline, never as `value`):
        # same inputs, same batch, same warm-up / timing protocol, its own isolated kernel timing, and its
        # tolerance check against the oracle further down
        fast_leg = None
        if args.mode == "strict" and not args.no_extras and not args.no_sync and rank == 0:
            planf = apt.Plan(settings, rate, True, max_samples=n, max_batch=B, device=local_rank, mode=apt.MODE_FAST)
            if planf.info.fused in (1, 3, 4):
                def fstep(j):
                    planf.decode_device(sigs[j % n_inputs], nn, out, caps)
                for j in range(24):
                    fstep(j)
                torch.cuda.synchronize()
                planf.enable_timing(2)
                for j in range(16):
                    fstep(j)
                    torch.cuda.synchronize()
                f_iso = planf.collect_timing()
                planf.enable_timing(0)
                for j in range(args.warmup):
                    fstep(j)
                torch.cuda.synchronize()
                f0 = time.perf_counter()
                for j in range(args.steps):
                    fstep(j)
                torch.cuda.synchronize()
                f1 = time.perf_counter()
                fstep(0)
                fres = planf.results(1)[0]
                fpos = planf.sync_positions(0)
                torch.cuda.synchronize()
                fast_leg = {"elapsed": f1 - f0, "alone_ms": f_iso.get("fused_front_end", (0.0, 0))[0],
                            "rows": d_rows[:fres.n_out].clone(), "pos": fpos, "n_rows": int(fres.n_rows)}
            planf.close()
            step(0)  # d_rows holds the strict rows of recording 0 again
            plan.results(1)

    from noaa_apt_amd import shard
    # whole-job figures: MAX elapsed over ranks, SUM of samples over ranks (no other collective)
    elapsed, total_samples_per_step = shard.reduce_job(t1 - t0, float(n) * max(1, args.batch), device=dev)
    # what each rank saw (bookkeeping, outside the timed region): its own K steps, its device's power and clocks in the
    # settled loop, joules per call, and where its process and device sit
    try:
        aff = dict(zip(("pci", "numa_node", "cpulist"), apt.host_affinity(local_rank)))
    except Exception as e:  # noqa: BLE001
        aff = {"error": str(e)}
    settled = (power or {}).get("settled") or {}
    timed = (power or {}).get("window") or {}
    mine = {"rank": rank, "device": local_rank, "ms_per_step": round(1e3 * (t_local - t0) / args.steps, 5),
            "socket_w": (settled.get("socket_w") or {}).get("mean"),
            "gfxclk_mhz": (settled.get("gfxclk_mhz") or {}).get("mean_over_xcds"),
            "joules_per_call": (round(timed["socket_w_mean"] * timed["window_s"] / timed["steps"], 4)
                                if timed.get("socket_w_mean") and timed.get("window_s") and timed.get("steps") else None),
            "pci": aff.get("pci"), "numa_node": aff.get("numa_node"),
            "cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    per_rank = [mine]
    if dist is not None and world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=cpu_group) if cpu_group is not None else dist.all_gather_object(per_rank, mine)

    if res.status != 0:
        raise SystemExit(f"decode failed on rank {rank}: status {res.status} reason {res.reason}")

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_samples_per_step * args.steps / elapsed / 1e6
        # algorithmic bytes of one recording: every input f32 read once, every output pixel
        # written once: 4*N_in + 4*2080*rows
        b_alg = (4.0 * n + 4.0 * 2080.0 * res.n_rows) * max(1, args.batch)  # per call (= per front-end launch)
        src = dom_times if dom_times else ktimes
        dom = max(src.items(), key=lambda kv: kv[1][0]) if src else ("none", (0.0, 0))
        dom_ms = dom[1][0]                                   # in the timed region: overlapped with other launches
        alone_ms, alone_n = iso_times.get(dom[0], (0.0, 0))  # one launch at a time: GPU time per launch
        achieved = b_alg / (alone_ms * 1e-3) / 1e9 if alone_ms > 0 else 0.0
        kernel_sum_ms = sum(v[0] for v in ktimes.values())
        pipe_achieved = b_alg / (ms_per_step * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel as measured with rocprofv3 PMC counters
        # (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 2x read correction applied by
        # tools/summarize_pmc.py) for THIS workload and mode; null when no matching profile is committed
        traffic, traffic_src, sq = None, None, None
        if (args.rate, args.seconds, args.profile, max(1, args.batch)) == (48000, 600.0, "standard", 16) \
                and dom[0] == "fused_front_end" and args.mode in ("strict", "fast"):
            # committed counter summaries are only quoted when they were collected on THESE kernel sources
            # (tools/collect_profiles.sh stamps them with tools/csrc_hash.py's hash of noaa_apt_amd/csrc)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                from csrc_hash import csrc_sha16, matches as stamp_matches
                here = csrc_sha16(ROOT)
            except Exception:
                here = None
                stamp_matches = None
            import glob

            def newest_profile(stem):
                """The latest round's profiles/rNN_<stem>.json whose stamp matches the sources here (else the latest one)."""
                cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}.json")), reverse=True)
                first = None
                for fpath in cands:
                    try:
                        d_ = json.load(open(fpath))
                    except Exception:
                        continue
                    first = first or (fpath, d_)
                    if here and stamp_matches(d_, "front_end", ROOT):
                        return fpath, d_, True
                return (first[0], first[1], False) if first else (None, None, False)

            try:
                f, d_t, ok_t = newest_profile(f"hbm_traffic_{args.mode}")
                rel = os.path.relpath(f, ROOT) if f else None
                if ok_t:
                    traffic = d_t["per_launch"]["k_fused"]["hbm_total_MB"] * 1e6
                    traffic_src = (f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                   f"collected on these kernel sources: csrc {d_t.get('csrc_sha16')})")
                elif f:
                    traffic_src = (f"{rel} is stamped {d_t.get('csrc_sha16')}, the kernel "
                                   f"sources here hash to {here}: not quoted")
            except Exception:
                traffic = None
            try:
                _, d_s, ok_s = newest_profile(f"sq_counters_{args.mode}")
                sq = d_s if ok_s else None
            except Exception:
                sq = None
        # arithmetic the front end EXECUTES per work-rate sample: two flops per FIR tap, ~8 for the envelope,
        # 22 for the sync correlation from pulse sums (every mode: the strict front end only bounds the
        # reference's 114-term chain, which the picker then evaluates for ~3 % of the positions), +3 for the
        # strict mode's sum of |F| behind those bounds
        w_len = float(res.work_len)
        taps1 = plan.info.n_resample_taps / plan.info.l
        corr_ops = 22.0 if args.mode == "fast" else 25.0
        flops = w_len * (2.0 * taps1 + 8.0 + 2.0 * plan.info.n_lowpass_taps + corr_ops) * max(1, args.batch)
        valu = {
            "note": "the kernel is bound by VALU issue, not HBM (DESIGN.md §5.1): arithmetic the front end executes "
                    "per launch over the isolated kernel time, against the 157.3 TFLOP/s fp32 vector peak",
            "algorithmic_flops_per_launch": flops,
            "achieved_tflops": round(flops / (alone_ms * 1e-3) / 1e12, 3) if alone_ms > 0 else None,
            "pct_of_fp32_vector_peak": round(100.0 * flops / (alone_ms * 1e-3) / 157.3e12, 2) if alone_ms > 0 else None,
        }
        if sq:
            valu.update({k: sq[k] for k in ("valu_instructions_per_wave", "waves_per_launch", "issue_floor_us",
                                            "source") if k in sq})
            if alone_ms > 0 and "issue_floor_us" in sq:
                # (the counters were collected on single-recording launches: the floor of a launch over
                # `batch` recordings is that many times as long)
                valu["issue_floor_us_per_launch"] = round(sq["issue_floor_us"] * max(1, args.batch), 2)
                valu["achieved_frac_of_issue_floor"] = round(sq["issue_floor_us"] * max(1, args.batch) / (alone_ms * 1e3), 4)
                # the floor above is priced at 2.4 GHz; in the pipelined loop the package is at its power limit and the
                # SMU runs the XCDs slower (`power.settled`): the same floor at the clock the pipeline actually gets
                clk = ((power or {}).get("settled") or {}).get("gfxclk_mhz", {}).get("mean_over_xcds")
                if clk:
                    valu["issue_floor_us_per_launch_at_settled_clock"] = round(valu["issue_floor_us_per_launch"] * 2400.0 / clk, 2)
                    valu["settled_gfxclk_mhz"] = clk
        line = {
            "metric": "Msamples/sec WAV->APT-line decode",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "joules_per_call": mine["joules_per_call"],  # socket energy accumulator over `power.window` (settle + W + K pipelined steps) / that many calls: rank 0's device (every rank's: per_rank)
            "per_rank": per_rank,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic {args.rate} Hz APT recording, {args.seconds:g} s "
                            f"({n} samples) per GPU, profile={args.profile}: resample {plan.info.l}/"
                            f"{plan.info.m} ({plan.info.n_resample_taps} taps) -> AM envelope -> "
                            f"{plan.info.n_lowpass_taps}-tap low-pass -> sync correlation + peak "
                            f"picker -> {res.n_rows} rows x 2080 px",
                "parallelism": f"{world} GPU(s), {max(1, args.batch)} independent recordings per GPU and step, no collectives",
                "recordings_per_call": max(1, args.batch),
                "mode": args.mode,
                "rows": int(res.n_rows),
                "n_sync": int(res.n_sync),
                "input_resident_in_hbm": True,
                "distinct_inputs_round_robin": n_inputs,
                "picker": {"path": {0: "lds", 1: "sequential-walk", 2: "global"}.get(int(pflags[1]), "?"),
                           "node_capacity": int(pflags[3]), "visited_nodes": int(pflags[4]), "orbit": {1: "direct", 2: "doubling (LDS)", 3: "doubling over all nodes (LDS, no closure)"}.get(int(pflags[6]), "doubling (L2)"),
                           "cycle_stamps": [int(v) for v in pflags[8:11]]},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom[0],
                # ONE launch at a time (host synchronisation between steps), HIP events on the launch's stream
                "kernel_avg_ms": round(alone_ms, 5),
                "launches_timed": int(alone_n),
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                # BASELINE.md section 3's own definition — algorithmic bytes over the time of the WHOLE device-resident
                # decode (every kernel, the timed region's ms_per_step) — next to the dominant kernel's burst figure
                "device_resident_frac": round(pipe_achieved / HBM_PEAK_GBS, 5),
                # the same kernel launched back to back (front ends only, nothing beside them) — sustained, where
                # kernel_avg_ms is one launch at a time with a host synchronisation after each
                "sustained": sustained,
                "sustained_frac": (round(b_alg / (sustained["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                   if sustained and sustained.get("ms_per_launch") else None),
                "traffic": traffic,
                "traffic_source": traffic_src,
                # what the HBM system actually moved during the launch (the PMC bytes over the same per-launch time)
                "traffic_GBps": round(traffic / (alone_ms * 1e-3) / 1e9, 2) if (traffic and alone_ms > 0) else None,
                "traffic_frac_of_peak": round(traffic / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if (traffic and alone_ms > 0) else None,
                "algorithmic_bytes_per_launch": b_alg,
                # in the timed region several calls are in flight and their front-end launches share the
                # GPU: a launch then lasts `avg_concurrent_launches` x the time the GPU spends on it
                "overlapped": {
                    "kernel_avg_ms": round(dom_ms, 5),
                    "calls_in_flight": int(os.environ.get("APTGPU_STREAMS", "0")) or "default",
                    "avg_concurrent_launches": round(dom_ms / ms_per_step, 3) if ms_per_step > 0 else None,
                },
                "valu": valu,
                # the same kernel over a single recording per launch (2400 tiles = 3.1 rounds of workgroups:
                # a fifth of its time is the partly filled last round, which a batch amortises)
                "single_recording_launch": None if not single_ms else {
                    "kernel_avg_ms": round(single_ms, 5),
                    "algorithmic_bytes_per_launch": b_alg / max(1, args.batch),
                    "frac": round(b_alg / max(1, args.batch) / (single_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            },
            # the pipelined loop runs at the package power limit: socket power, the gfx clocks the SMU grants, the share
            # of time the power throttler was active (None when amd-smi cannot be read)
            "power": power,
            "pipeline": {
                "achieved": round(pipe_achieved, 2),
                "unit": "GB/s",
                "frac": round(pipe_achieved / HBM_PEAK_GBS, 5),
                "sum_kernel_span_ms": round(kernel_sum_ms, 5),
                "host_enqueue_ms_per_step": round(1e3 * (t_enq - t0) / args.steps, 5),
                # event-to-event spans of every kernel in a SEPARATE 10-step pass of the pipelined loop: they include the time
                # a launch waits behind the other streams' work (queueing), so they overlap and exceed ms_per_step — a picture
                # of the overlap, not GPU time.  The GPU time of a kernel is kernels_alone_ms (one call in flight at a time).
                "kernels_span_in_pipeline_ms": {k: round(v[0], 5) for k, v in sorted(ktimes.items())},
                "kernels_alone_ms": {k: round(v[0], 5) for k, v in sorted(iso_times.items())},
            },
        }
        if extras:
            line["extras"] = extras
        if fast_leg:
            f_ms = 1e3 * fast_leg["elapsed"] / args.steps
            f_frac = b_alg / (fast_leg["alone_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if fast_leg["alone_ms"] > 0 else None
            line["fast_mode"] = {
                "what": "the same workload and protocol with the plan in APTGPU_MODE_FAST (f32; fused multiply-adds, "
                        "native sqrt, pulse-sum sync correlation; tolerance of SURVEY.md 8(d)) - reported next to the "
                        "strict headline, not part of `value`",
                "value": round(float(n) * max(1, args.batch) * args.steps / fast_leg["elapsed"] / 1e6, 3),
                "unit": "Msamples/s", "ms_per_step": round(f_ms, 5),
                "roofline": {"bound": "hbm", "kernel": "fused_front_end", "kernel_avg_ms": round(fast_leg["alone_ms"], 5),
                             "achieved": round(f_frac * HBM_PEAK_GBS, 2) if f_frac else None, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(f_frac, 5) if f_frac else None},
            }
        if not args.no_cpu_baseline:
            from oracle import binding as oracle
            os_ = {k: getattr(settings, k) for k in ("work_rate", "resample_atten",
                                                      "resample_delta_freq", "resample_cutout",
                                                      "demodulation_atten")}
            c0 = time.perf_counter()
            ref, st = oracle.decode(x, args.rate, not args.no_sync, settings=os_, want_steps=True)
            c1 = time.perf_counter()
            got = ref_rows.cpu().numpy()
            parity = bool(got.size == ref.size and
                          np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
            line["cpu_baseline"] = {
                "value": round(n / (c1 - c0) / 1e6, 3),
                "unit": "Msamples/s",
                "cores": 1,
                "kind": "port",
                "sample": f"the full rank-0 recording once ({n} samples, {c1 - c0:.2f} s): C "
                          f"restatement of the reference's scalar loops (no Rust toolchain here)",
                "host_cores_available": os.cpu_count(),
                "stage_seconds": {k: round(st[k], 4) for k in ("t_resample", "t_demod", "t_filter",
                                                                "t_sync", "t_gather")},
            }
            # one oracle instance per host core on the same recording (the reference decodes one file per
            # process; a batch driver would run one process per core) — bounded: every instance decodes once
            try:
                from concurrent.futures import ThreadPoolExecutor
                cores = max(1, min(os.cpu_count() or 1, 64))
                a0 = time.perf_counter()
                with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL inside the oracle
                    list(ex.map(lambda _: oracle.decode(x, args.rate, not args.no_sync, settings=os_), range(cores)))
                a1 = time.perf_counter()
                line["cpu_baseline"]["all_cores"] = {
                    "value": round(cores * n / (a1 - a0) / 1e6, 3), "unit": "Msamples/s", "cores": cores,
                    "sample": f"{cores} concurrent oracle instances, one decode of the same recording each ({a1 - a0:.2f} s)"}
            except Exception as e:  # noqa: BLE001 - informational leg
                line["cpu_baseline"]["all_cores"] = {"error": str(e)}
            if (args.rate, args.profile) == (48000, "standard") and args.seconds >= 600:
                # BASELINE config 3's rate on a bounded sample (5 of its 60 minutes)
                x3 = synth_apt(96000, 300, seed=3)
                b0 = time.perf_counter()
                oracle.decode(x3, 96000, True, settings=os_)
                b1 = time.perf_counter()
                line["cpu_baseline"]["config3_sample"] = {
                    "value": round(x3.size / (b1 - b0) / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                    "sample": f"96 kHz x 300 s ({x3.size} samples, {b1 - b0:.2f} s): 1/12 of config 3"}
            if extras:
                from oracle import image_binding as oimg
                i0 = time.perf_counter()
                oimg.process_gray(ref, oimg.CONTRAST_PERCENT, 0.98)
                line["cpu_baseline"]["image_stage_seconds"] = round(time.perf_counter() - i0, 4)
            if args.mode in ("fp16taps", "fast"):
                same_shape = got.size == ref.size
                err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))) if same_shape and ref.size else float("nan")
                if args.mode == "fp16taps":
                    line["parity"] = (f"fp16-tap mode: rows {'equal' if same_shape else 'DIFFER'} in count, "
                                      f"max |err| / max |px| = {err:.2e} (tolerance 2e-3)")
                else:
                    wp = st["sync_pos"].astype(np.int64)
                    gp = ref_pos.astype(np.int64)
                    pos_ok = gp.size == wp.size
                    off = np.abs(gp - wp) if pos_ok else np.array([99])
                    ok = bool(same_shape and pos_ok and off.max(initial=0) <= 1 and (off == 0).mean() >= 0.999
                              and (err <= 1e-4 or (off != 0).any()))
                    line["parity"] = (f"fast mode vs oracle: rows {'equal' if same_shape else 'DIFFER'} in count, "
                                      f"sync positions identical {float((off == 0).mean()):.5f} (max off {int(off.max(initial=0))}), "
                                      f"max |err| / max |px| = {err:.2e} (tolerance 1e-4): {'within' if ok else 'OUTSIDE'} tolerance")
            else:
                line["parity"] = "bit-exact vs oracle" if parity else "MISMATCH vs oracle"
            if fast_leg:
                fg = fast_leg["rows"].cpu().numpy()
                same_shape = fg.size == ref.size
                ferr = float(np.max(np.abs(fg - ref)) / np.max(np.abs(ref))) if same_shape and ref.size else float("nan")
                wp, gp = st["sync_pos"].astype(np.int64), fast_leg["pos"].astype(np.int64)
                off = np.abs(gp - wp) if gp.size == wp.size else np.array([99])
                ok = bool(same_shape and off.max(initial=0) <= 1 and (off == 0).mean() >= 0.999
                          and (ferr <= 1e-4 or (off != 0).any()))
                line["fast_mode"]["parity"] = (
                    f"rows {'equal' if same_shape else 'DIFFER'} in count, sync positions identical "
                    f"{float((off == 0).mean()):.5f} (max off {int(off.max(initial=0))}), max |err| / max |px| = {ferr:.2e} "
                    f"(tolerance 1e-4): {'within' if ok else 'OUTSIDE'} tolerance")
        json_out.emit(json.dumps(line))
    plan.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
