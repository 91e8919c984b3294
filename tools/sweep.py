#!/usr/bin/env python3
"""Pipeline sweep on one GPU: the same synthetic recordings through plans of different mode /
recordings-per-call / calls-in-flight, one process, inputs generated once.

    python tools/sweep.py --configs strict:1:6,strict:8:3,fast:1:6,fast:8:3 --steps 40

Each config is mode:batch:streams[:ENV=VALUE;ENV=VALUE...] — the optional fourth field sets environment switches of
the library (APTGPU_WORDS_DPP, APTGPU_GATHER_ITERS, APTGPU_FRONT_STREAM, ...: read at plan creation or per launch)
for that config only.  Prints one JSON line per config: ms per recording in the
pipelined loop, Msamples/s, and the per-kernel times with one call in flight at a time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="strict:1:6,fast:1:6")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--inputs", type=int, default=4)
    ap.add_argument("--pcm16", action="store_true")
    ap.add_argument("--profile", default="standard", help="settings profile (default_settings.toml): standard, fast, slow")
    ap.add_argument("--power", action="store_true", help="sample amd-smi beside the timed loop: socket power, gfx clocks, joules per call")
    ap.add_argument("--no-sync", action="store_true", help="decode(sync=false): front end without stage 4, no picker")
    args = ap.parse_args()

    import torch
    import noaa_apt_amd as apt
    if os.environ.get("APTGPU_PROBE_LIB"):  # (the probe build: APTGPU_DEBUG_* switches; timing only)
        apt.use_library(os.environ["APTGPU_PROBE_LIB"])
    from noaa_apt_amd.testing.synth import synth_apt

    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    xs = [synth_apt(args.rate, args.seconds, seed=2 + 1000 * j) for j in range(args.inputs)]
    n = xs[0].size
    if args.pcm16:
        d_xs = [torch.from_numpy(v.astype(np.int16)).to(dev) for v in xs]
    else:
        d_xs = [torch.from_numpy(v).to(dev) for v in xs]
    torch.cuda.synchronize()
    print(json.dumps({"inputs": args.inputs, "samples": n, "synth_s": round(time.perf_counter() - t0, 1)}), flush=True)
    modes = {"strict": apt.MODE_STRICT, "fast": apt.MODE_FAST, "fp16taps": apt.MODE_FP16_TAPS,
             "generic": apt.MODE_GENERIC}
    spec = apt.WavSpec(1, 16, 2, 0, args.rate, 1, 0, 2 * n, n, n)

    for cfg in args.configs.split(","):
        parts = cfg.split(":")
        mode_s, b_s, st_s = parts[:3]
        B, S = int(b_s), int(st_s)
        os.environ["APTGPU_STREAMS"] = str(S)
        extra_env = dict(kv.split("=", 1) for kv in parts[3].split(";") if kv) if len(parts) > 3 else {}
        saved_env = {k_: os.environ.get(k_) for k_ in extra_env}
        os.environ.update(extra_env)
        plan = apt.Plan(apt.Settings.profile(args.profile), apt.Rate.hz(args.rate), not args.no_sync, max_samples=n, max_batch=B, mode=modes[mode_s])
        cap = int(plan.info.max_rows)
        # every call in flight needs its own output buffers
        outs = [[torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in range(B)] for _ in range(S)]
        torch.cuda.synchronize()
        k = [0]

        def step():
            j = k[0]
            k[0] += 1
            sig = [d_xs[(j * B + b) % args.inputs].data_ptr() for b in range(B)]
            out = [t.data_ptr() for t in outs[j % S]]
            if args.pcm16:
                plan.decode_device_wav(sig, [spec] * B, out, [cap] * B)
            else:
                plan.decode_device(sig, [n] * B, out, [cap] * B)

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        smu = None
        if args.power:
            from noaa_apt_amd.testing.smu import SmuSampler
            smu = SmuSampler()
            s0 = smu.snapshot()
            smu.__enter__()
        a = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_enq = time.perf_counter()
        torch.cuda.synchronize()
        b = time.perf_counter()
        power = None
        if smu is not None:
            smu.__exit__(None, None, None)
            power = smu.summary(skip_s=min(0.1, 0.25 * (b - a)))
            bt = smu.between(s0, smu.snapshot())
            if power is not None and bt and "socket_w_mean" in bt:
                power["joules_per_call"] = round(bt["socket_w_mean"] * bt["window_s"] / args.steps, 4)
                power["socket_w_mean_by_energy_counter"] = bt["socket_w_mean"]
                power["power_limit_throttled_frac"] = bt.get("power_limit_throttled_frac")
        plan.enable_timing(2)
        for _ in range(8):
            step()
            torch.cuda.synchronize()
        alone = plan.collect_timing()
        plan.enable_timing(0)
        res = plan.results(B)
        # a checksum of the rows of every recording of the last call (A/B variants must agree bit for bit)
        torch.cuda.synchronize()
        last = outs[(k[0] - 1) % S]
        chk = 0
        for b_ in range(B):
            nb = int(res[b_].n_out)
            chk = (chk * 1000003 + int(last[b_][:nb].view(torch.int32).to(torch.int64).sum().item()) + nb) % (1 << 61)
        ms_rec = 1e3 * (b - a) / (args.steps * B)
        print(json.dumps({
            "config": cfg, "ms_per_recording": round(ms_rec, 5), "Msamples_per_s": round(n / ms_rec / 1e3, 1),
            "host_enqueue_ms_per_call": round(1e3 * (t_enq - a) / args.steps, 4),
            "status": [int(r.status) for r in res][:4], "rows": int(res[0].n_rows), "fused": int(plan.info.fused), "rows_checksum": chk,
            "alone_ms_per_call": {kk: round(v[0], 5) for kk, v in sorted(alone.items())},
            **({"power": power} if args.power else {}),
        }), flush=True)
        plan.close()
        for k_, v_ in saved_env.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_


if __name__ == "__main__":
    main()
