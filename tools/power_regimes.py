#!/usr/bin/env python3
"""What the power manager does while the decode pipeline runs: socket power, the gfx clocks the SMU reports, HBM clock,
temperatures and the throttlers' residency counters (amd-smi's gpu_metrics / violation status), sampled every few
milliseconds from a second thread while ONE process goes through the regimes

    idle -> isolated (one call at a time, host synchronisation after each) -> idle -> pipeline (three calls in flight,
    several seconds, the step time of every block of steps recorded) -> isolated again straight after -> idle ->
    pipeline in fast mode

tools/clock_regimes.py reads the shader clock from inside a wave (s_memtime against s_memrealtime, and the period of an
s_sleep loop): constant in every regime.  But the same isolated front end takes 0.68 ms right after a pipelined run and
0.61 ms 25 ms later (profiles/r04_pipeline_trace_final.txt's run): something that is not the clock counter slows the
chip down under sustained load.  This tool asks the SMU.

    python tools/power_regimes.py [--pipeline-s 4] [--batch 16] > gpurun_out/power_regimes.txt

Prints JSON lines: the static facts (power cap, clock range), one summary per regime (power mean / max, clocks min /
mean, temperatures, throttler residency deltas, ms per call — for the pipeline per block of steps), then the raw
samples (decimated) for the pipeline regime.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KEEP = ("average_socket_power", "current_socket_power", "current_gfxclk", "average_gfxclk_frequency", "current_uclk",
        "average_uclk_frequency", "current_socclk", "temperature_hotspot", "temperature_mem", "temperature_edge",
        "temperature_vrgfx", "throttle_status", "indep_throttle_status", "average_gfx_activity", "average_umc_activity",
        "energy_accumulator", "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc",
        "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "gfx_activity_acc",
        "mem_activity_acc", "voltage_gfx", "voltage_soc", "firmware_timestamp", "gfxclk_lock_status")


def _num(v):
    return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None


class Smu:
    """amd-smi if it works here, the hwmon files otherwise; every read is best effort."""

    def __init__(self):
        self.smi = None
        self.h = None
        self.err = []
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.smi, self.h = amdsmi, hs[0]
        except Exception as e:  # noqa: BLE001
            self.err.append(f"amdsmi: {type(e).__name__}: {e}")
        self.hwmon = {}
        try:
            base = "/sys/class/drm"
            for c in sorted(os.listdir(base)):
                hw = os.path.join(base, c, "device", "hwmon")
                if c.startswith("card") and "-" not in c and os.path.isdir(hw):
                    for h in os.listdir(hw):
                        for f in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
                            p = os.path.join(hw, h, f)
                            if os.path.exists(p):
                                self.hwmon[f] = p
                    break
        except Exception as e:  # noqa: BLE001
            self.err.append(f"hwmon: {type(e).__name__}: {e}")

    def static(self):
        out = {"errors": self.err, "hwmon_files": sorted(self.hwmon)}
        if self.smi:
            for name, fn in (("power_cap", lambda: self.smi.amdsmi_get_power_cap_info(self.h)),
                             ("power", lambda: self.smi.amdsmi_get_power_info(self.h)),
                             ("gfx_clock", lambda: self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.GFX)),
                             ("mem_clock", lambda: self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.MEM)),
                             ("metrics_header", lambda: self.smi.amdsmi_get_gpu_metrics_header_info(self.h))):
                try:
                    out[name] = {k: (v if _num(v) is not None or isinstance(v, (str, bool)) else str(v)) for k, v in fn().items()}
                except Exception as e:  # noqa: BLE001
                    out[name] = f"{type(e).__name__}: {e}"
        return out

    def sample(self):
        s = {"t": time.perf_counter()}
        if self.smi:
            try:
                m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
                for k in KEEP:
                    v = _num(m.get(k))
                    if v is not None:
                        s[k] = v
                g = [x for x in (m.get("current_gfxclks") or []) if _num(x) is not None and 0 < x < 60000]
                if g:
                    s["gfxclks"] = g[:8]
            except Exception as e:  # noqa: BLE001
                s["metrics_error"] = f"{type(e).__name__}: {e}"
        for f, p in self.hwmon.items():
            try:
                s["hwmon_" + f] = int(open(p).read().strip())
            except Exception:  # noqa: BLE001
                pass
        return s

    def violations(self):
        if not self.smi:
            return None
        try:
            v = self.smi.amdsmi_get_violation_status(self.h)
            out = {}
            for k, x in v.items():
                if k.startswith(("acc_", "active_", "per_")):
                    if isinstance(x, list):
                        flat = [y for row in x for y in (row if isinstance(row, list) else [row]) if _num(y) is not None]
                        out[k] = flat[:8]
                    elif _num(x) is not None or isinstance(x, bool):
                        out[k] = x
            return out
        except Exception as e:  # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"}


def summarise(samples, t0, t1):
    sel = [s for s in samples if t0 <= s["t"] <= t1]
    out = {"samples": len(sel)}
    if not sel:
        return out

    def col(k):
        return np.array([s[k] for s in sel if k in s], dtype=np.float64)

    for k in ("current_socket_power", "average_socket_power", "hwmon_power1_average", "hwmon_power1_input", "current_gfxclk",
              "average_gfxclk_frequency", "current_uclk", "temperature_hotspot", "temperature_mem", "voltage_gfx",
              "average_gfx_activity", "hwmon_freq1_input"):
        c = col(k)
        if c.size:
            out[k] = {"min": float(c.min()), "mean": round(float(c.mean()), 1), "max": float(c.max())}
    g = [s["gfxclks"] for s in sel if "gfxclks" in s]
    if g:
        a = np.array([x for x in g if len(x) == len(g[0])], dtype=np.float64)
        out["gfxclks_per_xcd"] = {"min": a.min(axis=0).tolist(), "mean": np.round(a.mean(axis=0), 1).tolist(), "max": a.max(axis=0).tolist()}
    for k in ("ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc",
              "prochot_residency_acc", "accumulation_counter", "energy_accumulator", "gfx_activity_acc"):
        c = col(k)
        if c.size >= 2:
            out["delta_" + k] = float(c[-1] - c[0])
    ts = sorted({int(s["throttle_status"]) for s in sel if "throttle_status" in s})
    if ts:
        out["throttle_status_values"] = ts
    its = sorted({int(s["indep_throttle_status"]) for s in sel if "indep_throttle_status" in s})
    if its:
        out["indep_throttle_status_values"] = its[:16]
    return out


def viol_delta(a, b):
    if not a or not b or "error" in a or "error" in b:
        return {"before": a, "after": b}
    d = {}
    for k, x in b.items():
        y = a.get(k)
        if k.startswith("acc_"):
            if isinstance(x, list) and isinstance(y, list):
                dd = [p - q for p, q in zip(x, y)]
                if any(dd):
                    d[k] = dd
            elif _num(x) is not None and _num(y) is not None and x != y:
                d[k] = x - y
        elif k.startswith("active_"):
            if (isinstance(x, list) and any(x)) or (not isinstance(x, list) and x):
                d[k] = x
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--pipeline-s", type=float, default=4.0)
    ap.add_argument("--isolated-s", type=float, default=1.5)
    ap.add_argument("--idle-s", type=float, default=1.0)
    ap.add_argument("--block", type=int, default=100, help="steps per timed block of the pipelined regime")
    ap.add_argument("--period-ms", type=float, default=4.0)
    ap.add_argument("--inputs", type=int, default=16)
    args = ap.parse_args()

    import torch
    import noaa_apt_amd as apt
    from noaa_apt_amd.testing.synth import synth_apt

    smu = Smu()
    print(json.dumps({"static": smu.static()}), flush=True)
    dev = torch.device("cuda", 0)
    B = args.batch
    xs = [synth_apt(48000, 600.0, seed=2 + 1000 * j) for j in range(args.inputs)]
    n = xs[0].size
    d_xs = [torch.from_numpy(v).to(dev) for v in xs]
    S = 3
    plans = {m: apt.Plan(apt.Settings(), apt.Rate.hz(48000), True, max_samples=n, max_batch=B, mode=mm)
             for m, mm in (("strict", apt.MODE_STRICT), ("fast", apt.MODE_FAST))}
    cap = int(plans["strict"].info.max_rows)
    outs = [[torch.empty(cap * 2080, dtype=torch.float32, device=dev) for _ in range(B)] for _ in range(S)]
    k = [0]

    def step(plan):
        j = k[0]
        k[0] += 1
        sig = [d_xs[(j * B + b) % args.inputs].data_ptr() for b in range(B)]
        plan.decode_device(sig, [n] * B, [t.data_ptr() for t in outs[j % S]], [cap] * B)

    for p in plans.values():
        for _ in range(10):
            step(p)
    torch.cuda.synchronize()

    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smu.sample())
            time.sleep(args.period_ms * 1e-3)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    regimes = []

    def run(name, fn):
        v0 = smu.violations()
        t0 = time.perf_counter()
        extra = fn() or {}
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        v1 = smu.violations()
        regimes.append((name, t0, t1, extra, viol_delta(v0, v1)))

    def idle():
        time.sleep(args.idle_s)

    def isolated(plan, seconds, max_calls=100000):
        def f():
            ms = []
            t_end = time.perf_counter() + seconds
            while time.perf_counter() < t_end and len(ms) < max_calls:
                a = time.perf_counter()
                step(plan)
                torch.cuda.synchronize()
                ms.append(1e3 * (time.perf_counter() - a))
            q = len(ms) // 4
            return {"calls": len(ms), "ms_per_call_first_quarter": round(float(np.mean(ms[:max(1, q)])), 4),
                    "ms_per_call_last_quarter": round(float(np.mean(ms[-max(1, q):])), 4),
                    "ms_per_call_first_10": [round(x, 3) for x in ms[:10]], "ms_per_call_last_10": [round(x, 3) for x in ms[-10:]]}
        return f

    def pipeline(plan, seconds):
        def f():
            blocks = []
            t_end = time.perf_counter() + seconds
            while time.perf_counter() < t_end:
                a = time.perf_counter()
                for _ in range(args.block):
                    step(plan)
                plan.synchronize()
                blocks.append(1e3 * (time.perf_counter() - a) / args.block)
            return {"blocks": len(blocks), "steps_per_block": args.block, "ms_per_step_by_block": [round(x, 4) for x in blocks],
                    "ms_per_step_first_block": round(blocks[0], 4), "ms_per_step_last_quarter": round(float(np.mean(blocks[-max(1, len(blocks) // 4):])), 4)}
        return f

    run("idle", idle)
    run("isolated_strict_cold", isolated(plans["strict"], args.isolated_s))
    run("idle_2", idle)
    run("pipeline_strict", pipeline(plans["strict"], args.pipeline_s))
    run("isolated_strict_straight_after", isolated(plans["strict"], 0.25))
    run("idle_3", idle)
    run("pipeline_fast", pipeline(plans["fast"], args.pipeline_s * 0.6))
    run("idle_4", idle)
    stop.set()
    th.join()
    T0 = regimes[0][1]
    for (name, t0, t1, extra, vd) in regimes:
        out = {"regime": name, "from_s": round(t0 - T0, 3), "to_s": round(t1 - T0, 3)}
        out.update(extra)
        out["smu"] = summarise(samples, t0, t1)
        out["throttler_changes"] = vd
        print(json.dumps(out), flush=True)
    # raw samples of the strict pipeline and what follows, decimated to <= 400 lines
    t0 = regimes[3][1] - 0.2
    t1 = regimes[5][2]
    sel = [s for s in samples if t0 <= s["t"] <= t1]
    stepn = max(1, len(sel) // 400)
    for s in sel[::stepn]:
        r = {"t_s": round(s["t"] - T0, 4)}
        for kk in ("current_socket_power", "average_socket_power", "current_gfxclk", "gfxclks", "current_uclk", "temperature_hotspot",
                   "temperature_mem", "throttle_status", "indep_throttle_status", "ppt_residency_acc", "socket_thm_residency_acc",
                   "accumulation_counter", "voltage_gfx", "hwmon_power1_average", "hwmon_power1_input", "hwmon_freq1_input", "metrics_error"):
            if kk in s:
                r[kk] = s[kk]
        print(json.dumps(r))
    for p in plans.values():
        p.close()


if __name__ == "__main__":
    main()
