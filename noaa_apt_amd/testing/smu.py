"""Measurement helper (bench.py, tools/power_regimes.py): what the power manager reports while a workload runs.

amd-smi's gpu_metrics table — socket power, the gfx clock of every XCD, the accumulated residency of the power /
thermal throttlers — sampled from a second thread.  Not part of the product path; best effort (no amd-smi, no
permission: `SmuSampler.available` is False and `summary()` returns None).

Why it exists: the shader-clock counter a wave can read (s_memtime) ticks at a constant rate on this part, whatever the
real clock is; under the decode pipeline the package sits at its 1400 W limit and the SMU runs the XCDs at 1.95-2.1 GHz
instead of 2.4 (profiles/r04_power_regimes.txt).
"""
import threading
import time

_FIELDS = ("current_socket_power", "current_gfxclk", "current_uclk", "temperature_hotspot", "temperature_mem",
           "ppt_residency_acc", "socket_thm_residency_acc", "hbm_thm_residency_acc", "vr_thm_residency_acc",
           "prochot_residency_acc", "accumulation_counter", "energy_accumulator")


def _num(v):
    return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None


def _mean(xs):
    return sum(xs) / len(xs) if xs else None


class SmuSampler:
    """with SmuSampler(pci_bus_id=...) as s: ...work...;  s.summary() -> dict or None"""

    def __init__(self, pci_bus_id=None, device_index=0, period_s=0.002):
        self.period_s = period_s
        self.samples = []
        self.available = False
        self.error = None
        self._stop = threading.Event()
        self._thread = None
        self._smi = None
        self._h = None
        self.cap_w = None
        self.max_gfxclk = None
        try:
            import amdsmi
            try:
                amdsmi.amdsmi_init()
            except Exception as e:  # noqa: BLE001  (already initialised by another sampler: fine)
                if "already" not in str(e).lower():
                    raise
            handles = amdsmi.amdsmi_get_processor_handles()
            h = None
            if pci_bus_id is not None:
                for cand in handles:
                    try:
                        bdf = amdsmi.amdsmi_get_gpu_device_bdf(cand)  # "0000:05:00.0"
                        if int(bdf.split(":")[1], 16) == int(pci_bus_id):
                            h = cand
                            break
                    except Exception:  # noqa: BLE001
                        pass
            if h is None:
                h = handles[min(device_index, len(handles) - 1)]
            self._smi, self._h = amdsmi, h
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(h)["power_cap"]
                self.cap_w = cap / 1e6 if cap > 100000 else float(cap)  # reported in microwatts
            except Exception:  # noqa: BLE001
                pass
            try:
                self.max_gfxclk = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)["max_clk"]
            except Exception:  # noqa: BLE001
                pass
            self.available = bool(self._read())
        except Exception as e:  # noqa: BLE001
            self.error = f"{type(e).__name__}: {e}".replace("\n", " ").replace("\t", " ")

    def _read(self):
        """One reading of the metrics table: the raw ctypes struct (tens of microseconds of Python — the dictionary the
        high-level call builds holds the interpreter lock for milliseconds, which a thread enqueueing work beside it feels)."""
        s = {"t": time.perf_counter()}
        try:
            import ctypes
            w = self._smi.amdsmi_wrapper
            m = w.amdsmi_gpu_metrics_t()
            rc = w.amdsmi_get_gpu_metrics_info(self._h, ctypes.byref(m))
            if rc != 0:
                self.error = f"amdsmi_get_gpu_metrics_info: status {rc}"
                return None
            for k in _FIELDS:
                v = getattr(m, k, None)
                if isinstance(v, int) and v not in (0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF):
                    s[k] = v
            g = [int(x) for x in m.current_gfxclks if 0 < int(x) < 60000]
        except Exception as e:  # noqa: BLE001
            self.error = f"{type(e).__name__}: {e}".replace("\n", " ").replace("\t", " ")
            return None
        if g:
            s["gfxclks"] = g[:8]
        return s if len(s) > 1 else None

    def snapshot(self):
        """The accumulators now (energy, throttler residency): two of them bracket a region WITHOUT a thread beside it."""
        return self._read() if self.available else None

    @staticmethod
    def between(a, b):
        """Mean socket power and throttled share of the time between two snapshots (energy_accumulator counts 2^-16 J)."""
        if not a or not b or b["t"] <= a["t"]:
            return None
        out = {"window_s": round(b["t"] - a["t"], 5)}
        if b.get("accumulation_counter") == a.get("accumulation_counter"):
            # the driver hands out a cached copy of the table now and then: the second read saw the first one's data
            out["stale_table"] = True
            return out
        if "energy_accumulator" in a and "energy_accumulator" in b:
            out["socket_w_mean"] = round((b["energy_accumulator"] - a["energy_accumulator"]) * 2.0 ** -16 / (b["t"] - a["t"]), 1)
        ticks = b.get("accumulation_counter", 0) - a.get("accumulation_counter", 0)
        out["accumulator_ticks"] = ticks
        if ticks > 0:
            out["power_limit_throttled_frac"] = round((b.get("ppt_residency_acc", 0) - a.get("ppt_residency_acc", 0)) / ticks, 3)
        if "gfxclks" in b:
            out["gfxclk_mhz_at_end"] = round(_mean(b["gfxclks"]), 1)
        return out

    def _run(self):
        while not self._stop.is_set():
            s = self._read()
            if s:
                self.samples.append(s)
            time.sleep(self.period_s)

    def __enter__(self):
        self.samples = []
        self._stop.clear()
        if self.available:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        return False

    def summary(self, skip_s=0.0):
        """Means over the samples taken inside the `with` block (the first `skip_s` seconds left out)."""
        if not self.available or not self.samples:
            return None
        t0 = self.samples[0]["t"] + skip_s
        sel = [s for s in self.samples if s["t"] >= t0] or self.samples
        out = {"samples": len(sel), "window_s": round(sel[-1]["t"] - sel[0]["t"], 4)}
        p = [s["current_socket_power"] for s in sel if "current_socket_power" in s]
        if p:
            out["socket_w"] = {"mean": round(_mean(p), 1), "max": max(p)}
        if self.cap_w:
            out["cap_w"] = self.cap_w
        g = [s["gfxclks"] for s in sel if "gfxclks" in s]
        if g:
            n = min(len(x) for x in g)
            per = [_mean([x[i] for x in g]) for i in range(n)]
            out["gfxclk_mhz"] = {"mean_over_xcds": round(_mean(per), 1), "slowest_xcd": round(min(per), 1),
                                 "fastest_xcd": round(max(per), 1)}
        else:
            c = [s["current_gfxclk"] for s in sel if "current_gfxclk" in s]
            if c:
                out["gfxclk_mhz"] = {"mean_over_xcds": round(_mean(c), 1)}
        if self.max_gfxclk:
            out["max_gfxclk_mhz"] = self.max_gfxclk
        u = [s["current_uclk"] for s in sel if "current_uclk" in s]
        if u:
            out["hbm_clk_mhz"] = round(_mean(u), 1)
        for k in ("temperature_hotspot", "temperature_mem"):
            v = [s[k] for s in sel if k in s]
            if v:
                out[k + "_c"] = max(v)
        a, b = sel[0], sel[-1]
        ticks = (b.get("accumulation_counter", 0) - a.get("accumulation_counter", 0))
        if ticks > 0:
            out["throttled_frac"] = {
                "power_limit": round((b.get("ppt_residency_acc", 0) - a.get("ppt_residency_acc", 0)) / ticks, 3),
                "socket_thermal": round((b.get("socket_thm_residency_acc", 0) - a.get("socket_thm_residency_acc", 0)) / ticks, 3),
                "hbm_thermal": round((b.get("hbm_thm_residency_acc", 0) - a.get("hbm_thm_residency_acc", 0)) / ticks, 3),
            }
        return out
