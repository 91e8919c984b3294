#!/usr/bin/env python3
"""Issue cost of the front end's instruction kinds in cycles of the clock the chip ACTUALLY ran at.

tools/ubench/rates*.hip priced instructions against an assumed 2.4 GHz in launches of a millisecond.  The SMU lowers the
gfx clocks once the package reaches its power limit (tools/power_regimes.py), so here every kind runs as a train of
~5 ms launches for ~150 ms (tools/ubench/rates_power.hip) with amd-smi sampled beside it: per launch the duration, the
clock of the sample nearest its middle, and from the two the cycles per wave-instruction and SIMD.

    python tools/rates_power.py [--wps 6] [--ms 150]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OPS = ["v_fma_f32 (vvv)", "v_add_f32_e32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mul_f32 sgpr op_sel", "v_mov_b32_e32"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wps", type=int, default=6, help="waves per SIMD")
    ap.add_argument("--ms", type=float, default=150.0, help="how long each kind runs")
    ap.add_argument("--idle-ms", type=float, default=300.0, help="pause between kinds")
    args = ap.parse_args()
    from noaa_apt_amd.testing.smu import SmuSampler

    lib = C.CDLL(os.path.join(ROOT, "tools", "ubench", "librates_power.so"))
    lib.rates_power_launch.restype = C.c_float
    lib.rates_power_launch.argtypes = [C.c_int, C.c_int, C.c_int]
    smu = SmuSampler(period_s=0.001)
    print(json.dumps({"smu_available": smu.available, "error": smu.error, "cap_w": smu.cap_w, "wps": args.wps}), flush=True)
    lib.rates_power_launch(0, args.wps, 64)  # context, code object
    for op, name in enumerate(OPS):
        # calibrate: ~5 ms per launch
        ms = lib.rates_power_launch(op, args.wps, 256)
        rep = max(64, int(256 * 5.0 / max(ms, 1e-3)))
        time.sleep(args.idle_ms * 1e-3)
        launches = []
        with smu:
            t_end = time.perf_counter() + args.ms * 1e-3
            while time.perf_counter() < t_end:
                a = time.perf_counter()
                ms = lib.rates_power_launch(op, args.wps, rep)
                launches.append((a, time.perf_counter(), ms))
        samples = list(smu.samples)

        def clock_at(t):
            best = min(samples, key=lambda s: abs(s["t"] - t)) if samples else None
            return (sum(best["gfxclks"]) / len(best["gfxclks"]), best.get("current_socket_power")) if best and "gfxclks" in best else (None, None)

        rows = []
        for (a, b, ms) in launches:
            clk, pw = clock_at(0.5 * (a + b))
            ns = ms * 1e6 / (rep * 64.0 * args.wps)
            rows.append({"ns_per_instr_per_simd": round(ns, 4), "gfxclk_mhz": round(clk, 1) if clk else None, "socket_w": pw,
                         "cycles_at_that_clock": round(ns * clk / 1e3, 3) if clk else None, "cycles_at_2400": round(ns * 2.4, 3)})
        first, last = rows[0], rows[-1]
        tail = rows[len(rows) // 2:]
        cyc = [r["cycles_at_that_clock"] for r in tail if r["cycles_at_that_clock"]]
        print(json.dumps({"op": name, "launches": len(rows), "rep": rep, "first_launch": first, "last_launch": last,
                          "second_half_mean": {"ns_per_instr_per_simd": round(sum(r["ns_per_instr_per_simd"] for r in tail) / len(tail), 4),
                                               "gfxclk_mhz": round(sum(r["gfxclk_mhz"] or 0 for r in tail) / len(tail), 1),
                                               "cycles_at_that_clock": round(sum(cyc) / len(cyc), 3) if cyc else None,
                                               "socket_w": round(sum(r["socket_w"] or 0 for r in tail) / len(tail), 1)},
                          "smu": smu.summary(skip_s=0.05)}), flush=True)


if __name__ == "__main__":
    main()
