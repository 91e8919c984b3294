#!/usr/bin/env python3
"""Calibrate a tools/summarize_pmc.py summary on the one kernel whose HBM output is known exactly.

usage: calibrate_pmc.py <summary.json> <recordings_per_launch> <rows_per_recording> <input_samples_per_recording>

On this pool the FETCH_SIZE / WRITE_SIZE passes of rocprofv3 under-report every kernel of a run by one
common factor (the counters of some XCC instances are missing from the sum).  k_gather_rows_call writes
exactly recordings x rows x 2080 x 4 bytes per launch: every figure is divided by (what WRITE_SIZE said for
it) / (that).  hbm_total_MB becomes the calibrated total; raw_total_MB keeps what the counters said.  Also
adds the front end's traffic per recording against the algorithmic bytes 4*N_in + 4*2080*rows.
"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    nrec, rows, n_in = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    per = d["per_launch"]
    known = nrec * rows * 2080 * 4 / 1e6
    g = per.get("k_gather_rows_call")
    factor = g["hbm_write_MB"] / known if g and g["hbm_write_MB"] > 0 else 1.0
    for k, v in per.items():
        v["raw_total_MB"] = v["hbm_total_MB"]
        v["calibrated_read_MB"] = round(v["hbm_read_MB"] / factor, 2)
        v["calibrated_write_MB"] = round(v["hbm_write_MB"] / factor, 2)
        v["hbm_total_MB"] = round(v["calibrated_read_MB"] + v["calibrated_write_MB"], 2)
    d["calibration"] = {"factor": round(factor, 4),
                        "how": f"k_gather_rows_call writes exactly {nrec} x {rows} x 2080 x 4 B = {known:.2f} MB per launch; "
                               f"WRITE_SIZE reported {g['hbm_write_MB'] if g else None} MB: every figure is divided by their ratio "
                               "(the counters of some XCC instances are missing from this pool's sums). hbm_total_MB is the "
                               "calibrated total; raw_total_MB what the counters said."}
    alg = (4.0 * n_in + 4.0 * 2080 * rows) / 1e6
    for name in ("k_fused",):
        if name in per:
            tot = per[name]["hbm_total_MB"] / nrec
            d["per_recording"] = {"k_fused_total_MB": round(tot, 2), "k_fused_read_MB": round(per[name]["calibrated_read_MB"] / nrec, 2),
                                  "k_fused_write_MB": round(per[name]["calibrated_write_MB"] / nrec, 2),
                                  "algorithmic_MB": round(alg, 2), "ratio": round(tot / alg, 3)}
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
