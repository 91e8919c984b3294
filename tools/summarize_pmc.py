#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) into per-kernel HBM bytes per
launch.  FETCH_SIZE on gfx950 counts 128-B requests as 64 B for coalesced streaming reads, so
the read side is doubled as MI355X_MICROARCH.md §HBM prescribes; WRITE_SIZE is taken as is
(calibrated here on k_gather_rows: 9.97 MB reported for 9.97 MB of output pixels).

usage: summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> [note]
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    base = m.group(1) if m else name[:40]
    if base.startswith("k_fused") and re.search(r",\s*short\s*[,>]", name):
        base += "_pcm16"  # the int16-input instantiation (WAV ingest inside the front end)
    return base


def avg(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "apt::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch = avg(sys.argv[1], "FETCH_SIZE")
    write = avg(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        out[k] = {"FETCH_SIZE_KB": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1),
                  "hbm_read_MB": round(rd / 1e6, 2), "hbm_write_MB": round(wr / 1e6, 2),
                  "hbm_total_MB": round((rd + wr) / 1e6, 2)}
    print(json.dumps({"note": sys.argv[3] if len(sys.argv) > 3 else "", "per_launch": out}, indent=1))


if __name__ == "__main__":
    main()
