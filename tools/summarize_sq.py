#!/usr/bin/env python3
"""Average rocprofv3 --pmc SQ_* counters per launch of one kernel over several counter_collection.csv files.

usage: summarize_sq.py <kernel-substring> <csv> [<csv> ...] [--note TEXT]
SQ cycle counters are in quad-cycles (MI355X_MICROARCH.md §Per-instruction cycle constants)."""
import collections
import csv
import json
import sys


def main():
    args = sys.argv[1:]
    note = ""
    if "--note" in args:
        i = args.index("--note")
        note = args[i + 1]
        del args[i:i + 2]
    pat, files = args[0], args[1:]
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    per = {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())}
    d = {}
    w = per.get("SQ_WAVES")
    if w and "SQ_INSTS_VALU" in per:
        d["valu_instructions_per_wave"] = round(per["SQ_INSTS_VALU"] / w)
    if w and "SQ_WAVE_CYCLES" in per:
        d["wave_lifetime_quad_cycles"] = round(per["SQ_WAVE_CYCLES"] / w)
        for k, name in (("SQ_ACTIVE_INST_VALU", "valu_active"), ("SQ_WAIT_ANY", "waiting_waitcnt_or_barrier"),
                        ("SQ_WAIT_INST_ANY", "issue_stalled"), ("SQ_ACTIVE_INST_ANY", "issuing_any"),
                        ("SQ_ACTIVE_INST_LDS", "lds_active"), ("SQ_ACTIVE_INST_SCA", "scalar_active")):
            if k in per:
                d[name + "_fraction_of_wave_time"] = round(per[k] / per["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_IDX_ACTIVE" in per and "SQ_LDS_BANK_CONFLICT" in per and per["SQ_LDS_IDX_ACTIVE"]:
        d["lds_bank_conflict_fraction_of_lds_cycles"] = round(per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"], 4)
    out = {"note": note, "launches_averaged": max((len(v) for v in agg.values()), default=0),
           "per_launch": per, "derived": d}
    # what bench.py's roofline.valu quotes: the 100 %-issue floor of one launch — every VALU instruction keeps
    # its SIMD's pipe busy for SQ_ACTIVE_INST_VALU quad-cycles in all, over 256 CUs x 4 SIMDs at 2.4 GHz
    if w and "SQ_ACTIVE_INST_VALU" in per:
        out["valu_instructions_per_wave"] = d.get("valu_instructions_per_wave")
        out["waves_per_launch"] = w
        out["issue_floor_us"] = round(per["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.4e3, 2)
        out["source"] = "SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz, one recording per launch (tools/collect_sq.sh)"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
