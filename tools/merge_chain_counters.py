#!/usr/bin/env python3
"""profiles/rNN_sq_counters_chain.json from the four per-kernel files tools/collect_sq_kernel.sh wrote
(rNN_sq_counters_{words,slots,orbit,gather}.json) and the front end's (rNN_sq_counters_strict.json: one recording per
launch, scaled to the call's 16).

    python tools/merge_chain_counters.py r04 [note]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    note = sys.argv[2] if len(sys.argv) > 2 else ""
    p = os.path.join(ROOT, "profiles")
    kernels = {k: json.load(open(os.path.join(p, f"{rnd}_sq_counters_{k}.json"))) for k in ("words", "slots", "orbit", "gather")}
    fe = json.load(open(os.path.join(p, f"{rnd}_sq_counters_strict.json")))
    valu = {k: int(v["per_launch"]["SQ_INSTS_VALU"]) for k, v in kernels.items()}
    valu["chain_total"] = sum(valu.values())
    valu["front_end_16_recordings"] = int(fe["per_launch"]["SQ_INSTS_VALU"]) * 16
    stamps = {v.get("csrc_sha16") for v in kernels.values()}
    out = {
        "note": ("SQ counters of the kernels behind the front end per call of 16 recordings (strict, one call in flight): the four "
                 "per-kernel files of tools/collect_sq_kernel.sh in one (the name the round-3 review asked for)" + (". " + note if note else "")),
        "kernels": kernels,
        "csrc_sha16": stamps.pop() if len(stamps) == 1 else None,
        "csrc_groups_sha16": next(iter(kernels.values())).get("csrc_groups_sha16"),
        "valu_wave_instructions_per_call": valu,
        "chain_share_of_front_end": round(valu["chain_total"] / valu["front_end_16_recordings"], 4),
    }
    json.dump(out, open(os.path.join(p, f"{rnd}_sq_counters_chain.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("csrc_sha16", "valu_wave_instructions_per_call", "chain_share_of_front_end")}))


if __name__ == "__main__":
    main()
