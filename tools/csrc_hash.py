#!/usr/bin/env python3
"""sha256 (first 16 hex digits) of the kernel sources under noaa_apt_amd/csrc (*.hip, *.hpp, *.cpp, Makefile) — the
stamp that ties a committed profile (profiles/r03_hbm_traffic_*.json, r03_sq_counters_*.json) to the code it was
collected on; bench.py only quotes a profile whose stamp matches the sources it runs.  No git needed (the GPU box
has no .git)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    d = os.path.join(root, "noaa_apt_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
