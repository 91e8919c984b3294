import json, sys
d0 = sys.argv[1]
for f in sys.argv[2:] or ("strict", "fast", "c3"):
    try:
        d = json.load(open(f"{d0}/bench_{f}.json"))
        r = d["roofline"]
        print(f, "value", d["value"], "ms/step", d["ms_per_step"], "k_ms", r["kernel_avg_ms"], "frac", r["frac"],
              "single", (r.get("single_recording_launch") or {}).get("frac"), d["pipeline"]["kernels_alone_ms"], d.get("parity", ""))
    except Exception as e:
        print(f, "ERR", e)
