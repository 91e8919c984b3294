import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["pipeline"].get("kernels_span_in_pipeline_ms", d["pipeline"].get("kernels_ms")),
          d["config"].get("picker"), d.get("parity"))
