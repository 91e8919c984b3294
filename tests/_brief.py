import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print(d["value"], d["ms_per_step"], d["pipeline"]["kernels_ms"], d.get("parity"))
