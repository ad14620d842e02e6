"""Condense bench.py JSON lines (stdin) into one short line each -- used in gpurun sweeps."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(d["config"].get("view_lanes"), d["value"], d["ms_per_step"], {k: round(v["avg_ms"], 4) for k, v in d.get("kernels", {}).items() if isinstance(v, dict) and "avg_ms" in v})
