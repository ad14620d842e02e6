"""The 8-view step of each elevation band of the 64-camera orbit on ONE GPU (rank r of an 8-GPU run renders cameras [8r, 8r + 8): ranks 0-1 elevation -30, 2-3 0, 4-5 30, 6-7 60):
how unequal are the ranks' steps?  usage: python profiles/microbench/band_step_times.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for r in range(8):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--targets", "off", "--cpu-baseline", "off", "--as-rank", str(r)],
                         capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    rl = d.get("roofline") or {}
    print("cameras of rank %d: %.3f ms per step, %.0f Mpx/s, compositing backward %.3f ms" % (r, d["ms_per_step"], d["value"], rl.get("avg_ms", 0.0)), flush=True)
