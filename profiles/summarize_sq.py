"""Per-kernel averages of every counter in a rocprofv3 --pmc pass (rocpd sqlite).  usage: python profiles/summarize_sq.py <db> [out.csv]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events group by name, counter_name").fetchall()
tab = {}
for name, cn, n, v, d in rows:
    k = name.split("(")[0].replace("void ", "")
    tab.setdefault(k, {"launches": n, "avg_us": d / 1e3})[cn] = v
cols = sorted({cn for _, cn, _, _, _ in rows})
lines = ["kernel,launches,avg_us," + ",".join(cols)]
for k in sorted(tab, key=lambda k: -tab[k]["launches"] * tab[k]["avg_us"]):
    if "k_" not in k[:24]:
        continue
    lines.append("\"%s\",%d,%.1f," % (k, tab[k]["launches"], tab[k]["avg_us"]) + ",".join("%.4g" % tab[k].get(cn, float("nan")) for cn in cols))      # quoted: template argument lists contain commas
print("\n".join(lines))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
    import json
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-3d-pack_amd"))
    try:
        import c3d_hip
        dig = c3d_hip.code_digest()
    except Exception:
        dig = None
    json.dump({"code_digest": dig}, open(sys.argv[2] + ".meta.json", "w"))      # sidecar: which kernel code these counters describe
