"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes: both do not fit the TCC slots of one pass).  Units: the counters are in KiB.
gfx950 correction from the guide: FETCH_SIZE reports exactly half of a wide coalesced streaming read (16 B/lane) --
both the raw and the x2-corrected figure are printed; WRITE_SIZE is uncalibrated.
usage: python profiles/summarize_pmc.py <fetch.db> <write.db> [out.json]
The JSON carries "_meta": {"code_digest": ...} (c3d_hip.code_digest()): bench.py refuses traffic figures measured on other kernel code."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "comfyui-3d-pack_amd"))


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(counter_value), avg(duration) from pmc_events where counter_name=? group by name", (counter,)).fetchall()
    return {r[0].split("(")[0].replace("void ", ""): (r[1], r[2], r[3]) for r in rows}


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
print("kernel,launches,fetch_MB_raw,fetch_MB_x2,write_MB,avg_us")
for k in sorted(f, key=lambda k: -f[k][1] * f[k][0]):
    if not (k.startswith("k_") or k.startswith("k_") or "k_" in k[:20]):
        continue
    fe = f[k][1] * 1024 / 1e6
    wr = w.get(k, (0, 0, 0))[1] * 1024 / 1e6
    out[k] = {"launches": f[k][0], "fetch_MB_raw": round(fe, 2), "fetch_MB_x2": round(2 * fe, 2), "write_MB": round(wr, 2), "avg_us": round(f[k][2] / 1e3, 1)}
    print("%s,%d,%.2f,%.2f,%.2f,%.1f" % (k, f[k][0], fe, 2 * fe, wr, f[k][2] / 1e3))
if len(sys.argv) > 3:
    try:
        import c3d_hip
        out["_meta"] = {"code_digest": c3d_hip.code_digest()}
    except Exception as e:      # never lose a measurement over the stamp
        out["_meta"] = {"code_digest": None, "error": repr(e)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
