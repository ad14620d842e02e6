"""FETCH_SIZE / WRITE_SIZE of profiles/microbench/pmc_calib.hip's kernels against the bytes each kernel is known to move -> calibration factors.
usage: python profiles/summarize_calib.py <fetch.db> <write.db> <known.json (stdout of pmc_calib)> <out.json>
factor = counter bytes / known bytes (counters are in KiB).  bench.py divides a kernel's raw counters by the factor of its dominant pattern."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(counter_value), avg(duration) from pmc_events where counter_name=? group by name", (counter,)).fetchall()
    return {r[0].split("(")[0].replace("void ", ""): (r[1], r[2] * 1024.0, r[3]) for r in rows}


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
known = json.loads([ln for ln in open(sys.argv[3]).read().splitlines() if ln.startswith("{")][-1])
out = {}
for k, kb in known.items():
    fe, wr = f.get(k, (0, 0.0, 0.0)), w.get(k, (0, 0.0, 0.0))
    out[k] = {"known_read_B": kb["read"], "known_write_B": kb["write"], "FETCH_SIZE_B": round(fe[1]), "WRITE_SIZE_B": round(wr[1]), "avg_us": round(fe[2] / 1e3, 1),
              "fetch_over_known": round(fe[1] / kb["read"], 4) if kb["read"] else None, "write_over_known": round(wr[1] / kb["write"], 4) if kb["write"] else None,
              "GBps_known": round((kb["read"] + kb["write"]) / max(fe[2], 1) , 1)}
    if "read_lines" in kb:
        out[k]["fetch_over_touched_lines"] = round(fe[1] / kb["read_lines"], 4)
out["_what"] = "counter bytes / known bytes per access pattern on gfx950 (rocprofv3, ROCm 7.2); see profiles/microbench/pmc_calib.hip"
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in out.items():
    print(k, v)
