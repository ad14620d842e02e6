"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes: both do not fit the TCC slots of one pass).  Units: the counters are in KiB.
gfx950 correction from the guide: FETCH_SIZE reports exactly half of a wide coalesced streaming read (16 B/lane) --
both the raw and the x2-corrected figure are printed; WRITE_SIZE is uncalibrated.
usage: python profiles/summarize_pmc.py <fetch.db> <write.db> [out.json] [mesh]
`mesh`: additionally sum the kernels of each bench.py profiling group of the DiffRastMesh workload ("_groups"; "long" = per long launch of each kernel, i.e. per
launch of the step over all its views), which is what bench.py --workload mesh turns into roofline_group.traffic per view.  Round 5: the plain averages mix one-view launches
(target rendering) with the step's eight-view launches and are neither per view nor per launch -- kept for the record, not reported any more.
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


def per_kernel_long(db, counter):
    """the same, over each kernel's LONG launches only (duration >= 0.6 x its longest): a bench run mixes launches that cover one view (target rendering) with the step's
    launches over all views of a group -- an average over both is neither a per-view nor a per-launch figure (round 5: the mesh line's "3.1 x algorithmic" was such an average)"""
    c = sqlite3.connect(db)
    rows = c.execute("select name, counter_value, duration from pmc_events where counter_name=?", (counter,)).fetchall()
    by = {}
    for name, val, dur in rows:
        by.setdefault(name.split("(")[0].replace("void ", ""), []).append((val, dur))
    out = {}
    for k, lst in by.items():
        top = max(d for _, d in lst)
        sel = [(v, d) for v, d in lst if d >= 0.6 * top]
        out[k] = (len(sel), sum(v for v, _ in sel) / len(sel), sum(d for _, d in sel) / len(sel))
    return out


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
fl = per_kernel_long(sys.argv[1], "FETCH_SIZE")
wl = per_kernel_long(sys.argv[2], "WRITE_SIZE")
out = {}
print("kernel,launches,fetch_MB_raw,fetch_MB_x2,write_MB,avg_us")
for k in sorted(f, key=lambda k: -f[k][1] * f[k][0]):
    if not (k.startswith("k_") or k.startswith("k_") or "k_" in k[:20]):
        continue
    fe = f[k][1] * 1024 / 1e6
    wr = w.get(k, (0, 0, 0))[1] * 1024 / 1e6
    out[k] = {"launches": f[k][0], "fetch_MB_raw": round(fe, 2), "fetch_MB_x2": round(2 * fe, 2), "write_MB": round(wr, 2), "avg_us": round(f[k][2] / 1e3, 1)}
    if k in fl:      # per LONG launch (the step's launches over all views of a group)
        out[k]["long"] = {"launches": fl[k][0], "fetch_MB_raw": round(fl[k][1] * 1024 / 1e6, 2), "write_MB": round(wl.get(k, (0, 0, 0))[1] * 1024 / 1e6, 2), "avg_us": round(fl[k][2] / 1e3, 1)}
    print("%s,%d,%.2f,%.2f,%.2f,%.1f" % (k, f[k][0], fe, 2 * fe, wr, f[k][2] / 1e3))
MESH_GROUPS = {      # kernels of bench.py's profiling groups; round 3 added the fused pixel passes (k_view_pixel_fwd, k_aa2_pairs, k_view_shade_fwd_g, k_view_loss_shade_bwd, k_view_tex_bwd)
    "mesh_rasterize": ("k_ras_tri", "k_ras_big", "k_ras_resolve", "k_view_transform_fwd"), "mesh_interpolate": ("k_interp_fwd", "k_view_pixel_fwd"), "mesh_texture": ("k_tex_fwd",),
    "mesh_antialias": ("k_view_sigmoid_seed", "k_aa2_fwd", "k_aa_fwd", "k_aa2_pairs", "k_view_shade_fwd_g"),
    "other": ("k_view_shade_fwd", "k_view_shade_bwd", "k_view_loss_shade_bwd", "k_shade_fwd", "k_shade_bwd", "k_mesh_pixel_loss", "k_mesh_sum", "k_mesh_hwc_to_chw", "k_tex_acc_finalize"),
    "mesh_antialias_bwd": ("k_aa2_bwd", "k_view_sigmoid_bwd", "k_aa_bwd"), "mesh_texture_bwd": ("k_tex_bwd_tiled", "k_tex_bwd", "k_view_tex_bwd"), "mesh_interpolate_bwd": ("k_interp_bwd",),
    "mesh_rasterize_bwd": ("k_ras_bwd_tri", "k_ras_bwd_big", "k_vertex_gather4", "k_vertex_gather4_heavy", "k_view_transform_bwd")}
if len(sys.argv) > 4 and sys.argv[4] == "mesh":
    groups = {}
    for g, names in MESH_GROUPS.items():
        acc = {"fetch_MB_raw": 0.0, "fetch_MB_x2": 0.0, "write_MB": 0.0, "avg_us": 0.0, "kernels": [], "long": {"fetch_MB_raw": 0.0, "write_MB": 0.0, "avg_us": 0.0}}
        for k, rec in out.items():
            base = k.split("<")[0].strip()
            if base in names:
                for f_ in ("fetch_MB_raw", "fetch_MB_x2", "write_MB", "avg_us"):
                    acc[f_] = round(acc[f_] + rec[f_], 3)
                for f_ in ("fetch_MB_raw", "write_MB", "avg_us"):      # per LONG launch of every kernel of the group: the step's launches (bench.py divides by the views they cover)
                    acc["long"][f_] = round(acc["long"][f_] + rec.get("long", rec)[f_], 3)
                acc["kernels"].append(k)
        groups[g] = acc
    out["_groups"] = groups
if len(sys.argv) > 3:
    try:
        import c3d_hip
        out["_meta"] = {"code_digest": c3d_hip.code_digest()}
    except Exception as e:      # never lose a measurement over the stamp
        out["_meta"] = {"code_digest": None, "error": repr(e)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
