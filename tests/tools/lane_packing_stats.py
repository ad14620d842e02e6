"""Lane occupancy of the compositing walk and what half-quadrant / strip packing would buy, on the BASELINE view.

    python tests/tools/lane_packing_stats.py [N] [W] [H] [views]        (defaults: 1000000 1920 1080 2 -> profiles/r05_lane_packing_stats.txt)

Runs the float32 oracle forward (test infrastructure; nothing here is product code) on the bench's cloud and cameras, then prices the backward kernel's walk over the
oracle's per-tile lists four ways (lane_packing_stats.c).  The oracle's lists are the dependency's bounding-square lists (5.45 M pairs per view against the kernels'
4.01 M), but positions nobody blends are not walked by either, so the walked (quadrant, pair) sets agree up to float32 decisions."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    from c3d_hip import synthetic as S
    from oracle import gs_oracle as O
    from helpers import oracle_forward
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    V = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    so = os.path.join(ROOT, "oracle", "_build", "lane_packing_stats.so")
    O.build()
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "lane_packing_stats.c"), "-lm"])
    lib = C.CDLL(so)
    nt = os.cpu_count() or 8
    act = S.make_cloud(N, seed=1234, activated=True)
    poses = [(-30.0, 45.0), (0.0, 157.5), (30.0, 270.0), (60.0, 90.0)][:V]
    tot = np.zeros(8)
    lines = []
    for el, az in poses:
        st = S.camera_settings(W, H, 49.1, el, az, 2.2, bg=(1, 1, 1))
        t0 = time.time()
        _, _, _, _, ost = oracle_forward(act, st, dtype=np.float32, nthreads=nt)
        g, b, im = ost.geometry(), ost.binning(), ost.image_state()
        out = np.zeros(8)
        f = lambda a, t: np.ascontiguousarray(a, dtype=t)
        xy, co = f(g["xy"], np.float32), f(g["conic_opacity"], np.float32)
        pl, rg, nc = f(b["point_list"], np.uint32), f(b["ranges"], np.uint32), f(im["n_contrib"], np.uint32)
        lib.lane_packing_stats(W, H, (W + 15) // 16, (H + 15) // 16, xy.ctypes.data_as(C.c_void_p), co.ctypes.data_as(C.c_void_p), pl.ctypes.data_as(C.c_void_p),
                               rg.ctypes.data_as(C.c_void_p), nc.ctypes.data_as(C.c_void_p), nt, out.ctypes.data_as(C.c_void_p))
        lines.append(fmt("view el %g az %g (%.0f s)" % (el, az, time.time() - t0), out))
        tot += out
        fw = np.zeros(5)
        lib.fwd_walk_stats(W, H, (W + 15) // 16, (H + 15) // 16, xy.ctypes.data_as(C.c_void_p), co.ctypes.data_as(C.c_void_p), pl.ctypes.data_as(C.c_void_p),
                           rg.ctypes.data_as(C.c_void_p), nt, fw.ctypes.data_as(C.c_void_p))
        lines.append("    forward walk: %.3f M rectangle hits walked while a pixel of the quadrant is alive, %.3f M of them blend into a pixel (%.1f %%); hits against the alive pixels' bounding box (per 64-entry chunk): "
                     "%.3f M (%.3f x); hits that reach 1/255 on a pixel CENTRE: %.3f M (%.3f x), on the centre of a pixel alive at the chunk start: %.3f M (%.3f x)"
                     % (fw[0] / 1e6, fw[1] / 1e6, 100 * fw[1] / fw[0], fw[2] / 1e6, fw[2] / fw[0], fw[3] / 1e6, fw[3] / fw[0], fw[4] / 1e6, fw[4] / fw[0]))
        del ost
    lines.append(fmt("all %d views" % len(poses), tot))
    txt = "\n".join(["lane packing statistics of the backward compositing walk: %d Gaussians, %d x %d (oracle float32 lists)" % (N, W, H)] + lines)
    print(txt)
    if N == 1_000_000 and (W, H) == (1920, 1080):
        open(os.path.join(ROOT, "profiles", "r05_lane_packing_stats.txt"), "w").write(txt + "\n")


def fmt(name, o):
    now, lanes, adj, s2, s4, half, quads, reached = o
    return ("%s: walked (quadrant, pair) passes %.3f M, lanes on per pass %.1f of 64 (%.0f %%), pairs confined to one 32-lane half %.1f %%\n"
            "    passes with list-ADJACENT opposite halves sharing one: %.3f M (%.3f x)   two independent half-wave streams: %.3f M (%.3f x)   four 16-lane strip streams: %.3f M (%.3f x)"
            % (name, now / 1e6, lanes / max(now, 1), 100 * lanes / max(now, 1) / 64, 100 * half / max(now, 1), adj / 1e6, adj / max(now, 1), s2 / 1e6, s2 / max(now, 1), s4 / 1e6, s4 / max(now, 1)))


if __name__ == "__main__":
    main()
