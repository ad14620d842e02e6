"""Where does a radix-sort pass spend its time?  Per-tile wall_clock64 stamps (100 MHz) inside k_onesweep (c3d_test_sort_phases), for the two
sorts of a view: 1M depth keys (32 bits, float bits of depths in [1.2, 3.2]) and 4M tile ids (13 bits).  Run on the GPU box:
  python profiles/microbench/sort_phases.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd")]
import c3d_hip as h

lib = h.lib()
NAMES = ["load", "rank", "local scan + publish + LDS reorder", "chained scan", "global stores"]


def run(name, keys, bits):
    n = keys.shape[0]
    passes = (bits + 7) // 8
    nb = (n + 4095) // 4096
    kt = torch.tensor(keys.astype(np.int64), device="cuda").to(torch.int32)
    vt = torch.arange(n, device="cuda", dtype=torch.int32)
    for rep in range(3):           # warm
        k2, v2 = kt.clone(), vt.clone()
        h.check(lib.c3d_test_sort_pairs_u32(h.ptr(k2), h.ptr(v2), n, bits, h.stream()), "sort")
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    k2, v2 = kt.clone(), vt.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.check(lib.c3d_test_sort_pairs_u32(h.ptr(k2), h.ptr(v2), n, bits, h.stream()), "sort")      # includes hipMalloc/hipFree + sync of the test hook
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    stamps = torch.zeros((passes, nb, 8), dtype=torch.int64, device="cuda")
    lib.c3d_test_sort_phases(h.ptr(stamps))
    k2, v2 = kt.clone(), vt.clone()
    h.check(lib.c3d_test_sort_pairs_u32(h.ptr(k2), h.ptr(v2), n, bits, h.stream()), "sort")
    torch.cuda.synchronize()
    lib.c3d_test_sort_phases(None)
    s = stamps.cpu().numpy().astype(np.float64) / 100.0      # us
    order = np.argsort(keys, kind="stable")
    assert (k2.cpu().numpy().view(np.uint32) == keys[order]).all()
    print("%s: n = %d, %d bits, %d tiles, whole test call %.0f us" % (name, n, bits, nb, wall))
    for p in range(passes):
        t = s[p]
        span = t[:, 5].max() - t[:, 0].min()
        ph = [np.mean(t[:, i + 1] - t[:, i]) for i in range(5)]
        ph_max = [np.max(t[:, i + 1] - t[:, i]) for i in range(5)]
        print("  pass %d: kernel span %.1f us | start skew %.1f us | mean per tile: %s | max: %s" %
              (p, span, t[:, 0].max() - t[:, 0].min(), " ".join("%s %.1f" % (NAMES[i], ph[i]) for i in range(5)), " ".join("%.1f" % x for x in ph_max)))


rng = np.random.default_rng(0)
depth = rng.uniform(1.2, 3.2, size=1_000_000).astype(np.float32).view(np.uint32)
run("depth sort", depth, 32)
tiles = rng.integers(0, 8160, size=4_000_000, dtype=np.uint32)
tiles.sort()            # emission order is not sorted by tile, but locally coherent; use a blockwise shuffle
tiles = tiles.reshape(-1, 4).copy(); rng.shuffle(tiles); tiles = tiles.reshape(-1)
run("tile sort", tiles, 13)
