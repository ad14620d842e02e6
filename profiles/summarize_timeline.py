"""Timeline of one fused step from a rocprofv3 --kernel-trace database: which kernels overlap, how much of the step has NO kernel in flight, how much
has only small (latency-bound) kernels in flight.  usage: python profiles/summarize_timeline.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall() if "stream_id" in cols else \
    [r + (0,) for r in c.execute("select name, start, end from kernels order by start").fetchall()]
short = lambda n: n.split("(")[0].replace("void ", "")
# steps end with k_bwd_views_sh; take the last complete step
ends = [i for i, r in enumerate(rows) if "k_bwd_views_sh" in r[0]]
if len(ends) < 2:
    print("need >= 2 steps in the trace"); sys.exit(0)
lo, hi = ends[-2] + 1, ends[-1]
step = rows[lo:hi + 1]
t0, t1 = min(r[1] for r in step), max(r[2] for r in step)
print("step span %.3f ms, %d kernels, %d streams" % ((t1 - t0) / 1e6, len(step), len({r[3] for r in step})))
HEAVY = ("k_composite_bwd", "k_composite_fwd", "k_preprocess", "k_bwd_views_geom", "k_bwd_views_sh", "k_loss_grad", "k_ms_fwd", "k_ms_bwd", "k_emit")
ev = []
for n, s, e, _ in step:
    h = any(k in n for k in HEAVY)
    ev.append((s, 1, h)); ev.append((e, -1, h))
ev.sort()
n_all = n_heavy = 0
last = t0
idle = light_only = heavy1 = heavy2p = 0
for t, d, h in ev:
    dt = t - last
    if n_all == 0: idle += dt
    elif n_heavy == 0: light_only += dt
    elif n_heavy == 1: heavy1 += dt
    else: heavy2p += dt
    n_all += d
    if h: n_heavy += d
    last = t
tot = t1 - t0
print("no kernel in flight        %.3f ms (%.1f %%)" % (idle / 1e6, 100 * idle / tot))
print("only small kernels         %.3f ms (%.1f %%)" % (light_only / 1e6, 100 * light_only / tot))
print("exactly one heavy kernel   %.3f ms (%.1f %%)" % (heavy1 / 1e6, 100 * heavy1 / tot))
print(">= 2 heavy kernels         %.3f ms (%.1f %%)" % (heavy2p / 1e6, 100 * heavy2p / tot))
by = {}
for n, s, e, _ in step:
    k = short(n); by.setdefault(k, [0, 0.0]); by[k][0] += 1; by[k][1] += (e - s) / 1e6
for k, (cnt, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-44s x%-3d %.3f ms" % (k[:44], cnt, ms))
