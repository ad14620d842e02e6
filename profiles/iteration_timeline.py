"""One training iteration of a launch-bound run, kernel by kernel, from a rocprofv3 --kernel-trace database: start offset, duration and the gap in front of every
kernel (iterations end with the fused Adam launch).  usage: python profiles/iteration_timeline.py <results.db> [iteration index from the end, default 20]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = c.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "")[:60]
ends = [i for i, r in enumerate(rows) if "k_adam_multi" in r[0]]
if len(ends) < back + 2:
    print("need more iterations in the trace (%d Adam launches)" % len(ends)); sys.exit(0)
lo, hi = ends[-back - 1] + 1, ends[-back]
it = rows[lo:hi + 1]
t0 = rows[ends[-back - 1]][2]
span = it[-1][2] - t0
busy = sum(e - s for _, s, e in it)
print("iteration: %d launches, %.1f us from the end of the previous Adam launch to the end of this one; kernels %.1f us, gaps %.1f us" % (len(it), span / 1e3, busy / 1e3, (span - busy) / 1e3))
last = t0
for n, s, e in it:
    print("  +%7.1f us  gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - last) / 1e3, (e - s) / 1e3, short(n)))
    last = e
# the same over the last 200 iterations: mean duration and mean gap in front, per kernel name
agg = {}
for k in range(1, min(200, len(ends) - 1)):
    lo, hi = ends[-k - 1] + 1, ends[-k]
    last = rows[ends[-k - 1]][2]
    for n, s, e in rows[lo:hi + 1]:
        a = agg.setdefault(short(n), [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += (s - last) / 1e3
        last = e
nit = min(200, len(ends) - 1) - 1
print("mean over %d iterations, per kernel name: launches per iteration, us of kernel per iteration, us of gap in front per iteration" % nit)
for k, (cnt, du, gp) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("  %-60s x%5.2f  %7.2f  %7.2f" % (k, cnt / nit, du / nit, gp / nit))
