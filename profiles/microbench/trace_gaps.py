"""Where does the GPU idle in a kernel trace?  Busy share of the last 60 % of a rocprofv3 --kernel-trace database and the largest gaps by (kernel before, kernel after):
  python profiles/microbench/trace_gaps.py <results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# last 60 % of the trace
rows = rows[int(len(rows)*0.4):]
span = rows[-1][2]-rows[0][1]; busy = sum(e-s for _,s,e in rows)
print("kernels %d, span %.2f ms, busy %.2f ms (%.1f %%)" % (len(rows), span/1e6, busy/1e6, 100*busy/span))
# biggest gaps and what follows them
gaps = sorted(((rows[i][1]-rows[i-1][2], rows[i-1][0].split('(')[0][:40], rows[i][0].split('(')[0][:40]) for i in range(1,len(rows))), reverse=True)[:400]
agg = {}
for g,a,b in gaps:
    k=(a,b); agg.setdefault(k,[0,0]); agg[k][0]+=1; agg[k][1]+=g
for k,(n,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]:
    print("  %6.1f us x%-4d  after %-40s before %s" % (t/n/1e3, n, k[0], k[1]))
