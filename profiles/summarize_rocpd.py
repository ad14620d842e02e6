"""Turn a rocprofv3 rocpd database (ROCm 7.2 default output) into the per-kernel stats table we commit.
usage: python profiles/summarize_rocpd.py <results.db> > profiles/<name>_kernel_stats.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                 "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,SGPR,LDS,Scratch")
for r in rows:
    name = r[0].split("(")[0]
    print('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s' % (name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
