"""Per-kernel average / minimum duration and the median launch period from a rocprofv3 results database."""
import glob
import sqlite3
import statistics
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by sum(d.end-d.start) desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 8}")
for r in c.execute(q):
    print(f"{r[0][:70]:70s} calls {r[1]:6d}  avg {r[2] / 1e3:9.2f} us  min {r[3] / 1e3:9.2f} us")
rows = list(c.execute(f"select d.start, d.end from {kd} d order by d.start"))
if len(rows) > 400:
    tail = rows[-400:]
    gaps = [tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)]
    print("median gap between consecutive kernels (last 400): %.2f us" % (statistics.median(gaps) / 1e3))
