"""Per-kernel summary of a rocprofv3 rocpd database (…_results.db): calls, total ms, average us — what `--stats` prints as CSV."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("%-90s %8s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
for n, c, s, a, mn, mx in rows[:top]:
    print("%-90s %8d %10.3f %9.2f %9.2f %9.2f %6.1f" % (n[:90], c, s / 1e3, a, mn, mx, 100.0 * s / tot))
