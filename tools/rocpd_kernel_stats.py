"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite trace (trace_results.db)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select s.kernel_name, count(*), sum(d.end - d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (disp, sym)).fetchall()
total = sum(r[2] for r in rows)
span = cur.execute("select min(start), max(end) from %s" % disp).fetchone()
print("%-70s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for name, n, t, mn, mx in rows:
    short = name.split("(")[0][-70:]
    print("%-70s %8d %12.1f %10.2f %10.2f %10.2f %6.1f" % (short, n, t / 1e3, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
print("sum of kernel time %.1f us; first-to-last dispatch span %.1f us; kernels busy %.1f%% of span" % (total / 1e3, (span[1] - span[0]) / 1e3, 100.0 * total / (span[1] - span[0])))
# optional: duration histogram of the kernels whose name contains argv[2] (deciles), e.g. to tell the no-op calls of a conditional kernel from the real ones
if len(sys.argv) > 2:
    for name, in cur.execute("select distinct kernel_name from %s" % sym).fetchall():
        if sys.argv[2] not in name:
            continue
        d = sorted(r[0] / 1e3 for r in cur.execute("select d.end - d.start from %s d join %s s on d.kernel_id = s.id where s.kernel_name = ?" % (disp, sym), (name,)))
        if d:
            print("deciles of %s (%d calls): %s" % (name.split("(")[0][-60:], len(d), " ".join("%.1f" % d[min(len(d) - 1, int(q * len(d) / 10))] for q in range(11))))
