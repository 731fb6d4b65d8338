"""Timeline of the last N kernel dispatches before the end of a rocprofv3 rocpd trace (trace_results.db): start relative to the first of them,
duration, queue, kernel -- to see what runs beside what and where a stream waits.  python tools/rocpd_timeline.py trace.db [N] [from_kernel_substring]"""
import sqlite3
import sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if sys.argv[1].endswith(".csv"):
    # a *_kernel_trace.csv of rocprofv3 --output-format csv
    import csv
    with open(sys.argv[1]) as f:
        rows = []
        for r in csv.DictReader(f):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Queue_Id", 0) or 0), r["Kernel_Name"]))
            except (ValueError, TypeError, KeyError):          # a line another process of the run was still writing
                continue
        rows.sort()
else:
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute("select d.start, d.end, %s, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (qcol or "0", disp, sym)).fetchall()
rows = rows[-n:]
if len(sys.argv) > 3:
    k = [i for i, r in enumerate(rows) if sys.argv[3] in r[3]]
    if k: rows = rows[k[0]:]
t0 = rows[0][0]
queues = sorted(set(r[2] for r in rows))
last_end = {}
for s, e, q, name in rows:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-46:]
    print("%9.1f  %7.1f us  q%-2d %s%-46s  (%.1f us after the queue's previous kernel)" % ((s - t0) / 1e3, (e - s) / 1e3, queues.index(q), "    " * queues.index(q), short, gap))
