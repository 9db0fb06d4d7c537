#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd sqlite database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()])
    sys.exit(0)
cols = [d[0] for d in cur.execute("select * from %s limit 1" % view).description]
print("columns:", cols)
kcol = "kernel_name" if "kernel_name" in cols else "name"
q = "select %s, counter_name, count(*), avg(value), min(value), max(value) from %s group by %s, counter_name order by 4 desc" % (kcol, view, kcol)
print("| kernel | counter | dispatches | avg | min | max |\n|---|---|---|---|---|---|")
for r in cur.execute(q):
    name = r[0].split("(")[0].replace("void dpgo::", "").replace("dpgo::", "")
    print("| `%s` | %s | %d | %.1f | %.1f | %.1f |" % (name, r[1], r[2], r[3], r[4], r[5]))
