#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel stats and (optionally) the dispatch
timeline of a window, to see inter-kernel gaps inside a hipGraph replay."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for r in cur.execute("select * from top_kernels"):
    name = r[0].split("(")[0].replace("void dpgo::", "").replace("dpgo::", "")
    print("| `%s` | %d | %.1f | %.3f | %.2f |" % (name, r[1], r[2] / 1e3 if r[2] > 1e6 else r[2], r[3], r[4]))
if len(sys.argv) > 2:
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    print(cols)
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if sys.argv[2].isdigit():
        k0 = int(sys.argv[2])
    else:  # first dispatch whose name contains the given substring (e.g. a kernel that only runs in the timed loop)
        k0 = next(i for i, r in enumerate(rows) if sys.argv[2] in r[0]) + 40
    k1 = k0 + int(sys.argv[3]) if len(sys.argv) > 3 else k0 + 40
    base = rows[k0][1]
    for name, s, e in rows[k0:k1]:
        print("%-60s start %9.2f us  dur %7.2f us" % (name.split("(")[0][-58:], (s - base) / 1e3, (e - s) / 1e3))
