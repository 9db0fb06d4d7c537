"""share of phase-gated no-op launches in an RTR run: duration histogram of the tCG kernels from a rocprofv3 database"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, end - start from kernels"))
for key in ("k_tcg_hv", "k_precond<5, 2", "k_precond<5, 1", "k_rtr_eval2", "k_retract", "k_rtr_accept"):
    d = sorted((e / 1e3) for n, e in rows if key in n)
    if not d: continue
    short = sum(1 for x in d if x < 5.0)
    print("%-18s calls %6d  <5us %6d (%.0f %%)  median %.2f  p90 %.2f  total %.1f ms (short ones %.1f ms)" % (key, len(d), short, 100.0 * short / len(d), d[len(d) // 2], d[int(0.9 * len(d))], sum(d) / 1e3, sum(x for x in d if x < 5.0) / 1e3))
