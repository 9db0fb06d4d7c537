# one UPDATE_WEIGHT round of the bench's GNC graph, kernel by kernel (dispatch timeline from the 4th round on)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/upd -o upd -- python $GRAFT_REPO_ROOT/profiles/experiments/gnc_update_profile.py > /tmp/upd.log 2>&1
tail -3 /tmp/upd.log
python - <<'P'
import sqlite3
db = sqlite3.connect('/tmp/upd/upd_results.db')
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_residuals' in r[0]]
# rounds: groups of k_residuals launches (8 agents each)
starts = [i for q, i in enumerate(idx) if q == 0 or i - idx[q - 1] > 20]
k0 = starts[4]; k1 = starts[5]
base = rows[k0][1]
prev_end = base
agg = {}
for name, s, e in rows[k0:k1]:
    nm = name.split('(')[0].replace('void dpgo::', '')[-50:]
    a = agg.setdefault(nm, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(0.0, (s - prev_end) / 1e3)
    prev_end = e
print("round: %d launches, %.1f us first start to last end" % (k1 - k0, (rows[k1 - 1][2] - base) / 1e3))
print("| kernel | launches | busy us | idle in front us |")
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print("| %s | %d | %.1f | %.1f |" % (nm, a[0], a[1], a[2]))
print("--- timeline")
for name, s, e in rows[k0:k1]:
    print("%-50s start %9.1f dur %7.1f" % (name.split('(')[0].replace('void dpgo::', '')[-50:], (s - base) / 1e3, (e - s) / 1e3))
P
