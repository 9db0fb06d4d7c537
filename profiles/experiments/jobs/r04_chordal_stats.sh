# chordal initialisation of sphere2500: kernel statistics and wall time
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ch -o c -- python $GRAFT_REPO_ROOT/profiles/experiments/chordal_profile.py > /tmp/ch.log 2>&1
tail -1 /tmp/ch.log
python $GRAFT_REPO_ROOT/profiles/prof_query.py /tmp/ch/c_results.db | head -24
python - <<'P'
import sqlite3
db = sqlite3.connect('/tmp/ch/c_results.db')
rows = list(db.execute("select name, start, end from kernels order by start"))
# last call: from the last k_scatter pair
idx = [i for i, r in enumerate(rows) if 'k_scatter' in r[0]]
k0 = idx[-2]
print("last call: %d launches, %.2f ms from first start to last end" % (len(rows) - k0, (rows[-1][2] - rows[k0][1]) / 1e6))
gaps = 0.0
for a, b in zip(rows[k0:-1], rows[k0 + 1:]): gaps += max(0, b[1] - a[2])
print("idle between kernels: %.2f ms" % (gaps / 1e6))
P
