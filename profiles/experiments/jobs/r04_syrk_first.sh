# the trailing updates of the batched subdomain inversion in an UPDATE_WEIGHT round: duration and grid of every k_syrk launch
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/upd -o upd -- python $GRAFT_REPO_ROOT/profiles/experiments/gnc_update_profile.py > /tmp/upd.log 2>&1
python - <<'P'
import sqlite3
db = sqlite3.connect('/tmp/upd/upd_results.db')
rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_residuals' in r[0]]
starts = [i for q, i in enumerate(idx) if q == 0 or i - idx[q - 1] > 20]
k0, k1 = starts[4], starts[5]
for name, s, e, gx, gy, gz, wx in rows[k0:k1]:
    if 'k_syrk' in name or 'k_trsm' in name or 'k_potrf' in name or 'k_tl_' in name:
        print("%-22s dur %7.1f us  grid %5d x %3d x %3d" % (name.split('(')[0].replace('void dpgo::','')[-22:], (e - s) / 1e3, gx // max(wx,1), gy, gz))
P
